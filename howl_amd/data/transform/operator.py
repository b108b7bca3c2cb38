"""``ZmuvTransform`` and the collate helpers of ``howl/data/transform/operator.py`` on the MI355X path."""
from typing import Iterable

import torch
import torch.nn as nn

from howl_amd import ops

__all__ = ["ZmuvTransform", "Composition", "compose", "identity"]


class Composition(nn.Module):
    """``operator.py:24-34``."""

    def __init__(self, modules):
        super().__init__()
        self.modules_ = modules
        self._module_list = nn.ModuleList([m for m in modules if isinstance(m, nn.Module)])

    def forward(self, *args):
        for mod in self.modules_:
            args = (mod(*args),)
        return args[0]


def compose(*collate_modules):
    return Composition(collate_modules)


def identity(x):
    return x


class ZmuvTransform(nn.Module):
    """Running scalar mean / mean-of-squares normaliser (``operator.py:119-146``); buffers ``total, mean, mean2``
    (each shape ``(1,)``) so ``zmuv.pt.bin`` files are interchangeable.  ``update`` and ``forward`` run in the
    ``howl_zmuv_update`` / ``howl_zmuv_apply`` kernels; nothing synchronises the host."""

    def __init__(self):
        super().__init__()
        self.register_buffer("total", torch.zeros(1))
        self.register_buffer("mean", torch.zeros(1))
        self.register_buffer("mean2", torch.zeros(1))
        self._scratch = None
        self._pair = None
        self._pair_key = None      # (data_ptr, version) of mean / mean2 the cached pair was computed from

    def _dev_scratch(self):
        dev = self.mean.device
        if self._scratch is None or self._scratch.device != dev:
            self._scratch = torch.zeros(3, dtype=torch.float64, device=dev)
            self._pair = torch.zeros(2, dtype=torch.float32, device=dev)
        return self._scratch

    def update(self, data, mask=None):
        with torch.no_grad():
            if mask is not None:
                # masked variant (operator.py:128-130): sums over data * mask (broadcast), element count = the sum of the mask AS
                # GIVEN (a (B,1,1,T) mask counts B*T, not B*3*M*T), all on the device
                m = mask.to(device=data.device, dtype=torch.float32)
                scale = m.numel() / data.numel()
                ops.zmuv_update_masked(data.contiguous(), m.expand_as(data).contiguous(), self.total, self.mean, self.mean2,
                                       self._dev_scratch(), count_scale=scale)
                self._pair_key = None
                return
            ops.zmuv_update(data.contiguous(), self.total, self.mean, self.mean2, self._dev_scratch())
            self._pair_key = None      # the kernel wrote the buffers behind torch's back

    def initialize(self, iterable: Iterable[torch.Tensor]):
        for ex in iterable:
            self.update(ex)

    @property
    def std(self):
        return (self.mean2 - self.mean ** 2).sqrt()

    def pair(self) -> torch.Tensor:
        """Device tensor ``[mean, std]`` for the fused epilogues; recomputed (one tiny launch, no sync) only when the
        statistics changed since the last call -- not once per training step."""
        self._dev_scratch()
        key = (self.mean.data_ptr(), self.mean._version, self.mean2.data_ptr(), self.mean2._version, self._pair.data_ptr())
        if key != self._pair_key:
            ops.zmuv_pair(self.mean, self.mean2, self._pair)
            self._pair_key = key
        return self._pair

    def forward(self, x):
        return ops.zmuv_apply(x.contiguous(), self.pair())
