"""``Res8`` (``howl/model/cnn.py:107-145``) on hand-written gfx950 kernels (``howl_amd/csrc/res8.hip``).

The module keeps the reference's construction order (so ``set_random_seed`` gives the same initial weights), its
``state_dict`` keys / shapes (``conv0.weight``, ``bn{i}.running_mean|running_var|num_batches_tracked``,
``conv{i}.weight``, ``output.weight|bias``) and its call protocol ``model(x: (B, C>=1, M, T), lengths)``; the
submodules are parameter containers only -- forward and backward are single C-ABI calls.
"""
import ctypes
from typing import Tuple

import torch
import torch.nn as nn

from howl_amd import lib as _lib
from howl_amd import ops
from howl_amd.settings import _EnvSettings

from .base import RegisteredModel

__all__ = ["Res8", "Res8Settings"]


class Res8Settings(_EnvSettings):
    num_labels: int = 2
    pooling: Tuple[int, int] = (3, 4)
    num_maps: int = 45


def _vp(t):
    return t.data_ptr()


class _Res8Buffers:
    """Caller-owned activations / workspace for one (B, T) geometry, reused across steps."""

    def __init__(self, B, T, C, device):
        H = T // 3
        f32 = dict(dtype=torch.float32, device=device)
        self.key = (B, T, C, str(device))
        self.s = [torch.empty((B, 45, H, 10), **f32) for _ in range(7)]
        self.bn_stats = torch.zeros((6, 2, 48), **f32)
        self.pooled = torch.empty((B, 48), **f32)
        self.mask0 = torch.empty((B, 45, H, 10), dtype=torch.int16, device=device)
        nbytes = _lib.get().cdll.howl_res8_workspace_bytes(B, T)
        self.ws = torch.empty(nbytes, dtype=torch.uint8, device=device)
        self.saved = _lib.HowlRes8Saved()
        for i in range(7):
            self.saved.s[i] = _vp(self.s[i])
        self.saved.bn_stats = _vp(self.bn_stats)
        self.saved.pooled = _vp(self.pooled)
        self.saved.mask0 = _vp(self.mask0)


class _Res8Function(torch.autograd.Function):
    """autograd seam: forward = howl_res8_fwd, backward = howl_res8_bwd (parameter gradients only; the features
    carry no gradient, as in the reference where they come out of a no_grad frontend)."""

    @staticmethod
    def forward(ctx, module, feat, *params):
        logits = module._launch_forward(feat)
        ctx.module = module
        ctx.feat = feat
        ctx.version = module._fwd_version
        return logits

    @staticmethod
    def backward(ctx, dlogits):
        module = ctx.module
        if ctx.version != module._fwd_version:
            raise RuntimeError("Res8: backward called after a newer forward overwrote the saved activations")
        grads = module._launch_backward(ctx.feat, dlogits.contiguous())
        return (None, None) + tuple(grads)


class Res8(RegisteredModel, name="res8"):
    def __init__(self, num_labels: int, config: Res8Settings = None):
        super().__init__(num_labels)
        config = config or Res8Settings()
        n_maps = config.num_maps
        if n_maps != 45 or tuple(config.pooling) != (3, 4):
            raise NotImplementedError("the MI355X res8 kernels are specialised for num_maps=45, pooling=(3,4)")
        self.conv0 = nn.Conv2d(1, n_maps, (3, 3), padding=(1, 1), bias=False)
        self.pool = nn.AvgPool2d(config.pooling)
        self.n_layers = n_layers = 6
        self.convs = [nn.Conv2d(n_maps, n_maps, (3, 3), padding=1, bias=False) for _ in range(n_layers)]
        for i, conv in enumerate(self.convs):
            self.add_module(f"bn{i + 1}", nn.BatchNorm2d(n_maps, affine=False))
            self.add_module(f"conv{i + 1}", conv)
        self.output = nn.Linear(n_maps, num_labels)
        self._buffers_cache = {}   # (B, T, C, device) -> _Res8Buffers; a handful of geometries (batch max length varies)
        self._fwd_version = 0

    # ---- parameter plumbing ------------------------------------------------------------------------------
    def hot_parameters(self):
        """Parameters in the order of ``HowlRes8Grads``: conv0, conv1..6, output.weight, output.bias."""
        return [self.conv0.weight] + [getattr(self, f"conv{i}").weight for i in range(1, 7)] + \
               [self.output.weight, self.output.bias]

    def _params_struct(self):
        prm = _lib.HowlRes8Params()
        ps = self.hot_parameters()
        for p in ps:
            if not (p.is_cuda and p.is_contiguous() and p.dtype == torch.float32):
                raise _lib.HowlHipError("Res8 parameters must be contiguous fp32 tensors on a HIP device "
                                        "(call .to('cuda') first; there is no CPU fallback)")
        prm.conv0_w = _vp(ps[0])
        for i in range(6):
            bn = getattr(self, f"bn{i + 1}")
            prm.conv_w[i] = _vp(ps[1 + i])
            prm.bn_running_mean[i] = _vp(bn.running_mean)
            prm.bn_running_var[i] = _vp(bn.running_var)
            prm.bn_num_batches[i] = _vp(bn.num_batches_tracked)
        prm.out_w = _vp(ps[7])
        prm.out_b = _vp(ps[8])
        return prm

    def _get_buffers(self, B, T, device):
        key = (B, T, self.num_labels, str(device))
        buf = self._buffers_cache.pop(key, None)
        if buf is None:
            if len(self._buffers_cache) >= 6:
                self._buffers_cache.pop(next(iter(self._buffers_cache)))   # drop the least recently used geometry
            buf = _Res8Buffers(B, T, self.num_labels, device)
        self._buffers_cache[key] = buf
        return buf

    @staticmethod
    def _feat_view(x):
        """(B, C, M, T) any strides -> channel-0 base pointer + (sb, st, sm) element strides."""
        x0 = x[:, 0]
        if not x0.is_cuda or x0.dtype != torch.float32:
            raise _lib.HowlHipError("Res8 input must be an fp32 tensor on a HIP device (no CPU fallback)")
        return x0, x0.stride(0), x0.stride(2), x0.stride(1)

    # ---- launches ----------------------------------------------------------------------------------------------
    def _launch_forward(self, feat, grads_struct=None):
        x0, sb, st, sm = self._feat_view(feat)
        B, M, T = x0.shape
        buf = self._get_buffers(B, T, x0.device)
        logits = torch.empty((B, self.num_labels), dtype=torch.float32, device=x0.device)
        prm = self._params_struct()
        self._fwd_version += 1
        _lib.get().call("howl_res8_fwd", ctypes.byref(prm), ctypes.c_void_p(x0.data_ptr()), sb, st, sm, B, T, M,
                        self.num_labels, int(self.training), ctypes.byref(buf.saved), ctypes.c_void_p(logits.data_ptr()),
                        ctypes.c_void_p(buf.ws.data_ptr()), buf.ws.numel(), ops._stream())
        return logits

    def _launch_backward(self, feat, dlogits, out_grads=None):
        if not self.training:
            raise NotImplementedError("Res8 backward is implemented for training-mode BatchNorm (batch statistics), "
                                      "the only mode the reference trains in")
        x0, sb, st, sm = self._feat_view(feat)
        B, M, T = x0.shape
        buf = self._get_buffers(B, T, x0.device)
        ps = self.hot_parameters()
        grads = out_grads if out_grads is not None else [torch.empty_like(p) for p in ps]
        gr = _lib.HowlRes8Grads()
        gr.conv0_w = _vp(grads[0])
        for i in range(6):
            gr.conv_w[i] = _vp(grads[1 + i])
        gr.out_w = _vp(grads[7])
        gr.out_b = _vp(grads[8])
        prm = self._params_struct()
        _lib.get().call("howl_res8_bwd", ctypes.byref(prm), ctypes.c_void_p(x0.data_ptr()), sb, st, sm, B, T, M,
                        self.num_labels, ctypes.byref(buf.saved), ctypes.c_void_p(dlogits.data_ptr()), ctypes.byref(gr),
                        ctypes.c_void_p(buf.ws.data_ptr()), buf.ws.numel(), ops._stream())
        return grads

    def forward(self, x, lengths=None):
        """x: (B, C>=1, M, T); channel 0 (log-mels) is used, ``lengths`` is ignored (``cnn.py:127-128``)."""
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.hot_parameters()):
            return _Res8Function.apply(self, x, *self.hot_parameters())
        return self._launch_forward(x)
