"""Audio frontend modules with the interface of ``howl/data/transform/transform.py``, running on MI355X kernels.

``StandardAudioTransform`` keeps the reference's call protocol (``__call__(audio, mels_only=, deltas_only=)``,
``compute_lengths``, ``.spec_transform.win_length/.hop_length``, train()/eval() selecting VTLP) and its RNG protocol
(one ``self.rand.random()`` draw per call even in eval; VTLP alpha from the *global* ``random`` module).
"""
import math
import random
from dataclasses import dataclass
from types import SimpleNamespace
from typing import Sequence

import numpy as np
import torch
import torch.nn as nn

from howl_amd import ops
from howl_amd.lib import MAX_MELS, fb_packed_floats
from howl_amd.settings import SETTINGS

__all__ = ["AugmentationParameter", "AugmentModule", "StandardAudioTransform", "SpecAugmentTransform",
           "mel_corner_points", "vtlp_warp_points"]


@dataclass
class AugmentationParameter:
    """``transform.py:33-58``."""
    domain: Sequence[float]
    name: str
    current_value_idx: int = None
    prob: float = 0.75
    enabled: bool = True

    def copy_from(self, op: "AugmentationParameter"):
        self.current_value_idx = op.current_value_idx
        self.prob = op.prob
        self.enabled = op.enabled

    @property
    def magnitude(self):
        return self.domain[self.current_value_idx]

    @classmethod
    def from_dict(cls, data_dict):
        return cls(data_dict["domain"], data_dict["name"], data_dict["current_value_idx"], data_dict["prob"])


class AugmentModule(nn.Module):
    """``transform.py:61-97``: per parameter, ``enabled and rand.random() < prob and training`` picks augment."""

    def __init__(self, seed: int = None):
        super().__init__()
        self.augment_params = self.default_params
        self.rand = random if seed is None else random.Random(seed)
        self.seed = seed

    @property
    def default_params(self) -> Sequence[AugmentationParameter]:
        raise NotImplementedError

    def augment(self, param: AugmentationParameter, examples, **kwargs):
        raise NotImplementedError

    def passthrough(self, examples, **kwargs):
        return examples

    def forward(self, x, **kwargs):
        for param in self.augment_params:
            if param.enabled and self.rand.random() < param.prob and self.training:
                x = self.augment(param, x, **kwargs)
            else:
                x = self.passthrough(x, **kwargs)
        return x


def mel_corner_points(n_mels: int, sample_rate: int) -> torch.Tensor:
    """The n_mels + 2 HTK triangle corner frequencies, computed with the same fp32 torch ops as
    ``transform.py:388-392`` so that the device-built filterbank matches the reference's bit for bit."""
    f_min, f_max = 0.0, float(sample_rate // 2)
    m_min = 2595.0 * math.log10(1.0 + (f_min / 700.0))
    m_max = 2595.0 * math.log10(1.0 + (f_max / 700.0))
    m_pts = torch.linspace(m_min, m_max, n_mels + 2)
    return 700.0 * (10 ** (m_pts / 2595.0) - 1.0)


def vtlp_warp_points(f_pts: torch.Tensor, alpha: float, sample_rate: int, f_hi: int = 4800) -> torch.Tensor:
    """VTLP warp of the corner points, statement for statement ``transform.py:394-401`` (including the
    re-evaluation of the ``>`` mask on the already scaled tensor when alpha > 1)."""
    S = sample_rate
    f_pts = f_pts.clone()
    thr = f_hi * min(alpha, 1) / alpha
    f_pts[f_pts <= thr] *= alpha
    f = f_pts[f_pts > thr]
    f_pts[f_pts > thr] = S / 2 - ((S / 2 - f_hi * min(alpha, 1)) / (S / 2 - thr)) * (S / 2 - f)
    return f_pts


def vtlp_warp_points_np(f_pts32: np.ndarray, alpha: float, sample_rate: int, f_hi: int = 4800) -> np.ndarray:
    """``vtlp_warp_points`` on a float32 numpy array: the same IEEE fp32 operations in the same order (scalar operands rounded to
    fp32 first, as ATen does for a Python scalar next to a float tensor; no fused multiply-add on either side), bit for bit the same
    corner points (``tests/test_host.py`` compares them over thousands of draws) at a fifteenth of the host time: the torch version
    is eight tensor ops on 42 elements, ~100 us of a 300-us small-batch step on 75 % of the steps."""
    S = sample_rate
    f = f_pts32.copy()
    thr = f_hi * min(alpha, 1) / alpha
    f[f <= np.float32(thr)] *= np.float32(alpha)
    hi = f > np.float32(thr)                          # re-evaluated on the scaled values: the alpha > 1 quirk (transform.py:399-401)
    c = np.float32((S / 2 - f_hi * min(alpha, 1)) / (S / 2 - thr))
    half = np.float32(S / 2)
    f[hi] = half - c * (half - f[hi])
    return f


class StandardAudioTransform(AugmentModule):
    """``transform.py:234-296``.  (B, L) fp32 PCM on the device -> (B, 3, M, T) [log-mel, deltas, accels]."""

    def __init__(self):
        super().__init__()
        settings = SETTINGS.audio_transform
        if settings.use_meyda_spectrogram:
            raise NotImplementedError("MeydaMelSpectrogram (transform.py:241-247) is outside the MI355X hot path")
        if settings.num_fft != 512 or settings.hop_length != 200:
            raise NotImplementedError("the fused frontend kernel is built for n_fft=512, hop=200 (all reference presets)")
        self.n_mels = settings.num_mels
        self.sample_rate = settings.sample_rate
        spec = SimpleNamespace(n_mels=settings.num_mels, sample_rate=settings.sample_rate, n_fft=settings.num_fft,
                               win_length=settings.num_fft, hop_length=settings.hop_length, f_min=0.0,
                               f_max=float(settings.sample_rate // 2))
        self.spec_transform = spec
        self.vtlp_transform = spec
        self._points = mel_corner_points(self.n_mels, self.sample_rate)  # host, 42 floats
        self._points_np = self._points.numpy().copy()
        if not 1 <= self.n_mels <= MAX_MELS:
            raise NotImplementedError(f"the frontend kernel contracts up to {MAX_MELS} mel bins (NUM_MELS={self.n_mels})")
        self.register_buffer("fb_standard", torch.zeros(fb_packed_floats(self.n_mels)), persistent=False)
        self.register_buffer("fb_vtlp", torch.zeros(fb_packed_floats(self.n_mels)), persistent=False)
        self._fb_ready = None
        self.last_vtlp_alpha = None

    @property
    def default_params(self) -> Sequence[AugmentationParameter]:
        return (AugmentationParameter([0], "vtlp", 0),)

    # -- filterbanks -----------------------------------------------------------------------------------------
    def _standard_fb(self):
        if self._fb_ready != self.fb_standard.data_ptr():
            ops.fb_from_points(self._points.tolist(), self.n_mels, self.sample_rate // 2, self.fb_standard)
            self._fb_ready = self.fb_standard.data_ptr()
        return self.fb_standard

    def _vtlp_fb(self):
        alpha = random.random() * 0.2 + 0.9  # global `random`, as transform.py:441
        self.last_vtlp_alpha = alpha
        pts = vtlp_warp_points_np(self._points_np, alpha, self.sample_rate)
        return ops.fb_from_points(pts, self.n_mels, self.sample_rate // 2, self.fb_vtlp)

    # -- reference protocol ------------------------------------------------------------------------------------
    @torch.no_grad()
    def _execute_op(self, fbp, audio, mels_only=False, deltas_only=False, zmuv_pair=None):
        if deltas_only:
            return ops.deltas(audio.contiguous(), zmuv_pair)
        log_mels = ops.logmel(audio, fbp, self.n_mels, zmuv_pair if mels_only else None, layout=0)
        if mels_only:
            return log_mels
        return ops.deltas(log_mels, zmuv_pair)

    def augment(self, param, examples: torch.Tensor, **kwargs):
        fbp = None if kwargs.get("deltas_only") else self._vtlp_fb()
        return self._execute_op(fbp, examples, **kwargs)

    def passthrough(self, examples: torch.Tensor, **kwargs):
        fbp = None if kwargs.get("deltas_only") else self._standard_fb()
        return self._execute_op(fbp, examples, **kwargs)

    @torch.no_grad()
    def compute_lengths(self, length: torch.Tensor):
        return (torch.div(length - self.spec_transform.win_length, self.spec_transform.hop_length,
                          rounding_mode="floor") + 1).long()

    # -- fused fast path ------------------------------------------------------------------------------------------
    @torch.no_grad()
    def log_mel_for_model(self, audio: torch.Tensor, zmuv) -> torch.Tensor:
        """ZMUV-normalised log-mels as a (B, 1, M, T) *view* of a (B, T, M) buffer -- the layout res8 / the LSTMs
        read directly.  Same RNG protocol as ``forward`` (one draw per call, VTLP when training and < 0.75).
        Equivalent to ``zmuv(self(audio))[:, :1]`` without materialising deltas or a permute."""
        param = self.augment_params[0]
        use_vtlp = param.enabled and self.rand.random() < param.prob and self.training
        fbp = self._vtlp_fb() if use_vtlp else self._standard_fb()
        feat = ops.logmel(audio, fbp, self.n_mels, zmuv.pair() if zmuv is not None else None, layout=1)
        return feat.permute(0, 2, 1).unsqueeze(1)


    @torch.no_grad()
    def log_mel_for_model_args(self, audio: torch.Tensor, zmuv):
        """``log_mel_for_model`` as a deferred call: (``HowlLogmelArgs`` record, the (B, 1, M, T) feature view its launch will fill,
        keep-alive tensors) for an entry point that runs this batch's frontend inside ITS launch (``howl_lstm_fwd_next``: a
        one-batch look-ahead).  Only without a VTLP draw pending (eval mode, or the augment disabled): the draw of a train-mode
        call belongs to the step that consumes the features -- returns None then, and the caller computes them in that step."""
        param = self.augment_params[0]
        if param.enabled and self.training:
            return None
        if param.enabled:
            self.rand.random()  # forward()'s draw happens on every call, eval mode included (transform.py:93): keep the stream's position
        rec, feat, keep = ops.logmel_args(audio, self._standard_fb(), self.n_mels, zmuv.pair() if zmuv is not None else None, layout=1)
        return rec, feat.permute(0, 2, 1).unsqueeze(1), keep


class SpecAugmentTransform(AugmentModule):
    """``transform.py:299-339``: per-sample frequency / time masks.  The draws come from ``self.rand`` in the
    reference's order; the masking itself is one ``howl_specaug_mask`` launch per augment instead of a Python loop
    of slice assignments."""

    @property
    def default_params(self) -> Sequence[AugmentationParameter]:
        return (AugmentationParameter([2, 5, 10, 20, 25], "sa_freq", 2),
                AugmentationParameter([10, 50, 75, 125, 150], "sa_time", 2))

    def _launch(self, x, f0, f, t0, t):
        dev = x.device
        as_dev = lambda a: torch.tensor(a, dtype=torch.int32).to(dev, non_blocking=True)
        return ops.specaug_mask(x, as_dev(f0), as_dev(f), as_dev(t0), as_dev(t))

    def tmask(self, x, T):
        B = x.size(0)
        t0s, ts = [0] * B, [0] * B
        for idx in range(B):
            t = self.rand.randrange(0, T)
            try:
                t0 = self.rand.randrange(0, x.size(3) - t)
            except ValueError:
                continue
            t0s[idx], ts[idx] = t0, t
        return self._launch(x, [0] * B, [0] * B, t0s, ts)

    def fmask(self, x, F):
        B = x.size(0)
        f0s, fs = [0] * B, [0] * B
        for idx in range(B):
            f = self.rand.randrange(0, F)
            f0s[idx], fs[idx] = self.rand.randrange(0, x.size(2) - f), f
        return self._launch(x, f0s, fs, [0] * B, [0] * B)

    @torch.no_grad()
    def augment(self, param, examples, **kwargs):
        if param.name == "sa_freq":
            return self.fmask(examples, param.magnitude)
        if param.name == "sa_time":
            return self.tmask(examples, param.magnitude)
        raise RuntimeError(f"Invalid parameter name for SpecAugmentTransform: {param.name}")
