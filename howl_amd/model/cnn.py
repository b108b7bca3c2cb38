"""``Res8`` (``howl/model/cnn.py:107-145``) on hand-written gfx950 kernels (``howl_amd/csrc/res8.hip``).

The module keeps the reference's construction order (so ``set_random_seed`` gives the same initial weights), its
``state_dict`` keys / shapes (``conv0.weight``, ``bn{i}.running_mean|running_var|num_batches_tracked``,
``conv{i}.weight``, ``output.weight|bias``) and its call protocol ``model(x: (B, C>=1, M, T), lengths)``; the
submodules are parameter containers only -- forward and backward are single C-ABI calls.
"""
import ctypes
import logging
from typing import Tuple

import torch
import torch.nn as nn

from howl_amd import lib as _lib
from howl_amd import ops, parallel
from howl_amd.settings import _EnvSettings

from .base import RegisteredModel

__all__ = ["Res8", "Res8Settings", "MobileNetClassifier"]


class Res8Settings(_EnvSettings):
    num_labels: int = 2
    pooling: Tuple[int, int] = (3, 4)
    num_maps: int = 45


def _vp(t):
    return t.data_ptr()


class _Res8Buffers:
    """Caller-owned activations / workspace for one (B, T, M) geometry, reused across steps.  The activations are the
    library's: (B, 45, T/3, M/4) floats each, in the reference's NCHW order at 40 mel bins and as two 10-column strips per
    utterance at 80 (``include/howl_hip.h``, ``HowlRes8Saved``)."""

    def __init__(self, B, T, C, device, M=40):
        H = T // 3
        f32 = dict(dtype=torch.float32, device=device)
        self.key = (B, T, M, C, str(device))
        n = _lib.get().cdll.howl_res8_saved_floats(B, T, M)     # (B, 45, H, M/4) up to 83 frames; row strips beyond (padded)
        self.s = [torch.empty(n, **f32) for _ in range(7)]
        self.bn_stats = torch.zeros((6, 2, 48), **f32)
        self.pooled = torch.empty((B, 48), **f32)
        self.mask0 = torch.empty(n, dtype=torch.int16, device=device)
        nbytes = _lib.get().cdll.howl_res8_workspace_bytes_mels(B, T, M)
        self.ws = torch.empty(nbytes, dtype=torch.uint8, device=device)
        self.saved = _lib.HowlRes8Saved()
        for i in range(7):
            self.saved.s[i] = _vp(self.s[i])
        self.saved.bn_stats = _vp(self.bn_stats)
        self.saved.pooled = _vp(self.pooled)
        self.saved.mask0 = _vp(self.mask0)


class _Res8EvalBuffers:
    """What an eval-mode forward needs: three activation buffers in rotation (layer i reads s[i-1], s[i-2] and writes s[i]), the
    forward part of the workspace -- 4 activation-sized tensors instead of the 14 of a training step."""

    def __init__(self, B, T, C, device, M=40):
        f32 = dict(dtype=torch.float32, device=device)
        n = _lib.get().cdll.howl_res8_saved_floats(B, T, M)
        self.rot = [torch.empty(n, **f32) for _ in range(3)]
        self.bn_stats = torch.zeros((6, 2, 48), **f32)
        self.pooled = torch.empty((B, 48), **f32)
        self.mask0 = torch.empty(n, dtype=torch.int16, device=device)
        self.ws = torch.empty(_lib.get().cdll.howl_res8_eval_workspace_bytes_mels(B, T, M), dtype=torch.uint8, device=device)
        self.saved = _lib.HowlRes8Saved()
        for i in range(7):
            self.saved.s[i] = _vp(self.rot[i % 3])
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


RES8_MELS = (40, 80)     # envs/res8.env sets 40; 80 is howl/settings.py:32's default


def res8_mels_message():
    from howl_amd.settings import SETTINGS
    n = SETTINGS.audio_transform.num_mels
    if n in RES8_MELS:
        return None
    return (f"Res8 on MI355X is built for NUM_MELS=40 (envs/res8.env) or 80 (stock Howl's default): AvgPool (3,4) over strips of "
            f"40 mel bins, but SETTINGS.audio_transform.num_mels is {n}: export NUM_MELS before howl_amd.settings is imported")


def require_supported_mels(model):
    """For callers that build the frontend from SETTINGS and the model together (training.run.*): fail before the first batch."""
    if isinstance(model, Res8) and res8_mels_message():
        raise ValueError(res8_mels_message())


class Res8(RegisteredModel, name="res8"):
    def __init__(self, num_labels: int, config: Res8Settings = None):
        super().__init__(num_labels)
        config = config or Res8Settings()
        n_maps = config.num_maps
        if n_maps != 45 or tuple(config.pooling) != (3, 4):
            raise NotImplementedError("the MI355X res8 kernels are specialised for num_maps=45, pooling=(3,4)")
        # the reference's default is 80 mel bins (settings.py:32), every res8 preset sets 40 (envs/res8.env); the kernels take
        # both.  Anything else: say so when the model is built -- as a warning, so that a model can still be constructed to load,
        # convert or inspect a state_dict (cnn.py:113 constructs regardless of the settings); the entry points, which build the
        # frontend and the model together, turn it into an error (require_supported_mels), and the first forward on another
        # width raises in any case (_feat_view)
        msg = res8_mels_message()
        if msg:
            logging.getLogger(__name__).warning(msg)
        self.conv0 = nn.Conv2d(1, n_maps, (3, 3), padding=(1, 1), bias=False)
        self.pool = nn.AvgPool2d(config.pooling)
        self.n_layers = n_layers = 6
        self.convs = [nn.Conv2d(n_maps, n_maps, (3, 3), padding=1, bias=False) for _ in range(n_layers)]
        for i, conv in enumerate(self.convs):
            self.add_module(f"bn{i + 1}", nn.BatchNorm2d(n_maps, affine=False))
            self.add_module(f"conv{i + 1}", conv)
        self.output = nn.Linear(n_maps, num_labels)
        self._buffers_cache = {}   # (B, T, C, device) -> _Res8Buffers; a handful of geometries (batch max length varies)
        self._eval_cache = {}      # the same for eval-mode forwards (_Res8EvalBuffers), bounded in BYTES: see _get_eval_buffers
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
            if not (ops.on_device(p) and p.is_contiguous() and p.dtype == torch.float32):
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

    def _get_buffers(self, B, T, device, M=40):
        key = (B, T, M, self.num_labels, str(device))
        buf = self._buffers_cache.pop(key, None)
        if buf is None:
            if len(self._buffers_cache) >= 6:
                self._buffers_cache.pop(next(iter(self._buffers_cache)))   # drop the least recently used geometry
            buf = _Res8Buffers(B, T, self.num_labels, device, M)
        self._buffers_cache[key] = buf
        return buf

    EVAL_CACHE_BYTES = 256 << 20     # eval buffers kept between calls (streaming engines re-use one small geometry)

    def _get_eval_buffers(self, B, T, device, M=40):
        """Eval-mode buffers: cached while small (an engine's (1, 51..83)-frame windows), transient beyond -- an evaluation pass
        batches the windows of many clips (``infer_many``) with a different B per call, and a long clip's strips are large."""
        key = (B, T, M, self.num_labels, str(device))
        buf = self._eval_cache.pop(key, None)
        if buf is None:
            buf = _Res8EvalBuffers(B, T, self.num_labels, device, M)
        size = lambda b: b.ws.numel() + 4 * 3 * b.rot[0].numel() + 2 * b.mask0.numel()
        if size(buf) <= self.EVAL_CACHE_BYTES // 4:
            self._eval_cache[key] = buf
            while sum(size(b) for b in self._eval_cache.values()) > self.EVAL_CACHE_BYTES:
                self._eval_cache.pop(next(iter(self._eval_cache)))
        return buf

    @staticmethod
    def _feat_view(x):
        """(B, C, M, T) any strides -> channel-0 base pointer + (sb, st, sm) element strides."""
        x0 = x[:, 0]
        if not ops.on_device(x0) or x0.dtype != torch.float32:
            raise _lib.HowlHipError("Res8 input must be an fp32 tensor on a HIP device (no CPU fallback)")
        if x0.shape[1] not in RES8_MELS:
            raise ValueError(f"Res8 on MI355X is built for NUM_MELS=40 (envs/res8.env) or 80 (stock default); got "
                             f"{x0.shape[1]} mel bins -- set NUM_MELS before howl_amd.settings is imported")
        if x0.shape[2] < 3:
            raise ValueError(f"Res8 needs at least 3 frames (one pooled row); got T={x0.shape[2]}")
        return x0, x0.stride(0), x0.stride(2), x0.stride(1)

    # ---- launches ----------------------------------------------------------------------------------------------
    MAX_FRAMES = 83     # one utterance's pooled map (27 rows) fits the kernels' tile; longer inputs run as row strips with exchanged
                        # halo rows (howl_res8_fwd / _bwd, training and eval; up to 1024 strips = 82,944 frames).  The windowed
                        # eval-mode forward of rounds 2-5 (howl_res8_fwd_long: overlapping 27-row windows, 2.1 x the arithmetic,
                        # up to 64 windows) stays in the library as an entry point (_launch_forward_long; tests/test_emu_res8.py, test_gpu_res8.py)

    def _launch_forward_long(self, x0, sb, st, sm):
        """``howl_res8_fwd_long`` (overlapping 27-row windows, up to 64 of them): the module itself takes long inputs as row strips
        (``_launch_forward``); the device tests reach the entry point through this wrapper."""
        B, M, T = x0.shape
        nbytes = _lib.get().cdll.howl_res8_long_workspace_bytes_mels(B, T, M)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=x0.device)
        logits = torch.empty((B, self.num_labels), dtype=torch.float32, device=x0.device)
        prm = self._params_struct()
        _lib.get().call("howl_res8_fwd_long", ctypes.byref(prm), ctypes.c_void_p(x0.data_ptr()), sb, st, sm, B, T, M,
                        self.num_labels, ctypes.c_void_p(logits.data_ptr()), ctypes.c_void_p(ws.data_ptr()), ws.numel(),
                        ops._stream())
        return logits

    def _launch_forward(self, feat, grads_struct=None):
        x0, sb, st, sm = self._feat_view(feat)
        B, M, T = x0.shape
        if not self.training:     # cnn.py:127-145 takes any T; eval mode on three rotating activation buffers
            buf = self._get_eval_buffers(B, T, x0.device, M)
        else:
            buf = self._get_buffers(B, T, x0.device, M)
        logits = torch.empty((B, self.num_labels), dtype=torch.float32, device=x0.device)
        prm = self._params_struct()
        self._fwd_version += 1
        _lib.get().call("howl_res8_fwd", ctypes.byref(prm), ctypes.c_void_p(x0.data_ptr()), sb, st, sm, B, T, M,
                        self.num_labels, int(self.training), ctypes.byref(buf.saved), ctypes.c_void_p(logits.data_ptr()),
                        ctypes.c_void_p(buf.ws.data_ptr()), buf.ws.numel(), ops._stream())
        return logits

    XENT_MAX_LABELS = 64    # howl_res8_fwd_xent keeps a row of logits in LDS

    def _launch_forward_xent(self, feat, labels):
        """Training-mode forward with ``nn.CrossEntropyLoss()`` in its last launch (``howl_res8_fwd_xent``): returns (logits,
        nll (B,), dlogits (B, C)); the mean loss comes out of ``_launch_backward(..., xent=(nll, loss))``.  Same bits as
        ``_launch_forward`` + ``ops.xent`` + ``_launch_backward``, two launches fewer."""
        x0, sb, st, sm = self._feat_view(feat)
        B, M, T = x0.shape
        if not self.training or self.num_labels > self.XENT_MAX_LABELS:
            raise NotImplementedError("fused forward + cross-entropy: training mode, <= 64 labels")
        buf = self._get_buffers(B, T, x0.device, M)
        f32 = dict(dtype=torch.float32, device=x0.device)
        logits = torch.empty((B, self.num_labels), **f32)
        nll, dlogits = torch.empty((B,), **f32), torch.empty((B, self.num_labels), **f32)
        labels = labels.to(device=x0.device, dtype=torch.int64).contiguous()
        prm = self._params_struct()
        self._fwd_version += 1
        _lib.get().call("howl_res8_fwd_xent", ctypes.byref(prm), ctypes.c_void_p(x0.data_ptr()), sb, st, sm, B, T, M,
                        self.num_labels, ctypes.byref(buf.saved), _vp(labels), _vp(logits), _vp(nll), _vp(dlogits),
                        ctypes.c_void_p(buf.ws.data_ptr()), buf.ws.numel(), ops._stream())
        return logits, nll, dlogits

    # flat-buffer offset up to which gradients are final only after part 2 of a two-part backward (conv0.weight comes first)
    LATE_GRAD_PARAMS = 1

    def _launch_backward(self, feat, dlogits, out_grads=None, part=0, xent=None, adamw=None):
        """``part`` 0: the whole pass; 1 then 2: the same pass in two calls, everything but conv0's gradient final after
        the first (``howl_res8_bwd_part``; the data-parallel step starts its all-reduce in between).  ``xent`` = (nll, loss)
        after ``_launch_forward_xent``: ``howl_res8_bwd_xent`` (the pooled gradient is in the workspace already; ``loss`` (1,)
        receives the batch mean).  ``adamw`` (with ``xent``, part 0) = (flat params, flat grads, m, v, lr, (beta1, beta2), eps,
        weight_decay, step, grad_scale): the optimiser step is part of the call (``HowlAdamW``: inside the last fold launch when
        ``out_grads`` are the flat buffer's views); ``self.optimizer_step_done`` says whether it was taken."""
        self.optimizer_step_done = False
        if not self.training:
            raise NotImplementedError("Res8 backward is implemented for training-mode BatchNorm (batch statistics), "
                                      "the only mode the reference trains in")
        x0, sb, st, sm = self._feat_view(feat)
        B, M, T = x0.shape
        buf = self._get_buffers(B, T, x0.device, M)
        ps = self.hot_parameters()
        grads = out_grads if out_grads is not None else [torch.empty_like(p) for p in ps]
        gr = _lib.HowlRes8Grads()
        gr.conv0_w = _vp(grads[0])
        for i in range(6):
            gr.conv_w[i] = _vp(grads[1 + i])
        gr.out_w = _vp(grads[7])
        gr.out_b = _vp(grads[8])
        prm = self._params_struct()
        if xent is not None:
            nll, loss = xent
            opt = None
            if adamw is not None and part == 0:
                flat, fgrad, m, v, lr, betas, eps, wd, step, gscale = adamw
                opt = ctypes.byref(_lib.HowlAdamW(_vp(flat), _vp(fgrad), _vp(m), _vp(v), flat.numel(), lr, betas[0], betas[1], eps, wd,
                                                  step, gscale))
            _lib.get().call("howl_res8_bwd_xent", ctypes.byref(prm), ctypes.c_void_p(x0.data_ptr()), sb, st, sm, B, T, M,
                            self.num_labels, ctypes.byref(buf.saved), ctypes.c_void_p(dlogits.data_ptr()), _vp(nll), _vp(loss),
                            ctypes.byref(gr), ctypes.c_void_p(buf.ws.data_ptr()), buf.ws.numel(), int(part), opt, ops._stream())
            self.optimizer_step_done = opt is not None
            return grads
        _lib.get().call("howl_res8_bwd_part", ctypes.byref(prm), ctypes.c_void_p(x0.data_ptr()), sb, st, sm, B, T, M,
                        self.num_labels, ctypes.byref(buf.saved), ctypes.c_void_p(dlogits.data_ptr()), ctypes.byref(gr),
                        ctypes.c_void_p(buf.ws.data_ptr()), buf.ws.numel(), int(part), ops._stream())
        return grads

    def forward(self, x, lengths=None):
        """x: (B, C>=1, M, T); channel 0 (log-mels) is used, ``lengths`` is ignored (``cnn.py:127-128``)."""
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.hot_parameters()):
            return _Res8Function.apply(self, x, *self.hot_parameters())
        return self._launch_forward(x)


# =============================================================================================================
# MobileNetClassifier ("mobilenet", howl/model/cnn.py:15-29; BASELINE configs[4])
# =============================================================================================================
class _InvertedResidualParams(nn.Module):
    """Parameter container with torchvision's ``InvertedResidual`` key layout (``.conv.{j}...``)."""

    def __init__(self, inp, oup, stride, t):
        super().__init__()
        hidden = inp * t
        layers = []
        if t != 1:
            layers.append(nn.Sequential(nn.Conv2d(inp, hidden, 1, bias=False), nn.BatchNorm2d(hidden), nn.ReLU6()))
        layers.append(nn.Sequential(nn.Conv2d(hidden, hidden, 3, stride, 1, groups=hidden, bias=False), nn.BatchNorm2d(hidden),
                                    nn.ReLU6()))
        layers += [nn.Conv2d(hidden, oup, 1, bias=False), nn.BatchNorm2d(oup)]
        self.conv = nn.Sequential(*layers)


class _MobileNetV2Params(nn.Module):
    """torchvision ``MobileNetV2`` (width 1.0) as a parameter container: same module tree, hence the same ``state_dict``
    keys and shapes, and torchvision's initialisation (kaiming-normal fan-out convolutions, unit BatchNorm,
    N(0, 0.01) classifier).  The reference starts from ImageNet weights (``mobilenet_v2(pretrained=True)``, cnn.py:22),
    which need the network; load them with ``load_state_dict`` where available."""
    SETTING = [[1, 16, 1, 1], [6, 24, 2, 2], [6, 32, 3, 2], [6, 64, 4, 2], [6, 96, 3, 1], [6, 160, 3, 2], [6, 320, 1, 1]]

    def __init__(self, num_classes):
        super().__init__()
        feats = [nn.Sequential(nn.Conv2d(3, 32, 3, 2, 1, bias=False), nn.BatchNorm2d(32), nn.ReLU6())]
        inp = 32
        for t, c, n, s in self.SETTING:
            for i in range(n):
                feats.append(_InvertedResidualParams(inp, c, s if i == 0 else 1, t))
                inp = c
        feats.append(nn.Sequential(nn.Conv2d(inp, 1280, 1, bias=False), nn.BatchNorm2d(1280), nn.ReLU6()))
        self.features = nn.Sequential(*feats)
        self.classifier = nn.Sequential(nn.Dropout(0.2), nn.Linear(1280, num_classes))
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out")
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.ones_(m.weight)
                nn.init.zeros_(m.bias)
            elif isinstance(m, nn.Linear):
                nn.init.normal_(m.weight, 0, 0.01)
                nn.init.zeros_(m.bias)


class _MobileNetFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, module, feat, *params):
        logits = module._launch_forward(feat)
        ctx.module, ctx.feat, ctx.version = module, feat, module._fwd_version
        return logits

    @staticmethod
    def backward(ctx, dlogits):
        module = ctx.module
        if ctx.version != module._fwd_version:
            raise RuntimeError("MobileNetClassifier: backward called after a newer forward overwrote the saved activations")
        grads = module._launch_backward(ctx.feat, dlogits.contiguous())
        return (None, None) + tuple(grads)


class MobileNetClassifier(RegisteredModel, name="mobilenet"):
    """``MobileNetClassifier`` of ``cnn.py:15-29`` on the kernels of ``howl_amd/csrc/mobilenet.hip``.  The submodules are
    parameter containers (reference / torchvision ``state_dict`` keys); all parameters are views into one flat buffer whose
    layout the library defines (``howl_mobilenet_layer``), forward and backward are single C-ABI calls."""

    def __init__(self, num_labels: int):
        super().__init__(num_labels)
        self.downsample = nn.Sequential(nn.Conv2d(1, 3, 3, padding=(1, 3)), nn.BatchNorm2d(3), nn.ReLU(),
                                        nn.MaxPool2d((1, 2)))
        self.model = _MobileNetV2Params(num_labels)
        self.dropout_p = 0.2
        self.forced_keep_mask = None     # tests: (B, 1280) 0/1 mask used instead of a fresh draw
        self._table = None
        self._flat = self._bflat = None
        self._ws_cache = {}
        self._fwd_version = 0
        self._last_mask = None

    # ---- layer table / parameter plumbing ----------------------------------------------------------------------
    def _layer_modules(self):
        """[(conv, bn, HowlMbLayer)] in the library's layer order, resolved from its state-dict key scheme."""
        if self._table is None:
            lb = _lib.get()
            table = []
            for i in range(lb.cdll.howl_mobilenet_num_layers()):
                d = _lib.HowlMbLayer()
                lb.call("howl_mobilenet_layer", i, ctypes.byref(d))
                if d.feat < 0:
                    conv, bn = self.downsample[0], self.downsample[1]
                elif d.sub < 0:
                    conv, bn = self.model.features[d.feat][0], self.model.features[d.feat][1]
                elif d.wrapped:
                    conv, bn = self.model.features[d.feat].conv[d.sub][0], self.model.features[d.feat].conv[d.sub][1]
                else:
                    conv, bn = self.model.features[d.feat].conv[d.sub], self.model.features[d.feat].conv[d.sub + 1]
                table.append((conv, bn, d))
            self._table = table
        return self._table

    def hot_parameters(self):
        """Parameters in the flat layout's order."""
        ps = []
        for conv, bn, d in self._layer_modules():
            ps.append(conv.weight)
            if d.bias:
                ps.append(conv.bias)
            ps += [bn.weight, bn.bias]
        lin = self.model.classifier[1]
        return ps + [lin.weight, lin.bias]

    def _stat_buffers(self):
        out = []
        for _, bn, _ in self._layer_modules():
            out += [bn.running_mean, bn.running_var]
        return out

    @staticmethod
    def _is_flat(tensors, flat):
        if flat is None or not ops.on_device(tensors[0]) or tensors[0].device != flat.device:
            return False
        off = 0
        for t in tensors:
            if t.data_ptr() != flat.data_ptr() + 4 * off or not t.is_contiguous():
                return False
            off += t.numel()
        return off == flat.numel()

    def _ensure_flat(self):
        """Re-home parameters / BN statistics into flat buffers if they are not already laid out that way (first call,
        after ``.to(device)``, after ``load_state_dict`` with ``assign``, or when a trainer re-homed them itself)."""
        ps = self.hot_parameters()
        for p in ps:
            if not (ops.on_device(p) and p.dtype == torch.float32):
                raise _lib.HowlHipError("MobileNetClassifier parameters must be fp32 tensors on a HIP device "
                                        "(call .to('cuda') first; there is no CPU fallback)")
        if not self._is_flat(ps, self._flat):
            # another owner (training.fused.FlatParams) may already hold them contiguously in the right order
            base = ps[0]
            n = sum(p.numel() for p in ps)
            probe = None
            if base.is_contiguous() and base.untyped_storage().nbytes() >= 4 * (base.storage_offset() + n):
                probe = torch.as_strided(base.detach(), (n,), (1,), base.storage_offset())
            if probe is not None and self._is_flat(ps, probe):
                self._flat = probe
            else:
                flat = torch.empty(n, dtype=torch.float32, device=base.device)
                off = 0
                with torch.no_grad():
                    for p in ps:
                        flat[off:off + p.numel()].copy_(p.data.reshape(-1))
                        p.data = flat[off:off + p.numel()].view(p.shape)
                        off += p.numel()
                self._flat = flat
        bs = self._stat_buffers()
        if not self._is_flat(bs, self._bflat):
            n = sum(b.numel() for b in bs)
            bflat = torch.empty(n, dtype=torch.float32, device=bs[0].device)
            off = 0
            with torch.no_grad():
                for _, bn, _ in self._layer_modules():
                    for name in ("running_mean", "running_var"):
                        b = getattr(bn, name)
                        bflat[off:off + b.numel()].copy_(b.reshape(-1))
                        setattr(bn, name, bflat[off:off + b.numel()].view(b.shape))
                        off += b.numel()
            self._bflat = bflat
        lb = _lib.get()
        assert self._flat.numel() == lb.cdll.howl_mobilenet_param_floats(self.num_labels)
        assert self._bflat.numel() == lb.cdll.howl_mobilenet_buffer_floats()

    def _workspace(self, B, M, T, device):
        key = (B, M, T, str(device))
        ws = self._ws_cache.pop(key, None)
        if ws is None:
            if len(self._ws_cache) >= 3:
                self._ws_cache.pop(next(iter(self._ws_cache)))
            nbytes = _lib.get().cdll.howl_mobilenet_workspace_bytes(B, M, T, self.num_labels)
            ws = torch.empty(nbytes, dtype=torch.uint8, device=device)
        self._ws_cache[key] = ws
        return ws

    EVAL_CACHE_BYTES = 256 << 20     # eval buffers kept between calls (streaming engines re-use one small geometry)

    def _get_eval_buffers(self, B, T, device, M=40):
        """Eval-mode buffers: cached while small (an engine's (1, 51..83)-frame windows), transient beyond -- an evaluation pass
        batches the windows of many clips (``infer_many``) with a different B per call, and a long clip's strips are large."""
        key = (B, T, M, self.num_labels, str(device))
        buf = self._eval_cache.pop(key, None)
        if buf is None:
            buf = _Res8EvalBuffers(B, T, self.num_labels, device, M)
        size = lambda b: b.ws.numel() + 4 * 3 * b.rot[0].numel() + 2 * b.mask0.numel()
        if size(buf) <= self.EVAL_CACHE_BYTES // 4:
            self._eval_cache[key] = buf
            while sum(size(b) for b in self._eval_cache.values()) > self.EVAL_CACHE_BYTES:
                self._eval_cache.pop(next(iter(self._eval_cache)))
        return buf

    @staticmethod
    def _feat_view(x):
        x0 = x[:, 0]
        if not ops.on_device(x0) or x0.dtype != torch.float32:
            raise _lib.HowlHipError("MobileNetClassifier input must be an fp32 tensor on a HIP device (no CPU fallback)")
        return x0, x0.stride(0), x0.stride(1), x0.stride(2)   # (B, M, T): sb, sm, st

    def _mask_args(self):
        m = self._last_mask
        if m is None:
            return None, 1.0
        return ctypes.c_void_p(m.data_ptr()), 1.0 / (1.0 - self.dropout_p)

    # ---- launches ----------------------------------------------------------------------------------------------
    def _launch_forward(self, feat):
        self._ensure_flat()
        x0, sb, sm, st = self._feat_view(feat)
        B, M, T = x0.shape
        ws = self._workspace(B, M, T, x0.device)
        logits = torch.empty((B, self.num_labels), dtype=torch.float32, device=x0.device)
        self._last_mask = None
        if self.training:
            if self.forced_keep_mask is not None:
                self._last_mask = self.forced_keep_mask.to(device=x0.device, dtype=torch.float32).contiguous()
            elif self.dropout_p > 0:
                # one launch of the counter-based device generator; the key comes from torch's CPU generator (a host draw:
                # reproducible under torch.manual_seed, nothing read back from the device)
                self._last_mask = torch.empty((B, 1280), dtype=torch.float32, device=x0.device)
                key = int(torch.randint(0, 2 ** 62, (1,)).item())
                key ^= (parallel.world_info()[0] * 0x9E3779B97F4A7C15) & (2 ** 62 - 1)   # replicas share the CPU seed, not the mask
                _lib.get().call("howl_dropout_mask", ctypes.c_void_p(self._last_mask.data_ptr()), self._last_mask.numel(),
                                float(self.dropout_p), ctypes.c_ulonglong(key), ops._stream())
            torch._foreach_add_([bn.num_batches_tracked for _, bn, _ in self._layer_modules()], 1)
        mask, scale = self._mask_args()
        self._fwd_version += 1
        _lib.get().call("howl_mobilenet_fwd", ctypes.c_void_p(self._flat.data_ptr()), ctypes.c_void_p(self._bflat.data_ptr()),
                        self.num_labels, ctypes.c_void_p(x0.data_ptr()), sb, sm, st, B, M, T, int(self.training), mask, scale,
                        ctypes.c_void_p(logits.data_ptr()), ctypes.c_void_p(ws.data_ptr()), ws.numel(), ops._stream())
        return logits

    def _launch_backward(self, feat, dlogits, out_grads=None):
        if not self.training:
            raise NotImplementedError("MobileNetClassifier backward is implemented for training-mode BatchNorm")
        x0, sb, sm, st = self._feat_view(feat)
        B, M, T = x0.shape
        ws = self._workspace(B, M, T, x0.device)
        ps = self.hot_parameters()
        if out_grads is not None:
            g0, n = out_grads[0], self._flat.numel()
            if g0.untyped_storage().nbytes() < 4 * (g0.storage_offset() + n):
                raise _lib.HowlHipError("out_grads must be views of one flat buffer in hot_parameters() order")
            gflat = torch.as_strided(g0, (n,), (1,), g0.storage_offset())
            if not self._is_flat(list(out_grads), gflat):
                raise _lib.HowlHipError("out_grads must be views of one flat buffer in hot_parameters() order")
            grads = out_grads
        else:
            gflat = torch.empty_like(self._flat)
            grads, off = [], 0
            for p in ps:
                grads.append(gflat[off:off + p.numel()].view(p.shape))
                off += p.numel()
        mask, scale = self._mask_args()
        _lib.get().call("howl_mobilenet_bwd", ctypes.c_void_p(self._flat.data_ptr()), self.num_labels,
                        ctypes.c_void_p(x0.data_ptr()), sb, sm, st, B, M, T, mask, scale, ctypes.c_void_p(dlogits.data_ptr()),
                        ctypes.c_void_p(gflat.data_ptr()), ctypes.c_void_p(ws.data_ptr()), ws.numel(), ops._stream())
        return grads

    def forward(self, x, lengths=None):
        """x: (B, C>=1, M, T); only channel 0 (log-Mels) is used (``cnn.py:27``), ``lengths`` is ignored."""
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.hot_parameters()):
            return _MobileNetFunction.apply(self, x, *self.hot_parameters())
        return self._launch_forward(x)
