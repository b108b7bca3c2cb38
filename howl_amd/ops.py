"""Thin torch-tensor front of the C ABI: pointer / stream plumbing only, no arithmetic.

Every function requires HIP-resident, contiguous fp32 tensors and enqueues on torch's current stream.  There is
deliberately no CPU branch: a CPU tensor raises.
"""
import ctypes
import math

import numpy as np
import torch

from . import lib as _lib

HOP = 200
N_FFT = 512


def on_device(t) -> bool:
    """The single residency check of the package: every tensor handed to the C ABI must live in HIP device memory."""
    return t.is_cuda


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t, dtype=torch.float32, allow_none=False):
    if t is None:
        if allow_none:
            return None
        raise ValueError("tensor required")
    if not on_device(t):
        raise _lib.HowlHipError("howl_amd ops need HIP-device tensors (no CPU fallback); got a CPU tensor")
    if t.dtype != dtype:
        raise TypeError(f"expected {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError("expected a contiguous tensor")
    return ctypes.c_void_p(t.data_ptr())


def num_frames(length: int) -> int:
    return 1 + length // HOP


def fb_pack(fb: torch.Tensor) -> torch.Tensor:
    """(257, M) filterbank on the device -> packed operand (one (260, 48) bank, two beyond 48 mel bins)."""
    out = torch.empty(_lib.fb_packed_floats(fb.shape[1]), dtype=torch.float32, device=fb.device)
    _lib.get().call("howl_fb_pack", _p(fb), fb.shape[1], _p(out), _stream())
    return out


def fb_from_points(f_pts, n_mels: int, nyquist: float, out: torch.Tensor) -> torch.Tensor:
    """M+2 corner frequencies (host floats) -> packed filterbank written into ``out`` (device, ``fb_packed_floats(M)`` floats)."""
    if out.numel() < _lib.fb_packed_floats(n_mels):
        raise ValueError(f"packed filterbank buffer holds {out.numel()} floats, {n_mels} mel bins need {_lib.fb_packed_floats(n_mels)}")
    pts = _lib.HowlMelPoints()
    vals = np.asarray(f_pts, dtype=np.float32)
    np.frombuffer(pts, dtype=np.float32)[:vals.size] = vals      # (one block copy: a Python loop over 42 floats cost 9 us)
    _lib.get().call("howl_fb_from_points", ctypes.byref(pts), n_mels, float(nyquist), _p(out), _stream())
    return out


def logmel(pcm: torch.Tensor, fbp: torch.Tensor, n_mels: int, zmuv_pair=None, layout: int = 0,
           log_eps: float = 1e-7) -> torch.Tensor:
    """(B, L) PCM -> log-mel (B, M, T) [layout 0] or (B, T, M) [layout 1]; optional fused ZMUV."""
    if pcm.dim() != 2:
        raise ValueError("pcm must be (B, L)")
    if pcm.stride(1) != 1:
        pcm = pcm.contiguous()
    B, L = pcm.shape
    T = num_frames(L)
    out = torch.empty((B, n_mels, T) if layout == 0 else (B, T, n_mels), dtype=torch.float32, device=pcm.device)
    if not on_device(pcm) or pcm.dtype != torch.float32:
        _p(pcm)
    _lib.get().call("howl_logmel_fwd", ctypes.c_void_p(pcm.data_ptr()), B, L, pcm.stride(0), _p(fbp), n_mels,
                    log_eps, _p(zmuv_pair, allow_none=True), _p(out), layout, _stream())
    return out


def logmel_args(pcm: torch.Tensor, fbp: torch.Tensor, n_mels: int, zmuv_pair=None, layout: int = 0, log_eps: float = 1e-7):
    """``logmel``'s call as a ``HowlLogmelArgs`` record for entry points that run the frontend of the NEXT batch themselves
    (``howl_lstm_fwd_next``): returns (record, out tensor, tensors the record points into -- keep them alive until the call)."""
    if pcm.dim() != 2:
        raise ValueError("pcm must be (B, L)")
    if pcm.stride(1) != 1:
        pcm = pcm.contiguous()
    B, L = pcm.shape
    T = num_frames(L)
    out = torch.empty((B, n_mels, T) if layout == 0 else (B, T, n_mels), dtype=torch.float32, device=pcm.device)
    if not on_device(pcm) or pcm.dtype != torch.float32:
        _p(pcm)
    rec = _lib.HowlLogmelArgs(ctypes.c_void_p(pcm.data_ptr()), B, L, pcm.stride(0), _p(fbp), n_mels, log_eps,
                              _p(zmuv_pair, allow_none=True), _p(out), layout)
    return rec, out, (pcm, fbp, zmuv_pair)


def deltas(logmel_bmt: torch.Tensor, zmuv_pair=None) -> torch.Tensor:
    B, M, T = logmel_bmt.shape
    out = torch.empty((B, 3, M, T), dtype=torch.float32, device=logmel_bmt.device)
    _lib.get().call("howl_deltas_fwd", _p(logmel_bmt), B, M, T, _p(zmuv_pair, allow_none=True), _p(out), _stream())
    return out


def zmuv_update(x, total, mean, mean2, scratch):
    _lib.get().call("howl_zmuv_update", _p(x), x.numel(), _p(total), _p(mean), _p(mean2),
                    _p(scratch, torch.float64), _stream())


def zmuv_update_masked(x, mask, total, mean, mean2, scratch, count_scale=1.0):
    """``mask`` already expanded to ``x``'s shape; ``count_scale`` = (unexpanded mask size) / x.numel()."""
    _lib.get().call("howl_zmuv_update_masked", _p(x), _p(mask), x.numel(), float(count_scale), _p(total), _p(mean), _p(mean2),
                    _p(scratch, torch.float64), _stream())


def zmuv_pair(mean, mean2, out):
    _lib.get().call("howl_zmuv_pair", _p(mean), _p(mean2), _p(out), _stream())
    return out


def zmuv_apply(x, pair):
    out = torch.empty_like(x)
    _lib.get().call("howl_zmuv_apply", _p(x), x.numel(), _p(pair), _p(out), _stream())
    return out


def collate_augment(bank, idx, src_len, shift, from_head, sigma, sp_prob, seed, lout, mix=None, dst_off=None):
    """``mix`` = (bg_bank (N, Lbg), bg_idx, bg_off, alpha) puts DatasetMixer in front of the chain; ``dst_off`` (int32 per
    row) places the samples that many columns into the zero row (frame batchifier's front padding)."""
    B = idx.numel()
    out = torch.empty((B, lout), dtype=torch.float32, device=bank.device)
    if mix is None:
        bg, bg_ld, bg_idx, bg_off, alpha = None, 0, None, None, None
    else:
        bg, bg_ld = _p(mix[0]), mix[0].stride(0)
        bg_idx, bg_off, alpha = _p(mix[1], torch.int32), _p(mix[2], torch.int32), _p(mix[3])
    _lib.get().call("howl_collate_augment_window", _p(bank), bank.stride(0), _p(idx, torch.int32), _p(src_len, torch.int32),
                    _p(shift, torch.int32), _p(from_head, torch.int32), _p(sigma), _p(sp_prob),
                    ctypes.c_ulonglong(seed & 0xFFFFFFFFFFFFFFFF), bg, bg_ld, bg_idx, bg_off, alpha,
                    _p(dst_off, torch.int32, allow_none=True), B, lout, _p(out), _stream())
    return out


def gather_windows(bank, idx, start, length, dst_off, lout):
    """Rows ``bank[idx[b], start[b]:start[b]+length[b]]`` placed at column ``dst_off[b]`` of a zero (B, lout) batch."""
    B = idx.numel()
    out = torch.empty((B, lout), dtype=torch.float32, device=bank.device)
    _lib.get().call("howl_gather_windows", _p(bank), bank.stride(0), _p(idx, torch.int32), _p(start, torch.int32),
                    _p(length, torch.int32), _p(dst_off, torch.int32), B, lout, _p(out), _stream())
    return out


def specaug_mask(x, f0, f, t0, t):
    B, C, M, T = x.shape
    if not on_device(x) or x.dtype != torch.float32:
        _p(x)
    sb, sc, sm, st = x.stride()
    _lib.get().call("howl_specaug_mask", ctypes.c_void_p(x.data_ptr()), B, C, M, T, sb, sc, sm, st, _p(f0, torch.int32),
                    _p(f, torch.int32), _p(t0, torch.int32), _p(t, torch.int32), _stream())
    return x


def xent(logits, labels, want_grad=True):
    B, C = logits.shape
    loss = torch.empty(1, dtype=torch.float32, device=logits.device)
    dlogits = torch.empty_like(logits) if want_grad else None
    _lib.get().call("howl_xent_fwd_bwd", _p(logits), _p(labels, torch.int64), B, C, _p(loss),
                    _p(dlogits, allow_none=True), _stream())
    return loss, dlogits


def _ctc_launch(scores, targets, input_lengths, target_lengths, blank, max_target, want_grad, defer_mean=False):
    T, B, C = scores.shape
    dev = scores.device
    # the gradient takes the memory layout of the model's (B, T, C) output buffer, handed back as a (T, B, C) view
    dlogits = torch.empty((B, T, C), dtype=torch.float32, device=dev).permute(1, 0, 2) if want_grad else None
    nll = torch.empty(B, dtype=torch.float32, device=dev)
    loss = torch.empty((), dtype=torch.float32, device=dev)
    vp = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())     # strided views: strides are passed explicitly
    # whole clips (more than 128 frames): the alpha rows of the windows behind the last one wait in a workspace
    ws_floats = int(_lib.get().cdll.howl_ctc_workspace_floats(T, B)) if want_grad else 0
    ws = torch.empty(ws_floats, dtype=torch.float32, device=dev) if ws_floats else None
    _lib.get().call("howl_ctc_loss", vp(scores), scores.stride(0), scores.stride(1), T, B, C, vp(targets),
                    targets.stride(0), max_target, _p(input_lengths, torch.int64), _p(target_lengths, torch.int64), blank,
                    _p(nll), None if defer_mean else _p(loss), vp(dlogits), 0 if dlogits is None else dlogits.stride(0),
                    0 if dlogits is None else dlogits.stride(1), vp(ws), ws_floats, _stream())
    if defer_mean:      # `loss` is filled by howl_head_bwd's HowlCtcMean rider
        return loss, dlogits, nll
    return loss, dlogits


class _CtcLoss(torch.autograd.Function):
    """loss = CTCLoss(blank)(log_softmax(scores, -1), targets, input_lengths, target_lengths), reduction "mean": the fused
    kernel returns the loss and d loss / d scores in one pass (``howl_ctc_loss``)."""

    @staticmethod
    def forward(ctx, scores, targets, input_lengths, target_lengths, blank, max_target):
        loss, dlogits = _ctc_launch(scores, targets, input_lengths, target_lengths, blank, max_target,
                                    ctx.needs_input_grad[0])
        ctx.save_for_backward(dlogits)
        return loss

    @staticmethod
    def backward(ctx, grad_out):
        (dlogits,) = ctx.saved_tensors
        return dlogits * grad_out, None, None, None, None, None


def _ctc_args(scores, targets, input_lengths, target_lengths, max_target):
    if not on_device(scores):
        raise _lib.HowlHipError("ctc_loss: scores must be on a HIP device (no CPU fallback)")
    if scores.dim() != 3 or targets.dim() != 2:
        raise ValueError("ctc_loss: scores must be (T, B, C) and targets the padded (B, L) matrix")
    if max_target is None:
        max_target = int(target_lengths.max()) if target_lengths.numel() else 0
    if max_target > targets.shape[1]:
        raise ValueError("ctc_loss: a target length exceeds the width of the target matrix")
    dev = scores.device
    targets = targets.to(dev, torch.int64)
    if targets.shape[1] > 0 and targets.stride(1) != 1:
        targets = targets.contiguous()
    return (targets, input_lengths.to(dev, torch.int64).contiguous(), target_lengths.to(dev, torch.int64).contiguous(),
            max_target)


def ctc_supported(T: int, C: int, max_target: int) -> bool:
    return bool(_lib.get().cdll.howl_ctc_supported(int(T), int(C), int(max_target)))


def ctc_loss_fwd_bwd(scores, targets, input_lengths, target_lengths, blank: int, max_target=None, defer_mean=False):
    """Loss and d loss / d scores without an autograd graph (training.fused.FusedTrainer.step_sequence): same kernel as
    ``ctc_loss``; the batch must be inside the kernel's range.  ``defer_mean``: returns (loss, dscores, nll, target_lengths)
    with ``loss`` still unwritten -- the batch mean is left to ``howl_head_bwd`` (``SequentialLstm._launch_backward(...,
    ctc_mean=(nll, target_lengths, loss))``), one launch fewer."""
    targets, input_lengths, target_lengths, max_target = _ctc_args(scores, targets, input_lengths, target_lengths, max_target)
    if scores.stride(2) != 1 or scores.dtype != torch.float32:
        raise ValueError("ctc_loss_fwd_bwd: scores must be fp32 with unit stride over the classes")
    out = _ctc_launch(scores, targets, input_lengths, target_lengths, int(blank), max_target, True, defer_mean)
    return out + (target_lengths,) if defer_mean else out


def ctc_loss(scores, targets, input_lengths, target_lengths, blank: int):
    """``nn.CTCLoss(blank)(torch.log_softmax(scores, -1), targets, input_lengths, target_lengths)`` of the reference's training
    loop (train.py:250-256, 291-296) for ``scores`` of shape (T, B, C) on the device, as ONE fused kernel.  ``targets`` is the
    padded (B, L) int64 matrix; the two length vectors may live on the host (as in the reference) or on the device.  Whole
    clips are inside the kernel's range (T <= 8192 frames, walked in 128-frame windows); a batch outside it (C > 64, a
    target longer than 31 labels, scores that are not fp32) raises -- the training path has no vendor fallback."""
    targets_d, in_d, tl_d, max_target = _ctc_args(scores, targets, input_lengths, target_lengths, None)
    T, B, C = scores.shape
    if scores.dtype != torch.float32:
        raise _lib.HowlHipError(f"ctc_loss: scores must be fp32 (got {scores.dtype})")
    if not _lib.get().cdll.howl_ctc_supported(T, C, max_target):
        raise _lib.HowlHipError(f"ctc_loss: T={T} frames, C={C} classes, targets of {max_target} labels: outside howl_ctc_loss's "
                                "range (T <= 8192, C <= 64, targets <= 31)")
    if scores.stride(2) != 1:
        scores = scores.contiguous()
    return _CtcLoss.apply(scores, targets_d, in_d, tl_d, int(blank), max_target)


def adamw_step(p, g, m, v, lr, betas, eps, weight_decay, step, grad_scale=1.0):
    _lib.get().call("howl_adamw_step", _p(p), _p(g), _p(m), _p(v), p.numel(), lr, betas[0], betas[1], eps,
                    weight_decay, step, grad_scale, _stream())
