"""``SequentialLstm`` ("seq-lstm") and ``SimpleLstm`` ("lstm") of ``howl/model/rnn.py:41-91`` on MI355X kernels
(``howl_amd/csrc/lstm.hip``): packed-sequence LSTM(40 -> 128) + Linear(128,256)-ReLU-Linear(256,C).

Same construction order / ``state_dict`` keys (``lstm.weight_ih_l0 ...``, ``dnn.0.*``, ``dnn.2.*``), same call protocol
``model(x: (B, C>=1, M, T), lengths)`` and streaming-state protocol as the reference; ``nn.LSTM`` / ``nn.Linear`` are
parameter containers only.  Like ``pack_padded_sequence`` the lengths must be sorted in decreasing order.
"""
import ctypes
from typing import Any

import torch
import torch.nn as nn

from howl_amd import lib as _lib
from howl_amd import ops
from howl_amd.settings import _EnvSettings

from .base import RegisteredModel

__all__ = ["LstmConfig", "SequentialLstm", "SimpleLstm"]

HID = 128


class LstmConfig(_EnvSettings):
    num_mels: int = 40
    hidden_size: int = 128
    num_labels: int = 2


def _vp(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _x_frames(x):
    """Frames per utterance of the buffer behind x (B,T,M): a leading-frames view of a longer (B,T_x,M) feature buffer is used in
    place (``HowlLstmSaved.x_frames``)."""
    M = x.shape[2]
    # (the stride of a one-frame time axis is never used to address anything -- and torch leaves it arbitrary, e.g. 1 after
    # permute(0, 2, 1).contiguous() of a (B, M, 1) tensor)
    if x.stride(2) != 1 or (x.shape[1] > 1 and x.stride(1) != M) or x.stride(0) % M or x.stride(0) < x.shape[1] * M:
        raise ValueError("LSTM input must be a (B,T,M) tensor with contiguous rows and a whole number of frames per utterance")
    return x.stride(0) // M


def _lstm_forward_raw(x, lengths, t_out, h0, c0, w_ih, w_hh, b_ih, b_hh, next_logmel=None):
    """``howl_lstm_fwd`` on x (B,T,M), contiguous or the leading frames of a longer contiguous buffer: returns (hs (B,t_out,128)
    view, hT, cT, saved buffers for the backward).  ``next_logmel``: a ``HowlLogmelArgs`` record -- the frontend of the NEXT batch
    runs with this call (``howl_lstm_fwd_next``: as rider blocks of the recurrence's launch where that leaves CUs idle)."""
    B, T, M = x.shape
    dev = x.device
    f32 = dict(dtype=torch.float32, device=dev)
    bufs = dict(gates=torch.empty((B, T, 4 * HID), **f32), c=torch.empty((B, T, HID), **f32),
                hseq=torch.empty((B, T + 1, HID), **f32))
    ws = torch.empty(_lib.get().cdll.howl_lstm_workspace_bytes(B, T), dtype=torch.uint8, device=dev)
    hT, cT = torch.empty((B, HID), **f32), torch.empty((B, HID), **f32)
    prm = _lib.HowlLstmParams(_vp(w_ih), _vp(w_hh), _vp(b_ih), _vp(b_hh))
    xf = _x_frames(x)
    # the (B, T, 512) projection buffer only where the library runs the projection GEMM (40 MB at 512 x 38 otherwise unused)
    gx = torch.empty((B, T, 4 * HID), **f32) if _lib.get().cdll.howl_lstm_needs_gx(ctypes.byref(prm), B, T, M, xf) else None
    sv = _lib.HowlLstmSaved(_vp(gx), _vp(bufs["gates"]), _vp(bufs["c"]), _vp(bufs["hseq"]), None, t_out, xf)
    if next_logmel is not None:
        _lib.get().call("howl_lstm_fwd_next", ctypes.byref(prm), _vp(x), B, T, M, _vp(lengths), _vp(h0), _vp(c0), ctypes.byref(sv),
                        _vp(hT), _vp(cT), _vp(ws), ws.numel(), ctypes.byref(next_logmel), ops._stream())
    else:
        _lib.get().call("howl_lstm_fwd", ctypes.byref(prm), _vp(x), B, T, M, _vp(lengths), _vp(h0), _vp(c0), ctypes.byref(sv),
                        _vp(hT), _vp(cT), _vp(ws), ws.numel(), ops._stream())
    saved = (x, lengths, c0, w_ih, w_hh, b_ih, b_hh, bufs["gates"], bufs["c"], bufs["hseq"], ws)
    return bufs["hseq"][:, 1:t_out + 1], hT, cT, saved


def _lstm_backward_raw(saved, t_out, d_hs, d_hT, d_cT, grads=None):
    """``howl_lstm_bwd``: gradients of (w_ih, w_hh, b_ih, b_hh), written into ``grads`` when given (flat-buffer views)."""
    x, lengths, c0, w_ih, w_hh, b_ih, b_hh, gates, cs, hseq, ws = saved
    B, T, M = x.shape
    dy = None
    if d_hs is not None:
        if t_out == T and d_hs.is_contiguous():
            dy = d_hs
        elif t_out == T:
            dy = d_hs.contiguous()
        else:
            dy = torch.zeros((B, T, HID), dtype=torch.float32, device=x.device)
            dy[:, :t_out].copy_(d_hs)
    d_hT = None if d_hT is None else d_hT.contiguous()
    d_cT = None if d_cT is None else d_cT.contiguous()
    dgates = torch.empty((B, T, 4 * HID), dtype=torch.float32, device=x.device)
    if grads is None:
        grads = [torch.empty_like(p) for p in (w_ih, w_hh, b_ih, b_hh)]
    prm = _lib.HowlLstmParams(_vp(w_ih), _vp(w_hh), _vp(b_ih), _vp(b_hh))
    sv = _lib.HowlLstmSaved(None, _vp(gates), _vp(cs), _vp(hseq), _vp(dgates), t_out, _x_frames(x))
    gr = _lib.HowlLstmGrads(*[_vp(g) for g in grads])
    _lib.get().call("howl_lstm_bwd", ctypes.byref(prm), _vp(x), B, T, M, _vp(lengths), _vp(c0), ctypes.byref(sv), _vp(dy),
                    _vp(d_hT), _vp(d_cT), ctypes.byref(gr), _vp(ws), ws.numel(), ops._stream())
    return grads


class _LstmFunction(torch.autograd.Function):
    """x (B,T,M) contiguous, lengths (B) int64 on the device or None -> (hs (B,t_out,128) view, hT (B,128), cT (B,128))."""

    @staticmethod
    def forward(ctx, x, lengths, t_out, h0, c0, w_ih, w_hh, b_ih, b_hh):
        hs, hT, cT, saved = _lstm_forward_raw(x, lengths, t_out, h0, c0, w_ih, w_hh, b_ih, b_hh)
        ctx.save_for_backward(*saved)
        ctx.t_out = t_out
        return hs, hT, cT

    @staticmethod
    def backward(ctx, d_hs, d_hT, d_cT):
        grads = _lstm_backward_raw(ctx.saved_tensors, ctx.t_out, d_hs, d_hT, d_cT)
        return (None, None, None, None, None) + tuple(grads)


def _row_geom(x):
    if x.dim() == 3:
        outer, inner = x.shape[0], x.shape[1]
        s_outer, s_inner = x.stride(0), x.stride(1)
    else:
        outer, inner, s_outer, s_inner = 1, x.shape[0], 0, x.stride(0)
    if x.stride(-1) != 1:
        raise ValueError("head: features must be unit-stride")
    return inner, s_outer, s_inner, outer * inner


def _head_forward_raw(x, w1, b1, w2, b2):
    """``howl_head_fwd``: (y1 = relu(x W1^T + b1), y2 = y1 W2^T + b2) for x (..., n_in) with unit-stride features."""
    n_hid, n_in = w1.shape
    n_out = w2.shape[0]
    inner, s_outer, s_inner, rows = _row_geom(x)
    f32 = dict(dtype=torch.float32, device=x.device)
    y1 = torch.empty(x.shape[:-1] + (n_hid,), **f32)
    y2 = torch.empty(x.shape[:-1] + (n_out,), **f32)
    prm = _lib.HowlHeadParams(_vp(w1), _vp(b1), _vp(w2), _vp(b2))
    _lib.get().call("howl_head_fwd", ctypes.byref(prm), _vp(x), inner, s_outer, s_inner, rows, n_in, n_hid, n_out, _vp(y1),
                    _vp(y2), ops._stream())
    return y1, y2


def _seq_head_ctc_supported(B, T, w1, w2, max_target):
    n_hid, n_in = w1.shape
    return bool(_lib.get().cdll.howl_seq_head_ctc_supported(int(B), int(T), n_in, n_hid, w2.shape[0], int(max_target)))


def _seq_head_ctc_raw(hs, w1, b1, w2, b2, targets, input_lengths, target_lengths, blank, max_target):
    """``howl_seq_head_ctc``: head forward + log_softmax / CTC + the head's backward over the rows in one launch.  hs (B, T, 128)
    view of the hidden states -> (y2 (B, T, n_out), nll (B,), dz1, dhs, head workspace holding the partial sums)."""
    n_hid, n_in = w1.shape
    n_out = w2.shape[0]
    B, T = hs.shape[:2]
    f32 = dict(dtype=torch.float32, device=hs.device)
    y2 = torch.empty((B, T, n_out), **f32)
    nll = torch.empty(B, **f32)
    dz1 = torch.empty((B, T, n_hid), **f32)
    dhs = torch.empty((B, T, n_in), **f32)
    head_ws = torch.empty(_lib.get().cdll.howl_head_workspace_bytes(n_in, n_hid, n_out), dtype=torch.uint8, device=hs.device)
    prm = _lib.HowlHeadParams(_vp(w1), _vp(b1), _vp(w2), _vp(b2))
    _lib.get().call("howl_seq_head_ctc", ctypes.byref(prm), _vp(hs), hs.stride(0), hs.stride(1), B, T, n_in, n_hid, n_out,
                    _vp(targets), targets.stride(0), int(max_target), _vp(input_lengths), _vp(target_lengths), int(blank), _vp(y2),
                    _vp(nll), _vp(dz1), _vp(dhs), _vp(head_ws), head_ws.numel(), ops._stream())
    return y2, nll, dz1, dhs, head_ws


def _head_backward_raw(x, y1, dy2, w1, b1, w2, b2, need_dx=True, grads=None, ctc_mean=None):
    """``howl_head_bwd``: dy2 (..., n_out) -> (dx or None, [dW1, db1, dW2, db2]), written into ``grads`` when given.
    ``ctc_mean`` = (nll, target_lengths, loss): the batch mean of a CTC loss taken by the same launch (``HowlCtcMean``)."""
    n_hid, n_in = w1.shape
    n_out = w2.shape[0]
    inner, s_outer, s_inner, rows = _row_geom(x)
    f32 = dict(dtype=torch.float32, device=x.device)
    dy2 = dy2.contiguous()
    dz1 = torch.empty_like(y1)
    dx = torch.empty(x.shape, **f32) if need_dx else None
    if grads is None:
        grads = [torch.empty_like(t) for t in (w1, b1, w2, b2)]
    ws = torch.empty(_lib.get().cdll.howl_head_workspace_bytes(n_in, n_hid, n_out), dtype=torch.uint8, device=x.device)
    prm = _lib.HowlHeadParams(_vp(w1), _vp(b1), _vp(w2), _vp(b2))
    gr = _lib.HowlHeadGrads(*[_vp(g) for g in grads])
    cm = None
    if ctc_mean is not None:
        nll, tl, loss = ctc_mean
        cm = ctypes.byref(_lib.HowlCtcMean(_vp(nll), _vp(tl), int(nll.numel()), _vp(loss)))
    _lib.get().call("howl_head_bwd", ctypes.byref(prm), _vp(x), inner, s_outer, s_inner, rows, n_in, n_hid, n_out, _vp(y1),
                    _vp(dy2), _vp(dz1), _vp(dx), ctypes.byref(gr), cm, _vp(ws), ws.numel(), ops._stream())
    return dx, grads


def _seq_backward_raw(saved, y1, dy2, head_params, grads, ctc_mean=None, adamw=None, head_rows_done=None):
    """``howl_seq_lstm_bwd``: head + LSTM backward of the sequence model in one call (t_out == T); ``grads`` = the eight flat-buffer
    views in ``hot_parameters()`` order.  ``adamw`` = (flat params, flat grads, m, v, lr, (beta1, beta2), eps, weight_decay, step,
    grad_scale): the optimiser step is part of the call (``HowlAdamW``).  ``head_rows_done`` = (dz1, dhs, head_ws) left by
    ``howl_seq_head_ctc`` (then ``y1`` / ``dy2`` are None: the call folds that launch's partial sums instead of running the rows)."""
    x, lengths, c0, w_ih, w_hh, b_ih, b_hh, gates, cs, hseq, ws = saved
    w1, b1, w2, b2 = head_params
    B, T, M = x.shape
    n_hid, n_in = w1.shape
    n_out = w2.shape[0]
    f32 = dict(dtype=torch.float32, device=x.device)
    dgates = torch.empty((B, T, 4 * HID), **f32)
    if head_rows_done is not None:
        dz1, dhs, head_ws = head_rows_done
        y1 = dy2 = None
    else:
        dy2 = dy2.contiguous()
        dz1 = torch.empty_like(y1)
        dhs = torch.empty((B, T, HID), **f32)
        head_ws = torch.empty(_lib.get().cdll.howl_head_workspace_bytes(n_in, n_hid, n_out), dtype=torch.uint8, device=x.device)
    hp = _lib.HowlHeadParams(_vp(w1), _vp(b1), _vp(w2), _vp(b2))
    hg = _lib.HowlHeadGrads(*[_vp(g) for g in grads[4:8]])
    cm = None
    if ctc_mean is not None:
        nll, tl, loss = ctc_mean
        cm = ctypes.byref(_lib.HowlCtcMean(_vp(nll), _vp(tl), int(nll.numel()), _vp(loss)))
    prm = _lib.HowlLstmParams(_vp(w_ih), _vp(w_hh), _vp(b_ih), _vp(b_hh))
    sv = _lib.HowlLstmSaved(None, _vp(gates), _vp(cs), _vp(hseq), _vp(dgates), T, _x_frames(x))
    gr = _lib.HowlLstmGrads(*[_vp(g) for g in grads[:4]])
    opt = None
    if adamw is not None:
        flat, fgrad, m, v, lr, betas, eps, wd, step, gscale = adamw
        opt = ctypes.byref(_lib.HowlAdamW(_vp(flat), _vp(fgrad), _vp(m), _vp(v), flat.numel(), lr, betas[0], betas[1], eps, wd, step, gscale))
    _lib.get().call("howl_seq_lstm_bwd", ctypes.byref(hp), n_hid, n_out, _vp(y1), _vp(dy2), _vp(dz1), _vp(dhs), ctypes.byref(hg), cm,
                    _vp(head_ws), head_ws.numel(), ctypes.byref(prm), _vp(x), B, T, M, _vp(lengths), _vp(c0), ctypes.byref(sv),
                    ctypes.byref(gr), _vp(ws), ws.numel(), opt, ops._stream())


class _HeadFunction(torch.autograd.Function):
    """Linear - ReLU - Linear (``self.dnn``, rnn.py:44-48) on the library's head kernels."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2):
        y1, y2 = _head_forward_raw(x, w1, b1, w2, b2)
        ctx.save_for_backward(x, y1, w1, b1, w2, b2)
        return y2

    @staticmethod
    def backward(ctx, dy2):
        x, y1, w1, b1, w2, b2 = ctx.saved_tensors
        dx, grads = _head_backward_raw(x, y1, dy2, w1, b1, w2, b2, ctx.needs_input_grad[0])
        return (dx,) + tuple(grads)


class _LstmBase(RegisteredModel):
    def __init__(self, num_labels: int, config: LstmConfig = None):
        super().__init__(num_labels)
        config = config or LstmConfig()
        if config.hidden_size != HID:
            raise NotImplementedError("the MI355X LSTM kernels are specialised for hidden_size=128")
        self.lstm = nn.LSTM(config.num_mels, config.hidden_size)
        self.dnn = nn.Sequential(nn.Linear(config.hidden_size, int(2 * config.hidden_size)), nn.ReLU(),
                                 nn.Linear(int(2 * config.hidden_size), num_labels))
        self.hc = None

    def _lstm_inputs(self, x, lengths, t_out=None):
        """-> (xb (B, t_out, M) contiguous, device lengths or None, t_out, h0, c0).

        ``lengths`` follows ``pack_padded_sequence``: one entry per sequence, sorted in decreasing order, 1..T.  A host tensor
        (the reference's batches carry them on the host) is checked here.  A DEVICE tensor together with ``t_out`` = its
        maximum is trusted as already checked -- reading it back would stall the stream the step is queued on."""
        x0 = x[:, 0]                                   # (B, M, T), log-mels only (rnn.py:61,86)
        if not ops.on_device(x0):
            raise _lib.HowlHipError("LSTM input must be on a HIP device (no CPU fallback)")
        xb = x0.permute(0, 2, 1)
        B, T, _ = xb.shape
        if lengths is not None and t_out is not None and ops.on_device(lengths):
            if lengths.numel() != B or not 1 <= t_out <= T:
                raise RuntimeError("lengths must have one entry per sequence and t_out must be in 1..T")
            lengths = lengths.to(torch.int64)
        elif lengths is not None:
            lc = lengths.detach().cpu().long()
            if lc.numel() != B:
                raise RuntimeError("lengths must have one entry per sequence")
            if (lc[:-1] < lc[1:]).any():
                raise RuntimeError("`lengths` array must be sorted in decreasing order (pack_padded_sequence semantics)")
            if lc.min() <= 0 or lc.max() > T:
                raise RuntimeError("lengths must be in 1..T")
            t_out = int(lc.max())
            lengths = lc.to(xb.device)
        else:
            t_out = T
        if t_out < T:                                  # frames no sequence reaches: the kernels and the saved buffers only see
            xb = xb[:, :t_out]                         # t_out steps (a view: the library takes the buffer's frame stride)
        M = xb.shape[2]
        if (xb.stride(2) != 1 or (xb.shape[1] > 1 and xb.stride(1) != M) or xb.stride(0) % M       # the fused frontend already hands over a (B,T,M) buffer;
                or xb.stride(0) < xb.shape[1] * M):                            # expanded / overlapping batch views are copied
            xb = xb.contiguous()
        hx = self.streaming_state if self.is_streaming and self.streaming_state is not None else None
        if hx is not None and (tuple(hx[0].shape) != (1, B, HID) or tuple(hx[1].shape) != (1, B, HID)):
            raise RuntimeError(f"Expected hidden size (1, {B}, {HID}), got {tuple(hx[0].shape)}")     # as nn.LSTM does
        h0 = hx[0][0].contiguous() if hx is not None else None
        c0 = hx[1][0].contiguous() if hx is not None else None
        return xb, lengths, t_out, h0, c0

    def _run_lstm(self, x, lengths):
        xb, lengths, t_out, h0, c0 = self._lstm_inputs(x, lengths)
        l = self.lstm
        return _LstmFunction.apply(xb, lengths, t_out, h0, c0, l.weight_ih_l0, l.weight_hh_l0, l.bias_ih_l0, l.bias_hh_l0)

    def hot_parameters(self):
        """Parameters in the order the fused trainer lays them out (and ``_launch_backward`` fills their gradients)."""
        l = self.lstm
        return [l.weight_ih_l0, l.weight_hh_l0, l.bias_ih_l0, l.bias_hh_l0, self.dnn[0].weight, self.dnn[0].bias,
                self.dnn[2].weight, self.dnn[2].bias]

    def _head(self, h):
        return _HeadFunction.apply(h, self.dnn[0].weight, self.dnn[0].bias, self.dnn[2].weight, self.dnn[2].bias)


class SequentialLstm(_LstmBase, name="seq-lstm"):
    @property
    def streaming_state(self) -> Any:
        return self.hc

    @streaming_state.setter
    def streaming_state(self, x: Any):
        self.hc = x

    def forward(self, x, lengths):
        hs, hT, cT = self._run_lstm(x, lengths)
        if self.is_streaming:
            self.streaming_state = (hT.detach().clone().unsqueeze(0), cT.detach().clone().unsqueeze(0))
        return self._head(hs).permute(1, 0, 2)         # (T_len, B, num_labels), as dnn(rnn_seq) in rnn.py:71

    # --- training.fused.FusedTrainer hooks: the same launches as forward() / autograd, without the autograd graph -----
    TAKES_NEXT_LOGMEL = True      # FusedTrainer.step_sequence(next_audio=...): the next batch's frontend rides in the forward call
    TAKES_CTC = True              # ... and _launch_forward(ctc=...): head + CTC + head backward rows as one launch where covered

    def _launch_forward(self, feat, lengths, t_out=None, next_logmel=None, ctc=None):
        """feat (B, C>=1, M, T) -> scores (T_len, B, num_labels) view; keeps what ``_launch_backward`` needs.  ``t_out`` with
        device-resident ``lengths``: see ``_lstm_inputs``.  ``next_logmel``: see ``_lstm_forward_raw``.
        ``ctc`` = (targets (B, Lmax) int64, target_lengths, blank, max_target), all on the device: where the library's fused launch
        covers the batch (``howl_seq_head_ctc``: short windows, >= 2048 rows) the head, log_softmax + CTC and the head's backward
        over the rows run as ONE launch behind the recurrence; ``self.ctc_nll`` is then the (B,) negative log likelihoods and
        ``_launch_backward`` takes ``dscores=None``.  Otherwise ``self.ctc_nll`` is None and the caller runs the loss itself."""
        xb, lengths, t_out, h0, c0 = self._lstm_inputs(feat, lengths, t_out)
        ps = self.hot_parameters()
        hs, hT, cT, saved = _lstm_forward_raw(xb, lengths, t_out, h0, c0, *ps[:4], next_logmel=next_logmel)
        self.ctc_nll = None
        if self.is_streaming:                          # same carry as forward() (rnn.py:64-68)
            self.streaming_state = (hT.detach().clone().unsqueeze(0), cT.detach().clone().unsqueeze(0))
        if (ctc is not None and lengths is not None and t_out == xb.shape[1]
                and _seq_head_ctc_supported(xb.shape[0], t_out, ps[4], ps[6], ctc[3])):
            targets, target_lengths, blank, max_target = ctc
            y2, nll, dz1, dhs, head_ws = _seq_head_ctc_raw(hs, *ps[4:8], targets, lengths, target_lengths, blank, max_target)
            self._seq_saved = (saved, t_out, hs, None, (dz1, dhs, head_ws))
            self.ctc_nll = nll
            return y2.permute(1, 0, 2)
        y1, y2 = _head_forward_raw(hs, *ps[4:8])
        self._seq_saved = (saved, t_out, hs, y1, None)
        return y2.permute(1, 0, 2)

    def _launch_backward(self, dscores, out_grads=None, ctc_mean=None, adamw=None):
        """dscores: d loss / d scores as a (T_len, B, num_labels) view of a (B, T_len, num_labels) buffer (ops.ctc_loss_fwd_bwd);
        ``ctc_mean`` = (nll, target_lengths, loss) when the loss launch left its batch mean to the head's backward;
        ``adamw``: see ``_seq_backward_raw`` -- ``self.optimizer_step_done`` says whether the call took it."""
        saved, t_out, hs, y1, rows_done = self._seq_saved
        ps = self.hot_parameters()
        grads = out_grads if out_grads is not None else [torch.empty_like(p) for p in ps]
        x = saved[0]
        self.optimizer_step_done = False
        if rows_done is not None:    # howl_seq_head_ctc ran the head's rows behind the forward recurrence: fold, dW1, LSTM backward
            _seq_backward_raw(saved, None, None, ps[4:8], grads, ctc_mean, adamw, head_rows_done=rows_done)
            self.optimizer_step_done = adamw is not None
        elif t_out == x.shape[1]:      # whole buffer ran: one call, the wide weight gradients and the slab folds merged (12 launches)
            _seq_backward_raw(saved, y1, dscores.permute(1, 0, 2), ps[4:8], grads, ctc_mean, adamw)
            self.optimizer_step_done = adamw is not None
        else:
            dhs, _ = _head_backward_raw(hs, y1, dscores.permute(1, 0, 2), *ps[4:8], True, grads[4:8], ctc_mean)
            _lstm_backward_raw(saved, t_out, dhs, None, None, grads[:4])
        self._seq_saved = None
        return grads


class SimpleLstm(_LstmBase, name="lstm"):
    NEEDS_LENGTHS = True       # FusedTrainer.step passes the frame lengths through

    # --- training.fused.FusedTrainer hooks (cross-entropy on the last hidden state, pretrain_gsc.py:126-133) ---------
    def _launch_forward(self, feat, lengths, t_out=None):
        """feat (B, C>=1, M, T), frame lengths -> logits (B, num_labels); keeps what ``_launch_backward`` needs."""
        if lengths is None:
            raise TypeError("SimpleLstm needs lengths (rnn.py:88 packs unconditionally)")
        xb, lengths, t_out, h0, c0 = self._lstm_inputs(feat, lengths, t_out)
        ps = self.hot_parameters()
        _, hT, _, saved = _lstm_forward_raw(xb, lengths, t_out, h0, c0, *ps[:4])
        y1, y2 = _head_forward_raw(hT, *ps[4:8])
        self._cls_saved = (saved, t_out, hT, y1)
        return y2

    def _launch_backward(self, feat, dlogits, out_grads=None):
        saved, t_out, hT, y1 = self._cls_saved
        ps = self.hot_parameters()
        grads = out_grads if out_grads is not None else [torch.empty_like(p) for p in ps]
        dhT, _ = _head_backward_raw(hT, y1, dlogits, *ps[4:8], True, grads[4:8])
        _lstm_backward_raw(saved, t_out, None, dhT, None, grads[:4])
        self._cls_saved = None
        return grads

    def forward(self, x, lengths):
        if lengths is None:
            raise TypeError("SimpleLstm needs lengths (rnn.py:88 packs unconditionally)")
        hs, hT, cT = self._run_lstm(x, lengths)
        if self.is_streaming:
            # rnn.py:89-90 assigns streaming_state, which the base class setter drops (base.py:35-37): stays stateless
            self.streaming_state = (hT.unsqueeze(0), cT.unsqueeze(0))
        return self._head(hT)
