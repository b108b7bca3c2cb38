"""LSTM recurrence / BPTT / classifier-head kernels on the hipemu CPU emulator vs the oracle (and golden G6)."""
import ctypes

import numpy as np
import pytest
import torch

from emu_util import emu_lib, ptr
from howl_amd.lib import (FB_PACKED_FLOATS, HowlAdamW, HowlHeadGrads, HowlHeadParams, HowlLogmelArgs, HowlLstmGrads, HowlLstmParams,
                          HowlLstmSaved)
from oracle import models as om


@pytest.fixture(scope="module")
def lib():
    return emu_lib()


def run_lstm(lib, sd, x_btm, lengths, h0=None, c0=None, x_frames=0):
    """x_frames > T: x_btm is the flat (B, x_frames, M) buffer whose first T frames per utterance are the input."""
    B, T, M = x_btm.shape
    if x_frames:
        assert x_btm.shape[1] == x_frames and lengths is not None
        T = int(lengths.max())
    npz = {k: np.ascontiguousarray(sd["lstm." + k].numpy()) for k in ("weight_ih_l0", "weight_hh_l0", "bias_ih_l0", "bias_hh_l0")}
    prm = HowlLstmParams(ptr(npz["weight_ih_l0"]), ptr(npz["weight_hh_l0"]), ptr(npz["bias_ih_l0"]), ptr(npz["bias_hh_l0"]))
    bufs = dict(gx=np.zeros((B, T, 512), np.float32), gates=np.zeros((B, T, 512), np.float32),
                c=np.zeros((B, T, 128), np.float32), hseq=np.full((B, T + 1, 128), np.nan, np.float32),
                dgates=np.full((B, T, 512), np.nan, np.float32))   # rows t >= t_out must never be read
    t_out = int(lengths.max()) if lengths is not None else T
    sv = HowlLstmSaved(ptr(bufs["gx"]), ptr(bufs["gates"]), ptr(bufs["c"]), ptr(bufs["hseq"]), ptr(bufs["dgates"]), t_out, x_frames)
    hT, cT = np.zeros((B, 128), np.float32), np.zeros((B, 128), np.float32)
    ws = np.zeros(lib.cdll.howl_lstm_workspace_bytes(B, T), np.uint8)
    x = np.ascontiguousarray(x_btm, np.float32)
    ln = None if lengths is None else np.ascontiguousarray(lengths, np.int64)
    lib.call("howl_lstm_fwd", ctypes.byref(prm), ptr(x), B, T, M, ptr(ln), ptr(h0), ptr(c0), ctypes.byref(sv), ptr(hT), ptr(cT),
             ptr(ws), ws.size, None)
    keep = dict(prm=prm, sv=sv, ws=ws, x=x, ln=ln, npz=npz, bufs=bufs, t_out=t_out)
    return bufs["hseq"][:, 1:t_out + 1].copy(), hT, cT, keep


@pytest.mark.parametrize("rows", ["4", "16"])
def test_lstm_forward_backward_ragged(lib, golden, monkeypatch, rows):
    monkeypatch.setenv("HOWL_LSTM_ROWS", rows)     # both recurrence pairs: 4x4x1_16b (4 sequences / workgroup) and 16x16x4
    if rows == "4":      # ... and both GEMMs for the projections: weights-stationary row streaming (default from 2048 rows) / 64 x 64 tiles
        monkeypatch.setenv("HOWL_ROWGEMM_MIN_ROWS", "1")
    g = golden("g6_seq_lstm")
    g2, g4 = golden("g2_frontend_gsc"), golden("g4_zmuv")
    lengths = g["frame_lengths"].astype(np.int64)          # descending, ragged: [78, 78, 78, 78, 69, 62]
    x4 = torch.from_numpy((g2["feats"] - g4["mean"]) / g4["std"])          # (6,3,40,81)
    x_btm = x4[:, 0].permute(0, 2, 1).contiguous().numpy()                 # (6,81,40)
    sd = om.lstm_init(5)
    hs, hT, cT, keep = run_lstm(lib, sd, x_btm, lengths)
    # oracle (explicit cell loop, packed-sequence semantics)
    xr = x4[:, 0].permute(2, 0, 1).contiguous()
    p = {k: v.clone().requires_grad_(True) for k, v in sd.items() if k.startswith("lstm.")}
    seq, (h_ref, c_ref) = om._lstm_cell_seq(p, xr, torch.from_numpy(lengths), None)
    np.testing.assert_allclose(hs, seq.detach().permute(1, 0, 2).numpy(), rtol=0, atol=2e-6)
    np.testing.assert_allclose(hT, h_ref[0].detach().numpy(), rtol=0, atol=2e-6)
    np.testing.assert_allclose(cT, c_ref[0].detach().numpy(), rtol=0, atol=2e-6)
    # seq-lstm logits through the head kernels vs the reference golden
    B, T = x_btm.shape[:2]
    t_out = keep["t_out"]
    w1, b1, w2, b2 = (np.ascontiguousarray(sd[k].numpy()) for k in ("dnn.0.weight", "dnn.0.bias", "dnn.2.weight", "dnn.2.bias"))
    hid = np.zeros((B * t_out, 256), np.float32)
    hseq = keep["bufs"]["hseq"]
    h1 = np.ascontiguousarray(hseq.reshape(-1)[128:])   # skip hseq[0][0] (h0): row (b,t) -> hseq[b][t+1]
    logits = np.zeros((B * t_out, 5), np.float32)
    hp = HowlHeadParams(ptr(w1), ptr(b1), ptr(w2), ptr(b2))
    lib.call("howl_head_fwd", ctypes.byref(hp), ptr(h1), t_out, (T + 1) * 128, 128, B * t_out, 128, 256, 5, ptr(hid), ptr(logits),
             None)
    np.testing.assert_allclose(logits.reshape(B, t_out, 5).transpose(1, 0, 2), g["logits"], rtol=0, atol=5e-6)

    # BPTT: random output gradient + final-state gradients
    rng = np.random.default_rng(0)
    dy = np.zeros((B, T, 128), np.float32)
    dy[:, :t_out] = rng.standard_normal((B, t_out, 128)).astype(np.float32)
    dhT = rng.standard_normal((B, 128)).astype(np.float32)
    dcT = rng.standard_normal((B, 128)).astype(np.float32)
    gr = {k: np.full_like(v, np.nan) for k, v in keep["npz"].items()}
    grads = HowlLstmGrads(ptr(gr["weight_ih_l0"]), ptr(gr["weight_hh_l0"]), ptr(gr["bias_ih_l0"]), ptr(gr["bias_hh_l0"]))
    lib.call("howl_lstm_bwd", ctypes.byref(keep["prm"]), ptr(keep["x"]), B, T, 40, ptr(keep["ln"]), None, ctypes.byref(keep["sv"]),
             ptr(dy), ptr(dhT), ptr(dcT), ctypes.byref(grads), ptr(keep["ws"]), keep["ws"].size, None)
    loss = (seq * torch.from_numpy(dy[:, :t_out]).permute(1, 0, 2)).sum() + (h_ref[0] * torch.from_numpy(dhT)).sum() + \
           (c_ref[0] * torch.from_numpy(dcT)).sum()
    loss.backward()
    for k in ("weight_ih_l0", "weight_hh_l0", "bias_ih_l0", "bias_hh_l0"):
        ref = p["lstm." + k].grad.numpy()
        np.testing.assert_allclose(gr[k], ref, rtol=0, atol=2e-5 * max(1.0, np.abs(ref).max()), err_msg=k)


@pytest.mark.parametrize("min_rows", ["1", "1000000"])
def test_lstm_input_inside_a_longer_feature_buffer(lib, monkeypatch, min_rows):
    """HowlLstmSaved.x_frames: the first T frames of a (B, T + 3, M) buffer used in place == the same frames copied out
    (input projection on the weights-stationary kernel and on the 64 x 64 tiles)."""
    monkeypatch.setenv("HOWL_ROWGEMM_MIN_ROWS", min_rows)
    rng = np.random.default_rng(5)
    B, T, M = 5, 7, 40
    xbuf = rng.standard_normal((B, T + 3, M)).astype(np.float32)
    lengths = np.array([7, 7, 6, 4, 2], np.int64)
    sd = om.lstm_init(5)
    a, hTa, cTa, ka = run_lstm(lib, sd, np.ascontiguousarray(xbuf[:, :T]), lengths)
    b, hTb, cTb, kb = run_lstm(lib, sd, xbuf, lengths, x_frames=T + 3)
    np.testing.assert_array_equal(a, b)
    np.testing.assert_array_equal(hTa, hTb)
    dy = np.zeros((B, T, 128), np.float32)
    dy[:] = rng.standard_normal((B, T, 128)).astype(np.float32)
    out = []
    for keep in (ka, kb):
        gr = {k: np.full_like(v, np.nan) for k, v in keep["npz"].items()}
        grads = HowlLstmGrads(ptr(gr["weight_ih_l0"]), ptr(gr["weight_hh_l0"]), ptr(gr["bias_ih_l0"]), ptr(gr["bias_hh_l0"]))
        lib.call("howl_lstm_bwd", ctypes.byref(keep["prm"]), ptr(keep["x"]), B, T, M, ptr(keep["ln"]), None, ctypes.byref(keep["sv"]),
                 ptr(dy), None, None, ctypes.byref(grads), ptr(keep["ws"]), keep["ws"].size, None)
        out.append(gr)
    for k in out[0]:
        np.testing.assert_array_equal(out[0][k], out[1][k], err_msg=k)


def test_lstm_streaming_carry_and_no_lengths(lib):
    rng = np.random.default_rng(1)
    x = rng.standard_normal((3, 9, 40)).astype(np.float32)
    sd = om.lstm_init(5)
    a, hT, cT, _ = run_lstm(lib, sd, x[:, :5], None)
    b, hT2, cT2, _ = run_lstm(lib, sd, x[:, 5:], None, h0=hT, c0=cT)      # carry (h, c) across calls (rnn.py:62,67-68)
    full, hTf, cTf, _ = run_lstm(lib, sd, x, None)
    np.testing.assert_allclose(np.concatenate([a, b], 1), full, rtol=0, atol=1e-6)
    np.testing.assert_allclose(hT2, hTf, rtol=0, atol=1e-6)


@pytest.mark.parametrize("rows,n_in,n_hid,n_out", [(77, 128, 256, 5), (1, 128, 256, 3), (300, 128, 256, 8), (41, 40, 64, 1),
                                                   (50, 128, 256, 12), (33, 64, 320, 4), (2100, 128, 256, 5)])
def test_head_forward_backward(lib, rows, n_in, n_hid, n_out):
    _head_case(lib, rows, n_in, n_hid, n_out)


@pytest.mark.parametrize("rows,n_out", [(77, 5), (16, 3), (1, 8), (300, 8), (531, 1)])
def test_head_row_streaming_kernels_on_small_shapes(lib, monkeypatch, rows, n_out):
    """The weights-stationary kernels of the 128 -> 256 -> n_out head (first layer + output layer in one launch; second layer's
    backward + ReLU mask + data gradient in one launch), which the library picks from 2048 rows: one tile per workgroup, partial
    last tiles, fewer tiles than a workgroup prefetches."""
    monkeypatch.setenv("HOWL_ROWGEMM_MIN_ROWS", "1")
    _head_case(lib, rows, 128, 256, n_out)


def _head_case(lib, rows, n_in, n_hid, n_out):
    """Linear - ReLU - Linear: the vector kernels (n_out <= 8, n_hid <= 256) and the GEMM path of the other shapes against
    numpy in float64; x rows through a two-level row map as the (B, T+1, 128) hidden-state buffer has; 2100 rows reach the
    wide weight-gradient kernel (128-row tiles, one block per CU) for the first layer."""
    rng = np.random.default_rng(2 + rows)
    inner = 7 if rows % 7 == 0 else rows
    outer = rows // inner
    xbuf = rng.standard_normal((outer, inner + 1, n_in)).astype(np.float32)      # row (o, i) lives at xbuf[o][i + 1]
    x = xbuf[:, 1:].reshape(rows, n_in)
    w1 = (rng.standard_normal((n_hid, n_in)) * 0.1).astype(np.float32)
    b1 = (rng.standard_normal(n_hid) * 0.1).astype(np.float32)
    w2 = (rng.standard_normal((n_out, n_hid)) * 0.1).astype(np.float32)
    b2 = (rng.standard_normal(n_out) * 0.1).astype(np.float32)
    dy2 = rng.standard_normal((rows, n_out)).astype(np.float32)
    y1, y2 = np.full((rows, n_hid), np.nan, np.float32), np.full((rows, n_out), np.nan, np.float32)
    hp = HowlHeadParams(ptr(w1), ptr(b1), ptr(w2), ptr(b2))
    x0 = np.ascontiguousarray(xbuf.reshape(-1)[n_in:])
    geom = (inner, (inner + 1) * n_in, n_in, rows, n_in, n_hid, n_out)
    lib.call("howl_head_fwd", ctypes.byref(hp), ptr(x0), *geom, ptr(y1), ptr(y2), None)
    z1 = x.astype(np.float64) @ w1.T.astype(np.float64) + b1
    r1 = np.maximum(z1, 0)
    np.testing.assert_allclose(y1, r1, rtol=0, atol=2e-5)
    np.testing.assert_allclose(y2, r1 @ w2.T.astype(np.float64) + b2, rtol=0, atol=2e-5)
    g = [np.full_like(a, np.nan) for a in (w1, b1, w2, b2)]
    gr = HowlHeadGrads(*[ptr(a) for a in g])
    dz1, dx = np.full((rows, n_hid), np.nan, np.float32), np.full((rows, n_in), np.nan, np.float32)
    ws = np.zeros(lib.cdll.howl_head_workspace_bytes(n_in, n_hid, n_out), np.uint8)
    lib.call("howl_head_bwd", ctypes.byref(hp), ptr(x0), *geom, ptr(y1), ptr(dy2), ptr(dz1), ptr(dx), ctypes.byref(gr), None,
             ptr(ws), ws.size, None)
    dz_ref = (dy2.astype(np.float64) @ w2) * (y1 > 0)
    scale = lambda a: max(1.0, np.abs(a).max())
    np.testing.assert_allclose(dz1, dz_ref, rtol=0, atol=2e-5)
    np.testing.assert_allclose(dx, dz_ref @ w1, rtol=0, atol=2e-5 * scale(dz_ref @ w1))
    for got, ref, name in ((g[0], dz_ref.T @ x, "dW1"), (g[1], dz_ref.sum(0), "db1"), (g[2], dy2.T.astype(np.float64) @ y1, "dW2"),
                           (g[3], dy2.sum(0, dtype=np.float64), "db2")):
        np.testing.assert_allclose(got, ref, rtol=0, atol=2e-5 * scale(ref), err_msg=name)
    # dx is optional (the first layer of a model that does not need its input's gradient)
    lib.call("howl_head_bwd", ctypes.byref(hp), ptr(x0), *geom, ptr(y1), ptr(dy2), ptr(dz1), None, ctypes.byref(gr), None,
             ptr(ws), ws.size, None)
    np.testing.assert_allclose(g[0], dz_ref.T @ x, rtol=0, atol=2e-5 * scale(dz_ref.T @ x))


def test_ctc_batch_mean_rides_in_the_head_backward(lib):
    """howl_ctc_loss(loss=NULL) leaves the batch mean to howl_head_bwd's HowlCtcMean rider: the same float as the loss launch's
    own mean, thin and GEMM head paths."""
    from howl_amd.lib import HowlCtcMean
    rng = np.random.default_rng(9)
    B, T, C = 7, 12, 5
    z = rng.standard_normal((B, T, C)).astype(np.float32)
    tg = np.ascontiguousarray(rng.integers(0, 4, (B, 3)), np.int64)
    il = np.full(B, T, np.int64)
    tl = np.array([3, 2, 3, 1, 0, 3, 2], np.int64)
    nll, loss_ref = np.zeros(B, np.float32), np.zeros(1, np.float32)
    lib.call("howl_ctc_loss", ptr(z), C, T * C, T, B, C, ptr(tg), 3, 3, ptr(il), ptr(tl), 4, ptr(nll), ptr(loss_ref), None, 0, 0, None, 0, None)
    nll2 = np.zeros(B, np.float32)
    lib.call("howl_ctc_loss", ptr(z), C, T * C, T, B, C, ptr(tg), 3, 3, ptr(il), ptr(tl), 4, ptr(nll2), None, None, 0, 0, None, 0, None)
    np.testing.assert_array_equal(nll2, nll)
    for n_out, with_dx in ((5, False), (12, False), (5, True)):       # vector kernels / GEMM path / row-streaming kernel (from 2048 rows)
        rows, n_in, n_hid = (2050 if with_dx else 40), 128, 256
        x = rng.standard_normal((rows, n_in)).astype(np.float32)
        w1, b1 = (rng.standard_normal((n_hid, n_in)) * 0.1).astype(np.float32), np.zeros(n_hid, np.float32)
        w2, b2 = (rng.standard_normal((n_out, n_hid)) * 0.1).astype(np.float32), np.zeros(n_out, np.float32)
        y1, y2 = np.zeros((rows, n_hid), np.float32), np.zeros((rows, n_out), np.float32)
        hp = HowlHeadParams(ptr(w1), ptr(b1), ptr(w2), ptr(b2))
        geom = (rows, 0, n_in, rows, n_in, n_hid, n_out)
        lib.call("howl_head_fwd", ctypes.byref(hp), ptr(x), *geom, ptr(y1), ptr(y2), None)
        g = [np.zeros_like(a) for a in (w1, b1, w2, b2)]
        gr = HowlHeadGrads(*[ptr(a) for a in g])
        dy2 = rng.standard_normal((rows, n_out)).astype(np.float32)
        dz1 = np.zeros((rows, n_hid), np.float32)
        ws = np.zeros(lib.cdll.howl_head_workspace_bytes(n_in, n_hid, n_out), np.uint8)
        loss = np.full(1, np.nan, np.float32)
        cm = HowlCtcMean(ptr(nll), ptr(tl), B, ptr(loss))
        dx = np.zeros((rows, n_in), np.float32) if with_dx else None
        lib.call("howl_head_bwd", ctypes.byref(hp), ptr(x), *geom, ptr(y1), ptr(dy2), ptr(dz1), ptr(dx), ctypes.byref(gr),
                 ctypes.byref(cm), ptr(ws), ws.size, None)
        np.testing.assert_array_equal(loss, loss_ref)
        np.testing.assert_allclose(g[3], dy2.sum(0), rtol=0, atol=2e-5 * max(1.0, np.abs(dy2.sum(0)).max()))


@pytest.mark.parametrize("big", [False, True])
def test_seq_lstm_backward_in_one_call(lib, monkeypatch, big):
    """howl_seq_lstm_bwd (head + LSTM backward, the wide weight gradients as one job-array launch, one slab fold) == howl_head_bwd
    followed by howl_lstm_bwd, bit for bit; ``big`` forces the 128-row-tile weight-gradient kernel onto this small shape so that
    the job array really carries three jobs."""
    if big:
        monkeypatch.setenv("HOWL_WGRAD_BIG_MIN_ROWS", "1")
    rng = np.random.default_rng(11)
    B, T, M = 9, 11, 40
    x = rng.standard_normal((B, T, M)).astype(np.float32)
    lengths = np.array([11, 11, 10, 9, 9, 7, 4, 2, 1], np.int64)
    sd = om.lstm_init(5)
    _, _, _, keep = run_lstm(lib, sd, x, lengths)
    assert keep["t_out"] == T
    w1, b1, w2, b2 = (np.ascontiguousarray(sd[k].numpy()) for k in ("dnn.0.weight", "dnn.0.bias", "dnn.2.weight", "dnn.2.bias"))
    hseq = keep["bufs"]["hseq"]
    np.nan_to_num(hseq, copy=False)            # rows past an utterance's length hold what the kernel left; h_0 row is NaN-filled
    h1 = np.ascontiguousarray(hseq.reshape(-1)[128:])
    y1, y2 = np.zeros((B * T, 256), np.float32), np.zeros((B * T, 5), np.float32)
    hp = HowlHeadParams(ptr(w1), ptr(b1), ptr(w2), ptr(b2))
    lib.call("howl_head_fwd", ctypes.byref(hp), ptr(h1), T, (T + 1) * 128, 128, B * T, 128, 256, 5, ptr(y1), ptr(y2), None)
    dy2 = rng.standard_normal((B * T, 5)).astype(np.float32)
    head_ws = np.zeros(lib.cdll.howl_head_workspace_bytes(128, 256, 5), np.uint8)

    def grads():
        hg = [np.full_like(a, np.nan) for a in (w1, b1, w2, b2)]
        lg = {k: np.full_like(v, np.nan) for k, v in keep["npz"].items()}
        return hg, lg, HowlHeadGrads(*[ptr(a) for a in hg]), HowlLstmGrads(*[ptr(lg[k]) for k in
                                                                              ("weight_ih_l0", "weight_hh_l0", "bias_ih_l0", "bias_hh_l0")])

    # two calls
    hg_a, lg_a, hgs, lgs = grads()
    dz1, dhs = np.zeros((B * T, 256), np.float32), np.zeros((B, T, 128), np.float32)
    lib.call("howl_head_bwd", ctypes.byref(hp), ptr(h1), T, (T + 1) * 128, 128, B * T, 128, 256, 5, ptr(y1), ptr(dy2), ptr(dz1), ptr(dhs),
             ctypes.byref(hgs), None, ptr(head_ws), head_ws.size, None)
    lib.call("howl_lstm_bwd", ctypes.byref(keep["prm"]), ptr(keep["x"]), B, T, M, ptr(keep["ln"]), None, ctypes.byref(keep["sv"]),
             ptr(dhs), None, None, ctypes.byref(lgs), ptr(keep["ws"]), keep["ws"].size, None)
    dgates_a = keep["bufs"]["dgates"].copy()
    # one call
    hg_b, lg_b, hgs, lgs = grads()
    dz1_b, dhs_b = np.zeros_like(dz1), np.zeros_like(dhs)
    keep["bufs"]["dgates"][:] = np.nan
    lib.call("howl_seq_lstm_bwd", ctypes.byref(hp), 256, 5, ptr(y1), ptr(dy2), ptr(dz1_b), ptr(dhs_b), ctypes.byref(hgs), None,
             ptr(head_ws), head_ws.size, ctypes.byref(keep["prm"]), ptr(keep["x"]), B, T, M, ptr(keep["ln"]), None,
             ctypes.byref(keep["sv"]), ctypes.byref(lgs), ptr(keep["ws"]), keep["ws"].size, None, None)
    np.testing.assert_array_equal(dz1, dz1_b)
    np.testing.assert_array_equal(dhs, dhs_b)
    np.testing.assert_array_equal(dgates_a, keep["bufs"]["dgates"])
    for a, b_ in zip(hg_a, hg_b):
        assert np.isfinite(a).all()
        np.testing.assert_array_equal(a, b_)
    for k in lg_a:
        assert np.isfinite(lg_a[k]).all(), k
        np.testing.assert_array_equal(lg_a[k], lg_b[k], err_msg=k)


@pytest.mark.parametrize("big", [False, True])
def test_seq_lstm_backward_with_the_optimiser_step_in_the_fold(lib, monkeypatch, big):
    """howl_seq_lstm_bwd(..., HowlAdamW) == howl_seq_lstm_bwd + howl_adamw_step on the same flat buffers, bit for bit: gradients,
    parameters and both moments -- whether the step rides in the slab fold (every gradient leaves through it: ``big``) or runs as
    the optimiser's own launch behind it (the small-shape GEMM path folds some gradients elsewhere)."""
    if big:
        monkeypatch.setenv("HOWL_WGRAD_BIG_MIN_ROWS", "1")
        monkeypatch.setenv("HOWL_ROWGEMM_MIN_ROWS", "1")
    rng = np.random.default_rng(5)
    B, T, M = 8, 9, 40
    x = rng.standard_normal((B, T, M)).astype(np.float32)
    lengths = np.array([9, 9, 8, 8, 6, 5, 3, 1], np.int64)
    sd = om.lstm_init(5)
    _, _, _, keep = run_lstm(lib, sd, x, lengths)
    names = ["lstm.weight_ih_l0", "lstm.weight_hh_l0", "lstm.bias_ih_l0", "lstm.bias_hh_l0", "dnn.0.weight", "dnn.0.bias", "dnn.2.weight",
             "dnn.2.bias"]
    sizes = [sd[k].numel() for k in names]
    offs = np.concatenate([[0], np.cumsum(sizes)])
    n = int(offs[-1])
    w1, b1, w2, b2 = (np.ascontiguousarray(sd[k].numpy()) for k in names[4:])
    hseq = keep["bufs"]["hseq"]
    np.nan_to_num(hseq, copy=False)
    h1 = np.ascontiguousarray(hseq.reshape(-1)[128:])
    y1, y2 = np.zeros((B * T, 256), np.float32), np.zeros((B * T, 5), np.float32)
    hp = HowlHeadParams(ptr(w1), ptr(b1), ptr(w2), ptr(b2))
    lib.call("howl_head_fwd", ctypes.byref(hp), ptr(h1), T, (T + 1) * 128, 128, B * T, 128, 256, 5, ptr(y1), ptr(y2), None)
    dy2 = rng.standard_normal((B * T, 5)).astype(np.float32)
    head_ws = np.zeros(lib.cdll.howl_head_workspace_bytes(128, 256, 5), np.uint8)

    def run(fused):
        flat = np.concatenate([sd[k].numpy().reshape(-1) for k in names]).astype(np.float32)
        g = np.full(n, np.nan, np.float32)
        m, v = rng.standard_normal(n).astype(np.float32) * 0.01, np.abs(rng.standard_normal(n)).astype(np.float32) * 1e-4
        m0, v0 = m.copy(), v.copy()
        views = [g[offs[i]:offs[i + 1]] for i in range(8)]
        lgs = HowlLstmGrads(*[ptr(a) for a in views[:4]])
        hgs = HowlHeadGrads(*[ptr(a) for a in views[4:]])
        dz1, dhs = np.zeros((B * T, 256), np.float32), np.zeros((B, T, 128), np.float32)
        keep["bufs"]["dgates"][:] = np.nan
        opt = HowlAdamW(ptr(flat), ptr(g), ptr(m), ptr(v), n, 0.01, 0.9, 0.999, 1e-8, 1e-2, 3, 1.0)
        lib.call("howl_seq_lstm_bwd", ctypes.byref(hp), 256, 5, ptr(y1), ptr(dy2), ptr(dz1), ptr(dhs), ctypes.byref(hgs), None,
                 ptr(head_ws), head_ws.size, ctypes.byref(keep["prm"]), ptr(keep["x"]), B, T, M, ptr(keep["ln"]), None,
                 ctypes.byref(keep["sv"]), ctypes.byref(lgs), ptr(keep["ws"]), keep["ws"].size, ctypes.byref(opt) if fused else None, None)
        if not fused:
            lib.call("howl_adamw_step", ptr(flat), ptr(g), ptr(m), ptr(v), n, 0.01, 0.9, 0.999, 1e-8, 1e-2, 3, 1.0, None)
        return flat, g, m, v, m0, v0

    rng_state = rng.bit_generator.state
    a = run(False)
    rng.bit_generator.state = rng_state      # the same starting moments for the second run
    b_ = run(True)
    for u, w in zip(a, b_):
        assert np.isfinite(u).all()
        np.testing.assert_array_equal(u, w)
    assert not np.array_equal(a[0], np.concatenate([sd[k].numpy().reshape(-1) for k in names]))      # the step did move the weights


def test_forward_with_the_next_batch_frontend_riding_in_the_launch(lib, monkeypatch):
    """howl_lstm_fwd_next (round 5): the log-mel frontend of the NEXT batch as rider blocks of the forward recurrence's launch
    (lstm_fwd4_kernel<40, true>: eight-wave logmel_body) == howl_lstm_fwd followed by howl_logmel_fwd, bit for bit -- the
    recurrence's outputs and the features, in both output layouts and with the ZMUV pair; with HOWL_LSTM_RIDE_LOGMEL=0 (the
    frontend as its own launch behind the recurrence) the same again."""
    from oracle import frontend as ofe
    rng = np.random.default_rng(23)
    B, T, M = 8, 9, 40
    x = rng.standard_normal((B, T, M)).astype(np.float32)
    lengths = np.array([9, 9, 8, 7, 5, 4, 2, 1], np.int64)
    sd = om.lstm_init(5)
    fb = np.ascontiguousarray(ofe.mel_fb(40).numpy())
    fbp = np.zeros(FB_PACKED_FLOATS, np.float32)
    lib.call("howl_fb_pack", ptr(fb), 40, ptr(fbp), None)
    Bn, L = 5, 3000
    pcm = (0.3 * rng.standard_normal((Bn, L))).astype(np.float32)
    zm = np.array([-3.0, 2.5], np.float32)
    Tn = 1 + L // 200
    ref_hs, ref_hT, ref_cT, ref_keep = run_lstm(lib, sd, x, lengths)
    for layout, zmuv in ((1, zm), (0, None)):
        want = np.full((Bn, Tn, 40) if layout else (Bn, 40, Tn), np.nan, np.float32)
        lib.call("howl_logmel_fwd", ptr(pcm), Bn, L, L, ptr(fbp), 40, 1e-7, ptr(zmuv), ptr(want), layout, None)
        for ride in ("1", "0"):
            monkeypatch.setenv("HOWL_LSTM_RIDE_LOGMEL", ride)
            npz = {k: np.ascontiguousarray(sd["lstm." + k].numpy()) for k in ("weight_ih_l0", "weight_hh_l0", "bias_ih_l0", "bias_hh_l0")}
            prm = HowlLstmParams(*[ptr(npz[k]) for k in ("weight_ih_l0", "weight_hh_l0", "bias_ih_l0", "bias_hh_l0")])
            bufs = dict(gates=np.zeros((B, T, 512), np.float32), c=np.zeros((B, T, 128), np.float32),
                        hseq=np.full((B, T + 1, 128), np.nan, np.float32))
            sv = HowlLstmSaved(None, ptr(bufs["gates"]), ptr(bufs["c"]), ptr(bufs["hseq"]), None, T, 0)
            hT, cT = np.zeros((B, 128), np.float32), np.zeros((B, 128), np.float32)
            ws = np.zeros(lib.cdll.howl_lstm_workspace_bytes(B, T), np.uint8)
            got = np.full_like(want, np.nan)
            nxt = HowlLogmelArgs(ptr(pcm), Bn, L, L, ptr(fbp), 40, 1e-7, ptr(zmuv), ptr(got), layout)
            lib.call("howl_lstm_fwd_next", ctypes.byref(prm), ptr(x), B, T, M, ptr(lengths), None, None, ctypes.byref(sv), ptr(hT), ptr(cT),
                     ptr(ws), ws.size, ctypes.byref(nxt), None)
            np.testing.assert_array_equal(got, want)
            np.testing.assert_array_equal(bufs["hseq"][:, 1:], ref_hs)
            np.testing.assert_array_equal(hT, ref_hT)
            np.testing.assert_array_equal(bufs["gates"], ref_keep["bufs"]["gates"])


@pytest.mark.parametrize("name", ["seq-lstm"])      # (the `lstm` half of G15 runs on the device and against the oracle: 70 s here)
def test_golden_whole_clips_on_the_emulator(golden, name):
    """G15 (the reference's recurrent models on clips of 318 / 258 / 206 / 128 frames) through the product's modules on the
    emulator: recurrences at eight times G6's length, the CTC kernel's 128-frame windows, streaming carry over a 160 + 161 split."""
    from emu_util import emulated_package
    from howl_amd import ops
    from howl_amd.model import RegisteredModel
    g = golden("g15_whole_clips_" + name.replace("-", "_"))
    x = torch.from_numpy(g["x"])
    flen = torch.from_numpy(g["frame_lengths"])
    with emulated_package():
        model = RegisteredModel.find_registered_class(name)(5)
        model.load_state_dict({k: v.clone() for k, v in om.lstm_init(5).items()})
        model.eval()
        with torch.no_grad():
            logits = model(x, flen)
        assert logits.shape == g["logits"].shape and np.abs(logits.numpy() - g["logits"]).max() < 2e-5
        model.train()
        sc = model(x, flen)
        if name == "lstm":
            loss = torch.nn.functional.cross_entropy(sc, torch.arange(4) % 5)
        else:
            loss = ops.ctc_loss(sc, torch.from_numpy(g["targets"]), flen, torch.from_numpy(g["target_lengths"]), 4)
        loss.backward()
        assert abs(loss.item() - float(g["loss0"])) < 1e-4 * max(1.0, float(g["loss0"]))
        for n, p in model.named_parameters():
            ref = g["grad0." + n]
            assert np.abs(p.grad.numpy() - ref).max() < 1e-4 * max(1.0, float(np.abs(ref).max())), n
        model.eval().streaming()
        with torch.no_grad():
            if name == "seq-lstm":
                a, b = model(x[:1, :, :, :160], None), model(x[:1, :, :, 160:], None)
            else:
                a, b = model(x[:1, :, :, :160], torch.tensor([160])), model(x[:1, :, :, 160:], torch.tensor([161]))
        assert np.abs(a.numpy() - g["stream_a"]).max() < 2e-5 and np.abs(b.numpy() - g["stream_b"]).max() < 2e-5


@pytest.mark.parametrize("B,T,C", [(4, 5, 2), (3, 16, 8), (2, 3, 3)])
def test_head_ctc_one_launch_other_label_counts_and_tiny_windows(monkeypatch, B, T, C):
    """howl_seq_head_ctc's template over the label count (2, 3, 8 outputs) and windows shorter than one 16-row tile: against the
    three launches it replaces (bit-identical loss / logits / LSTM gradients) and the oracle."""
    from emu_util import emulated_package
    from howl_amd.model import RegisteredModel
    from howl_amd.training.fused import FusedTrainer
    monkeypatch.setenv("HOWL_ROWGEMM_MIN_ROWS", "1")
    rng = np.random.default_rng(B * 10 + T)
    feat = torch.from_numpy(rng.standard_normal((B, 1, 40, T)).astype(np.float32))
    lengths = torch.sort(torch.from_numpy(rng.integers(max(2, T // 2), T + 1, B)), descending=True).values
    lengths[0] = T
    blank = C - 1
    targets = torch.from_numpy(rng.integers(0, C - 1, (B, 2)))
    tl = torch.from_numpy(np.minimum(rng.integers(0, 3, B), (lengths.numpy() + 1) // 2))
    out = {}
    with emulated_package():
        for fused in ("1", "0"):
            monkeypatch.setenv("HOWL_SEQ_HEAD_FUSED", fused)
            model = RegisteredModel.find_registered_class("seq-lstm")(C)
            model.load_state_dict({k: v.clone() for k, v in om.lstm_init(C).items()})
            model.train()
            tr = FusedTrainer(model, None, None, lr=1e-3, weight_decay=1e-5)
            loss = tr.step_sequence_on_features(feat, lengths, targets, tl, blank)
            assert (model.ctc_nll is not None) == (fused == "1")
            out[fused] = (loss.clone(), tr.last_logits.clone(), [g.clone() for g in tr.fp.grad_views])
    a, b = out["1"], out["0"]
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    for n, ga, gb in zip(om.lstm_param_names(), a[2], b[2]):
        if n in ("dnn.0.bias", "dnn.2.weight", "dnn.2.bias"):
            assert (ga - gb).abs().max().item() <= 2e-6 * max(1.0, gb.abs().max().item()), n
        else:
            assert torch.equal(ga, gb), n
    sd = {k: v.clone().requires_grad_(True) for k, v in om.lstm_init(C).items()}
    ref, _ = om.seq_lstm_forward(sd, feat, lengths)
    ref_loss = torch.nn.CTCLoss(blank)(torch.log_softmax(ref, -1), targets, lengths, tl)
    ref_loss.backward()
    if torch.isfinite(ref_loss):
        assert abs(a[0].item() - ref_loss.item()) < 1e-4 * max(1.0, abs(ref_loss.item()))
        for n, ga in zip(om.lstm_param_names(), a[2]):
            r = sd[n].grad
            assert (ga - r).abs().max().item() < 1e-4 * max(1.0, r.abs().max().item()), n


@pytest.mark.parametrize("B,T,odd", [(6, 11, False), (5, 38, True), (3, 70, False)])
def test_head_ctc_and_head_backward_rows_in_one_launch(monkeypatch, B, T, odd):
    """Round 6 (howl_seq_head_ctc): a workgroup owns whole utterances -- first layer in 16-row tiles with y1 kept in LDS, the thin
    output layer, the utterances' CTC recursions on waves 0 .. U-1, then the backward rows made from LDS.  Against the three launches
    it replaces (HOWL_SEQ_HEAD_FUSED=0), through FusedTrainer.step_sequence_on_features: same loss, logits, LSTM gradients and first
    head layer's weight gradient bit for bit (every row's products and sums are taken in the same order); the second layer's and
    the bias gradients, whose per-workgroup slabs cover other rows, to rounding.  Ragged lengths, 1-3 labels (a repeated one), an
    odd batch (a group with one utterance), U = 2 (T = 11, 38) and U = 1 (T = 70: two utterances' rows do not fit the LDS)."""
    from emu_util import emulated_package
    from howl_amd.model import RegisteredModel
    from howl_amd.training.fused import FusedTrainer
    monkeypatch.setenv("HOWL_ROWGEMM_MIN_ROWS", "1")
    rng = np.random.default_rng(B * 100 + T)
    feat = torch.from_numpy(rng.standard_normal((B, 1, 40, T)).astype(np.float32))
    lengths = torch.sort(torch.from_numpy(rng.integers(max(4, T // 2), T + 1, B)), descending=True).values
    lengths[0] = T
    targets = torch.tensor([[0, 1, 2], [3, 3, 0], [2, 0, 0], [1, 0, 3], [0, 0, 0], [2, 1, 3]][:B])
    tl = torch.tensor([3, 2, 1, 3, 0, 2][:B])
    out = {}
    with emulated_package():
        for fused in ("1", "0"):
            monkeypatch.setenv("HOWL_SEQ_HEAD_FUSED", fused)
            model = RegisteredModel.find_registered_class("seq-lstm")(5)
            model.load_state_dict({k: v.clone() for k, v in om.lstm_init(5).items()})
            model.train()
            tr = FusedTrainer(model, None, None, lr=1e-3, weight_decay=1e-5)
            loss = tr.step_sequence_on_features(feat, lengths, targets, tl, 4)
            assert (model.ctc_nll is not None) == (fused == "1")
            out[fused] = (loss.clone(), tr.last_logits.clone(), [g.clone() for g in tr.fp.grad_views], tr.fp.flat.clone())
    a, b = out["1"], out["0"]
    assert torch.isfinite(a[0]) and torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    names = om.lstm_param_names()
    for n, ga, gb in zip(names, a[2], b[2]):
        if n in ("dnn.0.bias", "dnn.2.weight", "dnn.2.bias"):
            assert (ga - gb).abs().max().item() <= 2e-6 * max(1.0, gb.abs().max().item()), n
        else:
            assert torch.equal(ga, gb), n
    assert (a[3] - b[3]).abs().max().item() < 1e-6
    # ... and against the oracle
    sd = {k: v.clone().requires_grad_(True) for k, v in om.lstm_init(5).items()}
    ref, _ = om.seq_lstm_forward(sd, feat, lengths)
    ref_loss = torch.nn.CTCLoss(4)(torch.log_softmax(ref, -1), targets, lengths, tl)
    ref_loss.backward()
    assert abs(a[0].item() - ref_loss.item()) < 1e-4 and (a[1] - ref).abs().max().item() < 1e-4
    for n, ga in zip(names, a[2]):
        r = sd[n].grad
        assert (ga - r).abs().max().item() < 1e-4 * max(1.0, r.abs().max().item()), n
