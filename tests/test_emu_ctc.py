"""Fused log_softmax + CTC kernel on the hipemu CPU emulator vs torch's CPU log_softmax + ctc_loss + autograd."""
import numpy as np
import pytest
import torch

from ctc_util import make_case, reference
from emu_util import emu_lib, ptr


@pytest.fixture(scope="module")
def lib():
    return emu_lib()


def run(lib, logits_btc, targets, in_len, tgt_len, blank, want_grad=True):
    B, T, C = logits_btc.shape
    z = np.ascontiguousarray(logits_btc.numpy())                  # (B,T,C) memory, addressed as (T,B,C)
    tg = np.ascontiguousarray(targets.numpy())
    il, tl = np.ascontiguousarray(in_len.numpy()), np.ascontiguousarray(tgt_len.numpy())
    nll, loss = np.zeros(B, np.float32), np.zeros(1, np.float32)
    dz = np.full((B, T, C), np.nan, np.float32) if want_grad else None
    nws = int(lib.cdll.howl_ctc_workspace_floats(T, B)) if want_grad else 0
    assert nws == (B * T * 64 if T > 128 else 0) or not want_grad
    ws = np.full(nws, np.nan, np.float32) if nws else None
    lib.call("howl_ctc_loss", ptr(z), C, T * C, T, B, C, ptr(tg), tg.shape[1], int(tl.max()), ptr(il), ptr(tl), blank, ptr(nll),
             ptr(loss), ptr(dz), C, T * C, ptr(ws), nws, None)
    return nll, float(loss[0]), dz


@pytest.mark.parametrize("T,B,C,Lmax,seed,tight", [(12, 6, 5, 3, 0, False), (20, 5, 9, 6, 1, True), (7, 3, 3, 2, 2, False)])
def test_ctc_vs_torch(lib, T, B, C, Lmax, seed, tight):
    logits, targets, in_len, tgt_len, blank = make_case(T, B, C, Lmax, seed, tight=tight)
    per, loss, grad = reference(logits, targets, in_len, tgt_len, blank)
    nll, l, dz = run(lib, logits, targets, in_len, tgt_len, blank)
    assert np.abs(nll - per.numpy()).max() < 2e-5 * max(1.0, float(per.abs().max()))
    assert abs(l - float(loss)) < 2e-5 * max(1.0, abs(float(loss)))
    assert np.abs(dz - grad.numpy()).max() < 1e-5                    # fp32 log-space recursions on both sides
    for b in range(B):
        assert not dz[b, int(in_len[b]):].any()                    # rows past the utterance's length are exactly zero


def test_ctc_first_label_is_blank_index_zero_and_loss_only(lib):
    logits, targets, in_len, tgt_len, blank = make_case(10, 4, 4, 3, 5, blank=0)
    per, loss, grad = reference(logits, targets, in_len, tgt_len, blank)
    nll, l, dz = run(lib, logits, targets, in_len, tgt_len, blank)
    assert np.abs(dz - grad.numpy()).max() < 1e-5 and abs(l - float(loss)) < 2e-5
    nll2, l2, none = run(lib, logits, targets, in_len, tgt_len, blank, want_grad=False)
    assert none is None and l2 == l and np.array_equal(nll2, nll)


@pytest.mark.parametrize("T,B,C,Lmax,seed", [(300, 5, 5, 3, 3), (129, 3, 7, 5, 4), (257, 4, 6, 4, 6), (384, 2, 5, 31, 7)])
def test_ctc_whole_clips_in_windows(lib, T, B, C, Lmax, seed):
    """Beyond 128 frames (whole clips, batchifier.py:14-34 + train.py:291-296) the kernel walks the time axis in 128-frame
    windows: alpha alone through the leading windows (rows parked in the workspace), the last window as before, beta back
    through the leading windows.  Ragged lengths put utterances of one, two and three windows in the same batch (incl. a
    length of exactly 128 / 256: a full last window)."""
    logits, targets, in_len, tgt_len, blank = make_case(T, B, C, Lmax, seed)
    if T >= 257:
        in_len[-1] = 128                     # one full window exactly
        in_len[-2] = 256 if B > 2 else in_len[-2]
    per, loss, grad = reference(logits, targets, in_len, tgt_len, blank)
    nll, l, dz = run(lib, logits, targets, in_len, tgt_len, blank)
    assert np.isfinite(per.numpy()).all()
    assert np.abs(nll - per.numpy()).max() < 2e-5 * max(1.0, float(per.abs().max()))
    assert abs(l - float(loss)) < 2e-5 * max(1.0, abs(float(loss)))
    # alpha + beta reach -700 .. -800 over these lengths (one fp32 ulp there is 6e-5): the fp32 log-space recursion of
    # torch's own kernel sits 2e-5 .. 8e-5 from the fp64 one, and so does this one -- the bound is torch-fp32's own distance
    _, _, grad64 = reference(logits.double(), targets, in_len, tgt_len, blank)
    noise = float((grad.double() - grad64).abs().max())
    assert np.abs(dz - grad64.numpy()).max() < 1.5 * noise + 1e-5
    assert np.abs(dz - grad.numpy()).max() < 2.5 * noise + 1e-5
    for b in range(B):
        assert not dz[b, int(in_len[b]):].any()
    nll2, l2, none = run(lib, logits, targets, in_len, tgt_len, blank, want_grad=False)      # loss only: no workspace
    assert none is None and l2 == l and np.array_equal(nll2, nll)


def test_ctc_window_boundaries_do_not_change_a_short_utterance(lib):
    """An utterance of <= 128 frames gives the same bits whether the batch's T makes the launch a one-window or a
    several-window one (its rows past the end are zeros either way)."""
    logits, targets, in_len, tgt_len, blank = make_case(100, 4, 5, 3, 11)
    nll_a, _, dz_a = run(lib, logits, targets, in_len, tgt_len, blank)
    wide = torch.cat([logits, torch.randn(4, 200, 5)], 1)
    nll_b, _, dz_b = run(lib, wide, targets, in_len, tgt_len, blank)
    assert np.array_equal(nll_a, nll_b) and np.array_equal(dz_a, dz_b[:, :100]) and not dz_b[:, 100:].any()


def test_ctc_range_is_checked(lib):
    assert lib.cdll.howl_ctc_supported(128, 64, 31) == 1 and lib.cdll.howl_ctc_supported(8192, 64, 31) == 1
    assert lib.cdll.howl_ctc_supported(8193, 5, 3) == 0 and lib.cdll.howl_ctc_supported(40, 65, 3) == 0
    assert lib.cdll.howl_ctc_supported(40, 5, 32) == 0
    from howl_amd.lib import HowlHipError
    lg, tg_, il_, tl_, bl = make_case(200, 2, 5, 3, 0)
    with pytest.raises(HowlHipError, match="workspace"):      # a gradient beyond 128 frames without the workspace
        z = np.ascontiguousarray(lg.numpy())
        lib.call("howl_ctc_loss", ptr(z), 5, 1000, 200, 2, 5, ptr(tg_.numpy()), 3, 3, ptr(il_.numpy()), ptr(tl_.numpy()), bl,
                 ptr(np.zeros(2, np.float32)), None, ptr(np.zeros((2, 200, 5), np.float32)), 5, 1000, None, 0, None)
    logits, targets, in_len, tgt_len, blank = make_case(12, 2, 5, 3, 0)
    from howl_amd.lib import HowlHipError
    with pytest.raises(HowlHipError):
        z = np.ascontiguousarray(logits.numpy())
        nll, loss = np.zeros(2, np.float32), np.zeros(1, np.float32)
        lib.call("howl_ctc_loss", ptr(z), 5, 60, 12, 2, 5, ptr(targets.numpy()), 3, 40, ptr(in_len.numpy()), ptr(tgt_len.numpy()),
                 blank, ptr(nll), ptr(loss), None, 0, 0, None, 0, None)
