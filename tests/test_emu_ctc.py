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
    lib.call("howl_ctc_loss", ptr(z), C, T * C, T, B, C, ptr(tg), tg.shape[1], int(tl.max()), ptr(il), ptr(tl), blank, ptr(nll),
             ptr(loss), ptr(dz), C, T * C, None)
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


def test_ctc_range_is_checked(lib):
    assert lib.cdll.howl_ctc_supported(128, 64, 31) == 1
    assert lib.cdll.howl_ctc_supported(129, 5, 3) == 0 and lib.cdll.howl_ctc_supported(40, 65, 3) == 0
    assert lib.cdll.howl_ctc_supported(40, 5, 32) == 0
    logits, targets, in_len, tgt_len, blank = make_case(12, 2, 5, 3, 0)
    from howl_amd.lib import HowlHipError
    with pytest.raises(HowlHipError):
        z = np.ascontiguousarray(logits.numpy())
        nll, loss = np.zeros(2, np.float32), np.zeros(1, np.float32)
        lib.call("howl_ctc_loss", ptr(z), 5, 60, 12, 2, 5, ptr(targets.numpy()), 3, 40, ptr(in_len.numpy()), ptr(tgt_len.numpy()),
                 blank, ptr(nll), ptr(loss), None, 0, 0, None)
