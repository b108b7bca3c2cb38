"""Frontend kernels run UNMODIFIED on the hipemu CPU emulator (tiny shapes) against the oracle and the goldens.
Catches indexing / layout / sync mistakes before a GPU run; the GPU parity tests are in test_gpu_*.py."""
import numpy as np
import pytest
import torch

from emu_util import emu_lib, ptr
from howl_amd.lib import FB_PACKED_FLOATS, HowlMelPoints, fb_packed_floats
from oracle import frontend as fe


@pytest.fixture(scope="module")
def lib():
    return emu_lib()


def pack_fb(lib, fb):
    fbp = np.zeros(fb_packed_floats(fb.shape[1]), np.float32)
    assert fbp.size == lib.cdll.howl_fb_packed_floats(fb.shape[1])
    fb = np.ascontiguousarray(fb, np.float32)
    lib.call("howl_fb_pack", ptr(fb), fb.shape[1], ptr(fbp), None)
    return fbp


def logmel(lib, audio, fbp, M=40, zmuv=None, layout=0):
    B, L = audio.shape
    T = 1 + L // 200
    out = np.full((B, M, T) if layout == 0 else (B, T, M), np.nan, np.float32)
    lib.call("howl_logmel_fwd", ptr(audio), B, L, L, ptr(fbp), M, 1e-7, ptr(zmuv), ptr(out), layout, None)
    return out


def test_fb_pack_and_points(lib):
    fb = fe.mel_fb(40).numpy()
    fbp = pack_fb(lib, fb)[:260 * 48].reshape(260, 48)
    assert np.array_equal(fbp[:257, :40], fb) and not fbp[257:].any() and not fbp[:, 40:].any()
    # corner points -> triangles on the "device", standard and VTLP-warped (alpha > 1 quirk included)
    import math
    for alpha in (None, 0.93, 1.0999):
        m_pts = torch.linspace(0.0, 2595.0 * math.log10(1.0 + 8000.0 / 700.0), 42)
        f_pts = 700.0 * (10 ** (m_pts / 2595.0) - 1.0)
        if alpha is not None:
            thr = 4800 * min(alpha, 1) / alpha
            f_pts[f_pts <= thr] *= alpha
            f = f_pts[f_pts > thr]
            f_pts[f_pts > thr] = 8000 - ((8000 - 4800 * min(alpha, 1)) / (8000 - thr)) * (8000 - f)
        pts = HowlMelPoints()
        for i, v in enumerate(f_pts.tolist()):
            pts.f[i] = v
        out = np.zeros(FB_PACKED_FLOATS, np.float32)
        lib.call("howl_fb_from_points", pts, 40, 8000.0, ptr(out), None)
        ref = fe.mel_fb(40, alpha=alpha).numpy()
        np.testing.assert_allclose(out[:260 * 48].reshape(260, 48)[:257, :40], ref, rtol=0, atol=2e-7)


def test_logmel_matches_oracle_and_golden(lib, golden):
    g = golden("g2_frontend_synth")
    fbp = pack_fb(lib, fe.mel_fb(40).numpy())
    for L in (8000, 13527):
        audio = np.ascontiguousarray(g[f"audio_{L}"][[0, 2, 4]])  # tone+noise, square wave, impulse at L-1
        out = logmel(lib, audio, fbp)
        ref = g[f"mels_{L}"][[0, 2, 4]]
        # log of a power spectrum: compare where mel power is not at the eps floor, abs elsewhere
        np.testing.assert_allclose(out, ref, rtol=0, atol=2e-3)
        strong = ref > -8
        assert np.abs(out - ref)[strong].max() < 2e-4
        out_t = logmel(lib, audio, fbp, layout=1)
        assert np.array_equal(out_t, out.transpose(0, 2, 1))


@pytest.mark.parametrize("M", [80, 64, 50])
def test_more_than_48_mel_bins_run_as_two_banks(lib, M):
    """NUM_MELS = 80 is the reference's stock default (settings.py:32): the contraction covers 48 bins per pass, so wider
    filterbanks are packed as two banks ([0, lo), [lo, M), lo = 4 ceil(M / 8)) and howl_logmel_fwd makes one pass per bank into
    the same output.  Packed images, the device-built triangles (standard and VTLP-warped) and the log-mels in both layouts
    against the oracle; 64 and 50: uneven banks."""
    import math
    lo = 4 * ((M + 7) // 8)
    fb = fe.mel_fb(M).numpy()
    fbp = pack_fb(lib, fb)
    assert fbp.size == 2 * FB_PACKED_FLOATS
    b0 = fbp[:260 * 48].reshape(260, 48)
    b1 = fbp[FB_PACKED_FLOATS:FB_PACKED_FLOATS + 260 * 48].reshape(260, 48)
    assert np.array_equal(b0[:257, :lo], fb[:, :lo]) and not b0[:, lo:].any()
    assert np.array_equal(b1[:257, :M - lo], fb[:, lo:]) and not b1[:, M - lo:].any()
    for alpha in (None, 1.0999):
        m_pts = torch.linspace(0.0, 2595.0 * math.log10(1.0 + 8000.0 / 700.0), M + 2)
        f_pts = 700.0 * (10 ** (m_pts / 2595.0) - 1.0)
        if alpha is not None:
            thr = 4800 * min(alpha, 1) / alpha
            f_pts[f_pts <= thr] *= alpha
            f = f_pts[f_pts > thr]
            f_pts[f_pts > thr] = 8000 - ((8000 - 4800 * min(alpha, 1)) / (8000 - thr)) * (8000 - f)
        pts = HowlMelPoints()
        for i, v in enumerate(f_pts.tolist()):
            pts.f[i] = v
        out = np.zeros(fb_packed_floats(M), np.float32)
        lib.call("howl_fb_from_points", pts, M, 8000.0, ptr(out), None)
        ref = fe.mel_fb(M, alpha=alpha).numpy()
        np.testing.assert_allclose(out[:260 * 48].reshape(260, 48)[:257, :lo], ref[:, :lo], rtol=0, atol=2e-7)
        np.testing.assert_allclose(out[FB_PACKED_FLOATS:FB_PACKED_FLOATS + 260 * 48].reshape(260, 48)[:257, :M - lo], ref[:, lo:],
                                   rtol=0, atol=2e-7)
    rng = np.random.default_rng(M)
    audio = (0.1 * rng.standard_normal((3, 2377))).astype(np.float32)
    zm = np.array([-2.0, 1.5], np.float32)
    out = logmel(lib, audio, fbp, M=M, zmuv=zm)
    ref = (fe.standard_audio_transform(torch.from_numpy(audio), fe.mel_fb(M), mels_only=True).numpy() + 2.0) / 1.5
    np.testing.assert_allclose(out, ref, rtol=0, atol=1e-4)
    out_t = logmel(lib, audio, fbp, M=M, zmuv=zm, layout=1)
    assert np.array_equal(out_t, out.transpose(0, 2, 1))


def test_80_mel_bins_in_one_pass_over_the_spectrum(lib, monkeypatch):
    """Round 6: the stock NUM_MELS = 80 no longer runs the transform twice and every (slot, group) pair per bank -- one launch keeps
    a quad's power values in registers and contracts them with both banks' banded fragment tables (16 + 62 pairs, swept over the
    VTLP warps like the 40-bin table; csrc/howl_logmel.hip.h wide_mask).  Same products in the same order as the two-launch
    all-pairs form (the pairs left out are exact zeros): bit-identical log-mels for the standard and for VTLP-warped filterbanks,
    in both layouts; the packed buffer says which table its banded image is laid out for (flag word [1]); a matrix the tables do
    not cover takes the all-pairs path inside the same launch."""
    import math
    M = 80
    rng = np.random.default_rng(80)
    audio = (0.1 * rng.standard_normal((3, 2377))).astype(np.float32)
    zm = np.array([-2.0, 1.5], np.float32)
    flags = lambda fbp, bank: fbp[bank * FB_PACKED_FLOATS:(bank + 1) * FB_PACKED_FLOATS][-32:].view(np.int32)[:2].tolist()
    cases = []
    fb = fe.mel_fb(M).numpy()
    cases.append(("standard", pack_fb(lib, fb), fe.mel_fb(M), True))
    for alpha in (0.9, 1.0999):
        m_pts = torch.linspace(0.0, 2595.0 * math.log10(1.0 + 8000.0 / 700.0), M + 2)
        f_pts = 700.0 * (10 ** (m_pts / 2595.0) - 1.0)
        thr = 4800 * min(alpha, 1) / alpha
        f_pts[f_pts <= thr] *= alpha
        f = f_pts[f_pts > thr]
        f_pts[f_pts > thr] = 8000 - ((8000 - 4800 * min(alpha, 1)) / (8000 - thr)) * (8000 - f)
        pts = HowlMelPoints()
        for i, v in enumerate(f_pts.tolist()):
            pts.f[i] = v
        out = np.zeros(fb_packed_floats(M), np.float32)
        lib.call("howl_fb_from_points", pts, M, 8000.0, ptr(out), None)
        cases.append((f"vtlp {alpha}", out, fe.mel_fb(M, alpha=alpha), True))
    dense = fb.copy()
    dense[200, 3] = 0.25                      # a weight far outside the triangles: not covered by the tables
    cases.append(("uncovered", pack_fb(lib, dense), torch.from_numpy(dense), False))
    for name, fbp, ref_fb, covered in cases:
        assert flags(fbp, 0)[0] == 0 and flags(fbp, 1)[0] == 0, name        # FBQ is not in the 40-bin table's layout
        assert (flags(fbp, 0)[1] == 1 and flags(fbp, 1)[1] == 1) == covered, (name, flags(fbp, 0), flags(fbp, 1))
        monkeypatch.delenv("HOWL_LOGMEL_TWO_LAUNCHES", raising=False)
        one = logmel(lib, audio, fbp, M=M, zmuv=zm)
        one_t = logmel(lib, audio, fbp, M=M, zmuv=zm, layout=1)
        monkeypatch.setenv("HOWL_LOGMEL_TWO_LAUNCHES", "1")
        two = logmel(lib, audio, fbp, M=M, zmuv=zm)
        monkeypatch.delenv("HOWL_LOGMEL_TWO_LAUNCHES", raising=False)
        assert np.array_equal(one, two), name
        assert np.array_equal(one_t, one.transpose(0, 2, 1)), name
        ref = (fe.standard_audio_transform(torch.from_numpy(audio), ref_fb, mels_only=True).numpy() + 2.0) / 1.5
        np.testing.assert_allclose(one, ref, rtol=0, atol=1e-4, err_msg=name)
    # a 40-bin filterbank still announces the 40-bin table
    fbp40 = pack_fb(lib, fe.mel_fb(40).numpy())
    assert fbp40[-32:].view(np.int32)[:2].tolist() == [1, 0]


def test_logmel_zmuv_and_ragged_tail(lib):
    rng = np.random.default_rng(0)
    audio = (0.1 * rng.standard_normal((3, 1000))).astype(np.float32)   # T = 6 -> 18 frames, chunk of 16 + 2
    fbp = pack_fb(lib, fe.mel_fb(40).numpy())
    zm = np.array([-3.0, 2.5], np.float32)
    out = logmel(lib, audio, fbp, zmuv=zm)
    ref = (fe.standard_audio_transform(torch.from_numpy(audio), fe.mel_fb(40), mels_only=True).numpy() + 3.0) / 2.5
    np.testing.assert_allclose(out, ref, rtol=0, atol=1e-4)


def test_deltas(lib):
    rng = np.random.default_rng(1)
    x = rng.standard_normal((2, 40, 9)).astype(np.float32)
    out = np.zeros((2, 3, 40, 9), np.float32)
    lib.call("howl_deltas_fwd", ptr(x), 2, 40, 9, None, ptr(out), None)
    ref = fe.standard_audio_transform(torch.from_numpy(x), None, deltas_only=True).numpy()
    np.testing.assert_allclose(out, ref, rtol=0, atol=1e-6)
    x1 = rng.standard_normal((1, 40, 3)).astype(np.float32)   # T < win_length: replicate padding dominates
    out1 = np.zeros((1, 3, 40, 3), np.float32)
    lib.call("howl_deltas_fwd", ptr(x1), 1, 40, 3, None, ptr(out1), None)
    np.testing.assert_allclose(out1, fe.standard_audio_transform(torch.from_numpy(x1), None, deltas_only=True).numpy(),
                               rtol=0, atol=1e-6)


def test_zmuv_update_and_specaug(lib, golden):
    rng = np.random.default_rng(2)
    total, mean, mean2 = (np.zeros(1, np.float32) for _ in range(3))
    scratch = np.zeros(2, np.float64)
    z = fe.Zmuv()
    for n in (1000, 77, 5000):
        x = rng.standard_normal(n).astype(np.float32) * 3 - 7
        lib.call("howl_zmuv_update", ptr(x), n, ptr(total), ptr(mean), ptr(mean2), ptr(scratch), None)
        z.update(torch.from_numpy(x))
    assert total[0] == z.total.item()
    np.testing.assert_allclose(mean, z.mean.numpy(), rtol=1e-6)
    np.testing.assert_allclose(mean2, z.mean2.numpy(), rtol=1e-6)
    pair = np.zeros(2, np.float32)
    lib.call("howl_zmuv_pair", ptr(mean), ptr(mean2), ptr(pair), None)
    np.testing.assert_allclose(pair, [z.mean.item(), z.std.item()], rtol=1e-5)
    # masked updates (operator.py:128-130) continue the same running statistics: count = mask.sum()
    scratch3 = np.zeros(3, np.float64)
    for n in (640, 3001):
        x = rng.standard_normal(n).astype(np.float32) * 2 + 1
        m = (rng.uniform(size=n) < 0.6).astype(np.float32)
        lib.call("howl_zmuv_update_masked", ptr(x), ptr(m), n, 1.0, ptr(total), ptr(mean), ptr(mean2), ptr(scratch3), None)
        z.update(torch.from_numpy(x), torch.from_numpy(m))
    assert total[0] == float(z.total)
    np.testing.assert_allclose(mean, np.asarray(z.mean).reshape(-1), rtol=1e-6)
    np.testing.assert_allclose(mean2, np.asarray(z.mean2).reshape(-1), rtol=1e-6)

    g = golden("g7_specaug")
    x = np.ascontiguousarray(g["x"])
    f0, f, t0, t = (np.ascontiguousarray(g[k], np.int32) for k in ("f0", "f", "t0", "t"))  # keep alive
    lib.call("howl_specaug_mask", ptr(x), 6, 3, 40, 81, 3 * 40 * 81, 40 * 81, 81, 1, ptr(f0), ptr(f), ptr(t0), ptr(t), None)
    assert np.array_equal(x, g["out"])


def test_collate_augment(lib):
    """gather + timeshift crop + right zero-pad exactly; noise: clamped, right scale, deterministic per (seed, b, n)."""
    rng = np.random.default_rng(4)
    bank = (0.1 * rng.standard_normal((5, 3000))).astype(np.float32)
    idx = np.array([3, 0, 4], np.int32)
    src_len = np.array([3000, 2500, 2000], np.int32)
    shift = np.array([100, 0, 700], np.int32)
    head = np.array([1, 0, 0], np.int32)
    zero = np.zeros(3, np.float32)
    out = np.full((3, 2950), np.nan, np.float32)
    args = lambda sg, sp, seed, o: ("howl_collate_augment", ptr(bank), 3000, ptr(idx), ptr(src_len), ptr(shift), ptr(head),
                                    ptr(sg), ptr(sp), seed, 3, 2950, ptr(o), None)
    lib.call(*args(zero, zero, 7, out))
    assert np.array_equal(out[0, :2900], bank[3, 100:3000]) and not out[0, 2900:].any()      # head crop
    assert np.array_equal(out[1, :2500], bank[0, :2500]) and not out[1, 2500:].any()         # no shift, zero pad
    assert np.array_equal(out[2, :1300], bank[4, :1300]) and not out[2, 1300:].any()         # tail crop
    sg = np.array([0.05, 0.0, 0.0], np.float32)
    sp = np.array([0.0, 0.2, 0.0], np.float32)
    a, b = np.zeros_like(out), np.zeros_like(out)
    lib.call(*args(sg, sp, 7, a))
    lib.call(*args(sg, sp, 7, b))
    assert np.array_equal(a, b)                                                              # reproducible
    d0 = a[0, :2900] - bank[3, 100:3000]
    assert abs(d0.std() - 0.05) < 0.005 and abs(d0.mean()) < 0.005                           # N(0, sigma)
    d1 = a[1, :2500] - bank[0, :2500]
    frac = (np.abs(d1) > 0.5).mean()
    assert 0.12 < frac < 0.25 and np.abs(a).max() <= 1.0                                     # ~p/2 salt + ~p/2 pepper (minus overlaps)
    assert np.array_equal(a[2], out[2])                                                      # untouched sample
    lib.call(*args(sg, sp, 8, b))
    assert not np.array_equal(a[0], b[0])                                                    # seed changes the noise


def test_collate_augment_mix(lib):
    """DatasetMixer in front of the chain: x*(1-alpha) + bg[off + n]*alpha on the pre-shift coordinates, then crop / pad."""
    rng = np.random.default_rng(5)
    bank = (0.1 * rng.standard_normal((4, 3000))).astype(np.float32)
    bg = (0.2 * rng.standard_normal((3, 5000))).astype(np.float32)
    idx = np.array([1, 3, 0], np.int32)
    src_len = np.array([3000, 2500, 2000], np.int32)
    shift = np.array([100, 0, 700], np.int32)
    head = np.array([1, 0, 0], np.int32)
    zero = np.zeros(3, np.float32)
    bg_idx = np.array([2, 0, 1], np.int32)
    bg_off = np.array([1234, 0, 2999], np.int32)
    alpha = np.array([0.15, 0.0, 1.0], np.float32)
    out = np.full((3, 2950), np.nan, np.float32)
    lib.call("howl_collate_augment_mix", ptr(bank), 3000, ptr(idx), ptr(src_len), ptr(shift), ptr(head), ptr(zero), ptr(zero), 9,
             ptr(bg), 5000, ptr(bg_idx), ptr(bg_off), ptr(alpha), 3, 2950, ptr(out), None)
    m0 = bank[1, :3000] * np.float32(0.85) + bg[2, 1234:4234] * np.float32(0.15)
    np.testing.assert_allclose(out[0, :2900], m0[100:], rtol=0, atol=1e-7)        # mixed, then head crop
    assert not out[0, 2900:].any()
    assert np.array_equal(out[1, :2500], bank[3, :2500]) and not out[1, 2500:].any()   # alpha 0: untouched
    np.testing.assert_allclose(out[2, :1300], bg[1, 2999:4299], rtol=0, atol=1e-7)     # alpha 1: replaced, tail crop
    assert not out[2, 1300:].any()


def test_frame_batchifier_vs_golden(lib, golden):
    """a12/f1: the product's WakeWordFrameBatchifier host decisions (same `random` stream as the reference class) + the
    howl_gather_windows kernel (emulated) reproduce the reference's batches of golden G10 bit for bit."""
    import random
    from howl_amd.data.transform.batchifier import DeviceClip, WakeWordFrameBatchifier
    from test_oracle_golden import G10_VARIANTS, g10_check, g10_inputs
    g = golden("g10_frame_batchifier")
    clips, maps = g10_inputs(g)
    lmax = max(c.numel() for c in clips)
    bank = np.zeros((len(clips), lmax), np.float32)
    for i, c in enumerate(clips):
        bank[i, : c.numel()] = c.numpy()
    examples = [DeviceClip(i, c.numel(), m) for i, (c, m) in enumerate(zip(clips, maps))]
    for trial, (seed, kw) in enumerate(G10_VARIANTS):
        rand = random.Random(seed)
        plan = WakeWordFrameBatchifier(4, rand=rand, **kw).plan(examples)
        i32 = lambda a: np.ascontiguousarray(a, np.int32)
        idx, start, length, dst = i32(plan.clip_id), i32(plan.start), i32(plan.length), i32(plan.dst_off)
        out = np.full((len(examples), plan.width), np.nan, np.float32)
        lib.call("howl_gather_windows", ptr(bank), lmax, ptr(idx), ptr(start), ptr(length), ptr(dst), len(examples),
                 plan.width, ptr(out), None)
        g10_check(g, trial, out, plan.labels, plan.length)
        assert rand.random() == float(g[f"next_draw_{trial}"])


def test_gather_windows_edges(lib):
    """Widths that are not a multiple of 4, empty windows, windows at either end of the row."""
    rng = np.random.default_rng(8)
    bank = rng.standard_normal((3, 1000)).astype(np.float32)
    for width in (1, 5, 250, 1003):
        idx = np.array([2, 0, 1, 1], np.int32)
        length = np.array([min(width, 1000), 0, min(3, width), min(width, 7)], np.int32)
        start = np.array([0, 10, 997, 500], np.int32)
        dst = np.array([width - length[0], 0, width - length[2], 0], np.int32)
        out = np.full((4, width), np.nan, np.float32)
        lib.call("howl_gather_windows", ptr(bank), 1000, ptr(idx), ptr(start), ptr(length), ptr(dst), 4, width, ptr(out), None)
        for b in range(4):
            exp = np.zeros(width, np.float32)
            exp[dst[b]:dst[b] + length[b]] = bank[idx[b], start[b]:start[b] + length[b]]
            assert np.array_equal(out[b], exp), (width, b)


@pytest.mark.parametrize("M", [1, 7, 40, 48])
def test_logmel_dense_filterbank_and_group_edges(lib, M):
    """The mel contraction walks band limits per group of 4 mel bins: a dense random filterbank (every band = all 17 bin
    groups), a filterbank with an empty column group in the middle, and mel counts that end inside / at the edge of a group."""
    rng = np.random.default_rng(M)
    B, L = 3, 1000                      # T = 6 frames per clip: 18 frames = 4.5 quads, frames straddle clips
    audio = (0.3 * rng.standard_normal((B, L))).astype(np.float32)
    power = fe.power_spectrogram(torch.from_numpy(audio))            # (B, 257, T)
    for variant in ("dense", "holes"):
        fb = rng.uniform(0.1, 1.0, (257, M)).astype(np.float32)
        if variant == "holes":
            fb[:, 4:8] = 0.0                                          # a whole group without weights (when M > 4)
            fb[:100, : M // 2] = 0.0
            fb[30:, M // 2:] *= (np.arange(257 - 30)[:, None] < 60)   # bands of different extent
            fb[0, 0] = 0.5                                            # keep column 0 alive for M = 1
        fbp = pack_fb(lib, fb)
        ref = torch.log(torch.matmul(power.transpose(-1, -2), torch.from_numpy(fb)) + 1e-7).numpy()   # (B, T, M)
        out = logmel(lib, audio, fbp, M=M, layout=1)
        np.testing.assert_allclose(out, ref, rtol=0, atol=3e-4)
        out0 = logmel(lib, audio, fbp, M=M, layout=0)
        np.testing.assert_array_equal(out0.transpose(0, 2, 1), out)


def test_logmel_vtlp_banded_and_unaligned_rows(lib):
    """VTLP-warped filterbanks stay on the banded fragment table (flag set by howl_fb_pack), also at the alpha > 1 quirk; rows
    with an odd stride take the per-sample load path and give the same numbers as 8-byte aligned rows."""
    rng = np.random.default_rng(11)
    B, L = 3, 2601                      # T = 14: interior quads (fast loads) and edge quads
    buf = (0.3 * rng.standard_normal((B, L + 2))).astype(np.float32)
    even = np.ascontiguousarray(buf[:, :L + 1])        # row stride L + 1 = 2602 (even)
    odd = np.ascontiguousarray(buf[:, :L + 2][:, :L + 2])
    audio = np.ascontiguousarray(buf[:, :L])
    T = 1 + L // 200
    for alpha in (0.9, 1.0999):
        fb = fe.mel_fb(40, alpha=alpha).numpy()
        fbp = pack_fb(lib, fb)
        assert fbp[260 * 48 + 17 * 64 * 4 + 17 * 12 * 64:].view(np.int32)[0] == 1      # banded table covers it
        ref = fe.standard_audio_transform(torch.from_numpy(audio), torch.from_numpy(fb), mels_only=True).numpy()
        outs = []
        for rows, ld in ((even, L + 1), (odd, L + 2)):
            out = np.full((B, 40, T), np.nan, np.float32)
            lib.call("howl_logmel_fwd", ptr(rows), B, L, ld, ptr(fbp), 40, 1e-7, None, ptr(out), 0, None)
            outs.append(out)
        np.testing.assert_allclose(outs[0], ref, rtol=0, atol=3e-4)
        np.testing.assert_array_equal(outs[0], outs[1])
    dense = pack_fb(lib, rng.uniform(0.1, 1.0, (257, 40)).astype(np.float32))
    assert dense[260 * 48 + 17 * 64 * 4 + 17 * 12 * 64:].view(np.int32)[0] == 0
