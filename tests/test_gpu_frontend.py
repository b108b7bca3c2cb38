"""-m gpu parity: HIP frontend (through the Python modules -> C ABI) vs golden vectors and the oracle.

Tolerances (fp32): log-mel <= 2e-4 abs where the mel power is above the 1e-7 floor region (log > -8), <= 2e-3 abs
everywhere (log amplifies relative error near the eps floor); integer outputs exact.
"""
import random

import numpy as np
import pytest
import torch

from gpu_util import DEV, maxerr, t
from howl_amd.lib import FB_PACKED_FLOATS
from oracle import frontend as ofe

pytestmark = pytest.mark.gpu


def logmel_close(out, ref):
    out, ref = out.detach().cpu(), torch.as_tensor(ref)
    assert out.shape == ref.shape
    d = (out - ref).abs()
    assert d.max().item() < 2e-3, d.max().item()
    strong = ref > -8
    assert d[strong].max().item() < 2e-4, d[strong].max().item()


@pytest.fixture(scope="module")
def std():
    from howl_amd.data.transform.transform import StandardAudioTransform
    return StandardAudioTransform().to(DEV).eval()


def test_golden_gsc_eval(std, golden):
    g = golden("g2_frontend_gsc")
    audio = t(g["audio"]).to(DEV)
    feats = std(audio)
    assert feats.shape == (6, 3, 40, 81)
    logmel_close(feats[:, 0], g["feats"][:, 0])
    assert maxerr(feats[:, 1], g["feats"][:, 1]) < 2e-3 and maxerr(feats[:, 2], g["feats"][:, 2]) < 2e-3
    logmel_close(std(audio, mels_only=True), g["mels"])
    d = std(t(g["mels"]).to(DEV), deltas_only=True)   # exact inputs -> tight tolerance on the delta kernel alone
    assert maxerr(d, g["feats"]) < 2e-6
    assert torch.equal(std.compute_lengths(t(g["lens_in"]).to(DEV)).cpu(), t(g["lens_out"]))


def test_golden_synth_edges(std, golden):
    g = golden("g2_frontend_synth")   # tone+noise, silence, full-scale square, impulses at n=0 and n=L-1
    for L in (8000, 16000, 13527):
        logmel_close(std(t(g[f"audio_{L}"]).to(DEV), mels_only=True), g[f"mels_{L}"])


def test_vtlp_train_mode(golden):
    from howl_amd.data.transform.transform import StandardAudioTransform
    g = golden("g2_frontend_gsc")
    std = StandardAudioTransform().to(DEV).train()
    random.seed(11)   # same global-RNG protocol as the reference: one gate draw, then alpha
    feats = std(t(g["audio"]).to(DEV))
    assert abs(std.last_vtlp_alpha - float(g["vtlp_alpha"])) == 0.0
    logmel_close(feats[:, 0], g["mels_vtlp"])
    for alpha in (0.9, 1.0999):   # alpha > 1 exercises the re-mapped corner points (transform.py:397-401)
        fb = ofe.mel_fb(40, alpha=alpha)
        from howl_amd import ops
        from howl_amd.data.transform.transform import mel_corner_points, vtlp_warp_points
        out = torch.empty(FB_PACKED_FLOATS, device=DEV)
        ops.fb_from_points(vtlp_warp_points(mel_corner_points(40, 16000), alpha, 16000).tolist(), 40, 8000, out)
        assert maxerr(out[:260 * 48].view(260, 48)[:257, :40], fb) < 2e-7


def test_zmuv_golden(std, golden):
    from howl_amd.data.transform.operator import ZmuvTransform
    g2, g4 = golden("g2_frontend_gsc"), golden("g4_zmuv")
    audio, lengths = t(g2["audio"]), g2["lengths"]
    z = ZmuvTransform().to(DEV)
    for pos in [0, 2, 3, 5, 4, 1]:   # file order of the clips inside the length-sorted batch
        z.update(std(audio[pos:pos + 1, : int(lengths[pos])].to(DEV)))
    assert z.total.item() == g4["total"][0]
    assert abs(z.mean.item() - g4["mean"][0]) < 2e-5 and abs(z.mean2.item() - g4["mean2"][0]) < 2e-4
    z.mean.copy_(t(g4["mean"]))
    z.mean2.copy_(t(g4["mean2"]))
    normed = z(std(audio.to(DEV)))
    d = (normed[:, 0].cpu() - t(g4["normed_ch0"])).abs()
    assert d.max().item() < 1e-3
    fused = std.log_mel_for_model(audio.to(DEV), z)   # (B,1,M,T) view of the (B,T,M) fused output
    assert fused.shape == (6, 1, 40, 81) and maxerr(fused[:, 0], normed[:, 0]) < 1e-5


def test_zmuv_masked_update_vs_oracle(std, golden):
    """ZmuvTransform.update(data, mask) (operator.py:128-130) on the device, broadcast mask included, against the oracle;
    the cached (mean, std) pair follows the new statistics."""
    from howl_amd.data.transform.operator import ZmuvTransform
    g2 = golden("g2_frontend_gsc")
    feats = t(g2["feats"])                                     # (6, 3, 40, 81)
    gen = torch.Generator().manual_seed(4)
    z, zo = ZmuvTransform().to(DEV), ofe.Zmuv()
    z.update(feats[:2].to(DEV))
    zo.update(feats[:2])
    before = z.pair().clone()
    for mask in ((torch.rand(4, 3, 40, 81, generator=gen) < 0.5).float(), (torch.rand(4, 1, 1, 81, generator=gen) < 0.7).float()):
        z.update(feats[2:].to(DEV), mask.to(DEV))
        zo.update(feats[2:], mask)          # the oracle (= the reference's arithmetic) counts the mask as given
    assert z.total.item() == float(zo.total)
    assert abs(z.mean.item() - float(zo.mean)) < 2e-6 * max(1.0, abs(float(zo.mean)))
    assert abs(z.mean2.item() - float(zo.mean2)) < 2e-6 * max(1.0, abs(float(zo.mean2)))
    after = z.pair()
    assert not torch.equal(after, before) and abs(after[1].item() - float(zo.std)) < 1e-5


def test_specaug_golden(golden):
    from howl_amd.data.transform.transform import SpecAugmentTransform
    g = golden("g7_specaug")
    sa = SpecAugmentTransform().train()
    sa.rand = random.Random(5)
    for p in sa.augment_params:
        p.prob = 1.1
    sa.augment_params[1].domain = [10, 50, 60, 125, 150]
    out = sa(t(g["x"]).to(DEV))
    assert torch.equal(out.cpu(), t(g["out"]))


def test_full_size_properties(std):
    """BASELINE sizes (512 x 1 s): spot-check against the oracle and size-independent invariants."""
    from howl_amd.utils.synth import synthetic_pcm
    pcm = synthetic_pcm(512, 16000)
    out = std(pcm.to(DEV), mels_only=True)
    idx = [0, 1, 63, 200, 511]
    logmel_close(out[idx], ofe.standard_audio_transform(pcm[idx], ofe.mel_fb(40), mels_only=True))
    alone = std(pcm[200:201].to(DEV), mels_only=True)
    # every frame is transformed by its own 16 lanes (real-input FFT-256, no pairing with a neighbour): a clip's features do
    # not depend on what else is in the batch, not even in the last bit
    assert torch.equal(alone[0], out[200])
    from howl_amd import ops
    tm = ops.logmel(pcm.to(DEV), std._standard_fb(), 40, None, layout=1)
    assert torch.equal(tm.permute(0, 2, 1), out)               # the two layouts are the same numbers
    assert torch.equal(std(pcm.to(DEV), mels_only=True), out)  # run-to-run determinism
    with pytest.raises(Exception):
        std(pcm[:, :200].to(DEV), mels_only=True)              # too short for reflect padding: loud error, like torch.stft
    with pytest.raises(Exception):
        std(pcm[:2])                                           # CPU tensor: no fallback


def test_device_collate_with_dataset_mixer(golden):
    """a12 / train.py:218: DatasetMixer -> truncate -> Timeshift -> Noise -> batchify as one launch.  With the noise
    parameters' gates closed the batch equals the oracle's DatasetMixer + crop on the same `random` stream exactly."""
    from howl_amd.data.collate import DeviceCollate
    from howl_amd.data import collate as collate_mod
    g = golden("g9_mixer")
    wf_lens, bg_lens = g["wf_lens"].tolist(), g["bg_lens"].tolist()
    rng = np.random.default_rng(3)
    bank = torch.zeros(len(wf_lens), max(wf_lens))
    for i, L in enumerate(wf_lens):
        bank[i, :L] = t(0.1 * rng.standard_normal(L).astype(np.float32))
    bgb = torch.zeros(len(bg_lens), max(bg_lens))
    for i, L in enumerate(bg_lens):
        bgb[i, :L] = t(0.2 * rng.standard_normal(L).astype(np.float32))
    labels = torch.arange(len(wf_lens))
    seen_mix = False
    for seed in (0, 3, 11, 12):
        dc = DeviceCollate(bank.to(DEV), torch.tensor(wf_lens), labels.to(DEV), max_len=16000, seed=seed,
                           background=(bgb.to(DEV), bg_lens))
        old = collate_mod.NOISE_PROB, collate_mod.TIMESHIFT_PROB
        collate_mod.NOISE_PROB, collate_mod.TIMESHIFT_PROB = -1.0, -1.0   # gates still draw, never fire
        try:
            batch = dc(list(range(len(wf_lens))))
        finally:
            collate_mod.NOISE_PROB, collate_mod.TIMESHIFT_PROB = old
        ref = ofe.dataset_mixer(random.Random(seed), [bank[i, :L] for i, L in enumerate(wf_lens)],
                                [bgb[i, :L] for i, L in enumerate(bg_lens)])
        order = sorted(range(len(wf_lens)), key=lambda k: -wf_lens[k])
        assert batch.lengths.tolist() == [wf_lens[k] for k in order]
        for row, k in enumerate(order):
            assert maxerr(batch.audio_data[row, :wf_lens[k]], ref[k]) < 1e-6
            assert not batch.audio_data[row, wf_lens[k]:].any()
        seen_mix |= any(a != 0 for a in dc.last_mix[2])
    assert seen_mix


@pytest.mark.parametrize("L", [257, 399, 600, 1000, 3001])
def test_short_and_odd_lengths_vs_oracle(std, L):
    """Clips of two to sixteen frames, lengths that are not multiples of the hop, every frame an edge frame (reflect padding on
    both sides at once for L < 512), frame counts that are not multiples of the 4 frames a wave works on."""
    gen = torch.Generator().manual_seed(L)
    pcm = (torch.rand(5, L, generator=gen) * 2 - 1) * 0.5
    out = std(pcm.to(DEV), mels_only=True)
    assert out.shape == (5, 40, 1 + L // 200)
    logmel_close(out, ofe.standard_audio_transform(pcm, ofe.mel_fb(40), mels_only=True))


@pytest.mark.parametrize("stride", [1008, 1009, 200, 7])
def test_overlapping_strided_rows(std, stride):
    """The frame engine hands the kernel all windows of a clip as ONE strided view (rows overlap, row stride = the evaluation
    stride in samples).  Even strides take the 8-byte sample loads, odd ones the per-sample path: both must give exactly what
    the same windows give as a contiguous batch."""
    from howl_amd import ops
    from howl_amd.utils.synth import synthetic_pcm
    clip = synthetic_pcm(1, 40000, seed=5)[0].to(DEV)
    n, L = 12, 8000
    windows = clip.as_strided((n, L), (stride, 1))
    got = ops.logmel(windows, std._standard_fb(), 40, None, layout=0)
    want = ops.logmel(windows.contiguous(), std._standard_fb(), 40, None, layout=0)
    assert torch.equal(got, want)
    logmel_close(got, ofe.standard_audio_transform(windows.cpu().contiguous(), ofe.mel_fb(40), mels_only=True))


def test_frontend_at_80_mel_bins(monkeypatch, golden):
    """NUM_MELS = 80 (stock Howl, settings.py:32): the filterbank is packed as two banks of 40 columns; since round 6 ONE launch
    contracts a quad's power spectrum with both (banded tables per bank).  Eval and VTLP train mode through the module against the oracle
    at full size, plus the golden clips (torchaudio's definition restated in the oracle, pinned at 40 bins by G1 / G2)."""
    from howl_amd.data.transform.operator import ZmuvTransform
    from howl_amd.data.transform.transform import StandardAudioTransform
    from howl_amd.settings import SETTINGS
    from howl_amd.utils.synth import synthetic_pcm
    monkeypatch.setattr(SETTINGS.audio_transform, "num_mels", 80)
    std = StandardAudioTransform().to(DEV).eval()
    fb = ofe.mel_fb(80)
    g = golden("g2_frontend_gsc")
    audio = t(g["audio"])
    feats = std(audio.to(DEV))
    assert feats.shape == (6, 3, 80, 81)
    ref = ofe.standard_audio_transform(audio, fb)
    logmel_close(feats[:, 0], ref[:, 0])
    assert maxerr(feats[:, 1], ref[:, 1]) < 2e-3 and maxerr(feats[:, 2], ref[:, 2]) < 2e-3
    pcm = synthetic_pcm(512, 16000)
    zmuv = ZmuvTransform().to(DEV)
    zmuv.update(std(pcm[:4].to(DEV)))
    view = std.log_mel_for_model(pcm.to(DEV), zmuv)        # (B, 1, 80, T) view of the (B, T, 80) buffer res8 reads
    z = ofe.Zmuv()
    z.update(ofe.standard_audio_transform(pcm[:4], fb))
    refz = z(ofe.standard_audio_transform(pcm[::37], fb, mels_only=True))
    d = (view[::37, 0].cpu() - refz).abs()
    assert d.max().item() < 2e-3
    std.train()
    random.seed(5)
    out = std(audio.to(DEV), mels_only=True)
    random.seed(5)
    gate = random.random()
    alpha = random.random() * 0.2 + 0.9 if gate < 0.75 else None
    logmel_close(out, ofe.standard_audio_transform(audio, ofe.mel_fb(80, alpha=alpha), mels_only=True))
    # round 6: one pass over the spectrum with both banks' banded tables == the two-launch all-pairs form, bit for bit (the pairs
    # the tables leave out are exact zeros); eval filterbank at full size and 40 VTLP draws on the golden clips
    std.eval()
    one = std.log_mel_for_model(pcm.to(DEV), zmuv).clone()
    monkeypatch.setenv("HOWL_LOGMEL_TWO_LAUNCHES", "1")
    two = std.log_mel_for_model(pcm.to(DEV), zmuv).clone()
    monkeypatch.delenv("HOWL_LOGMEL_TWO_LAUNCHES")
    assert torch.equal(one, two)
    for Bq, Lq in ((3, 2377), (7, 8000), (5, 13527), (1, 700), (33, 4001)):     # ragged tails, odd strides, one short clip
        x = synthetic_pcm(Bq, Lq, seed=Lq).to(DEV)
        a = std(x, mels_only=True).clone()
        monkeypatch.setenv("HOWL_LOGMEL_TWO_LAUNCHES", "1")
        b = std(x, mels_only=True).clone()
        monkeypatch.delenv("HOWL_LOGMEL_TWO_LAUNCHES")
        assert torch.equal(a, b), (Bq, Lq)
        logmel_close(a, ofe.standard_audio_transform(x.cpu(), fb, mels_only=True))
    std.train()
    for seed in range(40):
        random.seed(seed)
        a = std(audio.to(DEV), mels_only=True).clone()
        monkeypatch.setenv("HOWL_LOGMEL_TWO_LAUNCHES", "1")
        random.seed(seed)
        b = std(audio.to(DEV), mels_only=True).clone()
        monkeypatch.delenv("HOWL_LOGMEL_TWO_LAUNCHES")
        assert torch.equal(a, b), seed
