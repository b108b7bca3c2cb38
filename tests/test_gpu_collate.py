"""-m gpu parity of the collate-side waveform operations on the device: timeshift crops, noise,
DatasetMixer-free chain, WakeWordFrameBatchifier -- against oracle/collate.py (pinned to the reference classes by goldens G7b /
G10) on the same host draws.  Data movement is bit-exact; the noise *samples* come from the device's counter-based generator,
so they are compared in distribution (and bounded per sample)."""
import random

import numpy as np
import pytest
import torch

from gpu_util import DEV
from oracle import collate as oc
from test_oracle_golden import G10_VARIANTS, g10_check, g10_inputs

pytestmark = pytest.mark.gpu


def _bank(lens, seed=3):
    rng = np.random.default_rng(seed)
    clips = [torch.from_numpy((0.2 * rng.standard_normal(L)).astype(np.float32)) for L in lens]
    bank = torch.zeros(len(lens), max(lens))
    for i, c in enumerate(clips):
        bank[i, : c.numel()] = c
    return clips, bank


def _oracle_chain(seed, clips, max_len, labels):
    """compose(truncate_length, Timeshift.train(), Noise.train(), batchify) on the CPU with the same `random` stream; returns the
    crops BEFORE the noise (exact expectation for the data movement), the noise strengths, and the batch order."""
    rand = random.Random(seed)
    cropped = oc.timeshift(rand, oc.truncate_length(clips, max_len))
    rec = {}
    oc.noise(rand, cropped, record=rec)
    audio, lab, lengths, order = oc.batchify(cropped, labels)
    return audio, lab, lengths, order, rec


def _seed_with(lens, want_shift, want_white, want_sp, start=0):
    """First seed whose gate draws (timeshift, white, salt-pepper) open exactly the requested augmentations."""
    for seed in range(start, start + 2000):
        r = random.Random(seed)
        shift = r.random() < oc.TIMESHIFT_PROB
        if shift:
            for _ in lens:
                r.random(); r.random()
        white = r.random() < oc.NOISE_PROB
        if white:
            for _ in lens:
                r.random()
        sp = r.random() < oc.NOISE_PROB
        if (shift, white, sp) == (want_shift, want_white, want_sp):
            return seed
    raise AssertionError("no seed found")


def test_timeshift_crop_rows_equal_oracle():
    """Timeshift gate open, noise gates shut: every row of the device batch is the oracle's crop, bit for bit, in batchify's
    order, zero padded; labels and lengths follow."""
    from howl_amd.data.collate import DeviceCollate
    lens = [16000 - 137 * (i % 11) - (3000 if i % 5 == 0 else 0) for i in range(48)] + [20000, 9000]
    clips, bank = _bank(lens)
    labels = [i % 7 for i in range(len(lens))]
    for trial in range(3):
        seed = _seed_with(lens, True, False, False, start=1000 * trial)
        dc = DeviceCollate(bank.to(DEV), torch.tensor(lens), torch.tensor(labels).to(DEV), max_len=16000, seed=seed)
        batch = dc(list(range(len(lens))))
        audio, lab, lengths, _, rec = _oracle_chain(seed, clips, 16000, labels)
        assert rec["white"] == [0.0] * len(lens) and rec["salt_pepper"] == [0.0] * len(lens)
        assert torch.equal(batch.lengths.cpu(), lengths)
        assert torch.equal(batch.labels.cpu(), lab)
        assert torch.equal(batch.audio_data.cpu(), audio)
        assert (lengths < torch.tensor([min(l, 16000) for l in lens]).max()).any()       # something was actually cropped


def test_noise_on_gpu_matches_the_reference_distribution():
    """All gates open at config-5 scale (512 x 1 s): residual = device batch - oracle crop.  Per row: zero mean and standard
    deviation sigma_b of the white noise (sigma_b = 0.001 * draw); over the batch: salt-and-pepper events at the drawn rates;
    everything clamped to [-1, 1]; the same (seed, call) reproduces the batch exactly, another seed does not."""
    from howl_amd.data.collate import DeviceCollate
    B = 512
    lens = [16000 - 29 * (i % 50) for i in range(B)]
    clips, bank = _bank(lens, seed=5)
    labels = [i % 12 for i in range(B)]
    seed = _seed_with(lens, True, True, True)
    dc = DeviceCollate(bank.to(DEV), torch.tensor(lens), torch.tensor(labels).to(DEV), max_len=16000, seed=seed)
    batch = dc(list(range(B)))
    audio, lab, lengths, order, rec = _oracle_chain(seed, clips, 16000, labels)
    assert torch.equal(batch.lengths.cpu(), lengths) and torch.equal(batch.labels.cpu(), lab)
    got = batch.audio_data.cpu()
    assert got.abs().max().item() <= 1.0
    resid = (got - audio).double()
    events = expected_events = 0.0
    for row, k in enumerate(order):
        n = int(lengths[row])
        assert not got[row, n:].any()                                     # padding stays zero (no noise there)
        r = resid[row, :n]
        spikes = r.abs() > 0.02        # salt (+1) / pepper (-1), clamped at +-1: |r| = 1 -+ x; white noise is < 0.007
        events += int(spikes.sum())
        expected_events += rec["salt_pepper"][k] * n                      # P(exactly one of salt, pepper) ~ p
        w = r[~spikes]
        sigma = rec["white"][k]
        assert abs(w.mean().item()) < 5 * sigma / np.sqrt(n) + 1e-9, (row, w.mean().item(), sigma)
        assert abs(w.std().item() - sigma) < 0.05 * sigma + 1e-8, (row, w.std().item(), sigma)
        assert w.abs().max().item() < 7 * sigma + 1e-7
    assert abs(events - expected_events) < 5 * np.sqrt(expected_events) + 5, (events, expected_events)
    # determinism: same seed and call index -> same batch; a different seed -> different noise, same crops
    dc2 = DeviceCollate(bank.to(DEV), torch.tensor(lens), torch.tensor(labels).to(DEV), max_len=16000, seed=seed)
    assert torch.equal(dc2(list(range(B))).audio_data.cpu(), got)
    dc3 = DeviceCollate(bank.to(DEV), torch.tensor(lens), torch.tensor(labels).to(DEV), max_len=16000, seed=seed)
    dc3._seed = seed + 12345                                               # same host draws, other device noise key
    other = dc3(list(range(B))).audio_data.cpu()
    assert not torch.equal(other, got) and (other - audio).abs().max().item() <= 1.0 + 0.01


def test_white_noise_tails_and_clamp():
    """Large sigma / probability through the op itself: N(0, sigma) moments incl. the tails, Bernoulli rates, clamping."""
    from howl_amd import ops
    L = 1 << 16
    bank = torch.zeros(3, L, device=DEV)
    bank[2] = 0.95
    i32 = lambda a: torch.tensor(a, dtype=torch.int32, device=DEV)
    f32 = lambda a: torch.tensor(a, dtype=torch.float32, device=DEV)
    out = ops.collate_augment(bank, i32([0, 1, 2]), i32([L, L, L]), i32([0, 0, 0]), i32([1, 1, 1]), f32([0.05, 0.0, 0.1]),
                              f32([0.0, 0.2, 0.0]), 77, L).cpu().double()
    z = out[0] / 0.05
    assert abs(z.mean().item()) < 0.02 and abs(z.std().item() - 1.0) < 0.01
    assert abs((z.abs() > 2).double().mean().item() - 0.0455) < 0.004       # two-sigma tail mass of a normal
    assert abs((z ** 4).mean().item() - 3.0) < 0.15                          # kurtosis
    salt, pepper = (out[1] > 0.5).double().mean().item(), (out[1] < -0.5).double().mean().item()
    assert abs(salt - 0.1 * 0.9) < 0.006 and abs(pepper - 0.1 * 0.9) < 0.006    # p/2 each, minus coincidences
    assert out[2].max().item() == 1.0 and out[2].min().item() > 0.0         # 0.95 + N(0, 0.1) clamps at +1


def test_frame_batchifier_on_device_vs_reference_golden(golden):
    """WakeWordFrameBatchifier through howl_gather_windows on the GPU == the reference class's own batches (G10)."""
    from howl_amd.data.transform.batchifier import DeviceClip, WakeWordFrameBatchifier
    g = golden("g10_frame_batchifier")
    clips, maps = g10_inputs(g)
    bank = torch.zeros(len(clips), max(c.numel() for c in clips))
    for i, c in enumerate(clips):
        bank[i, : c.numel()] = c
    bank = bank.to(DEV)
    examples = [DeviceClip(i, c.numel(), m) for i, (c, m) in enumerate(zip(clips, maps))]
    for trial, (seed, kw) in enumerate(G10_VARIANTS):
        rand = random.Random(seed)
        batch = WakeWordFrameBatchifier(4, bank=bank, rand=rand, **kw)(examples)
        g10_check(g, trial, batch.audio_data.cpu().numpy(), batch.labels.cpu().numpy(), batch.lengths.numpy())
        assert rand.random() == float(g[f"next_draw_{trial}"])


def test_augmented_frame_batches_vs_oracle():
    """compose(Timeshift, Noise, WakeWordFrameBatchifier) (train.py:211-229) as ONE launch vs the oracle chain on the same
    `random` stream: labels, lengths, which columns are zero padding -- exact; samples -- exact when the noise gates are shut,
    within the drawn noise amplitude otherwise."""
    from howl_amd.data.collate import DeviceCollate
    from howl_amd.data.transform.batchifier import DeviceClip, WakeWordFrameBatchifier
    rng = np.random.default_rng(9)
    lens = [int(v) for v in rng.integers(6000, 40000, 40)]
    clips, bank = _bank(lens, seed=11)
    maps = []
    for i, L in enumerate(lens):
        k = i % 4                                                           # 0..3 labelled words per clip
        ends = sorted(float(v) for v in rng.uniform(100, L / 16 - 50, k))
        maps.append({e: int(j % 3) for j, e in enumerate(ends)})
    examples = [DeviceClip(i, L, m) for i, (L, m) in enumerate(zip(lens, maps))]
    exact = noisy = 0
    for seed in range(400):
        probe = random.Random(seed)                                       # which gates does this seed open?  (host only)
        pre = {}
        oc.noise(probe, oc.timeshift(probe, [c[:1] for c in clips]), record=pre)
        quiet = max(pre["white"]) == 0.0 and max(pre["salt_pepper"]) == 0.0
        if (quiet and exact >= 2) or (not quiet and noisy >= 4):
            continue
        # the global `random` module drives everything in the reference; here one Random per side, same seed
        r_dev, r_ora = random.Random(seed), random.Random(seed)
        dc = DeviceCollate(bank.to(DEV), torch.tensor(lens), None, max_len=max(lens), seed=seed)
        dc.rand = r_dev
        fb = WakeWordFrameBatchifier(3, window_size_ms=500, rand=r_dev)
        batch = dc.frame_batch(examples, fb)
        cropped = oc.timeshift(r_ora, [c for c in clips])
        rec = {}
        oc.noise(r_ora, cropped, record=rec)                              # strengths only; the crops stay noise-free
        audio, labels, lengths = oc.frame_batchify(r_ora, cropped, maps, 3, window_size_ms=500)
        assert r_dev.random() == r_ora.random()                           # both sides consumed the same draws
        got = batch.audio_data.cpu()
        assert torch.equal(batch.labels.cpu(), labels) and torch.equal(batch.lengths, lengths)
        amp = max(rec["white"]) * 7 + (1.0 if max(rec["salt_pepper"]) > 0 else 0.0)
        if amp == 0.0:
            assert torch.equal(got, audio)
            exact += 1
        else:
            d = (got - audio).abs()
            assert (d > 7 * max(rec["white"]) + 1e-7).double().mean().item() < 1e-3      # only salt/pepper events exceed it
            assert not d[audio == 0].gt(1.0).any()
            # padding columns carry no noise: a row's zeros outside its window stay exactly zero
            for row in range(audio.shape[0]):
                n = int(lengths[row])
                front = bool(audio[row, 0] == 0) and n > 0 and n < audio.shape[1]
                if n == 0:
                    assert not got[row].any()
                elif front:
                    assert not got[row, : audio.shape[1] - n].any()
                else:
                    assert not got[row, n:].any()
            noisy += 1
        if exact >= 2 and noisy >= 4:
            break
    assert exact >= 2 and noisy >= 4


def test_ragged_clip_bank_matches_the_dense_bank():
    """``WakeWordClipBank`` keeps the clips ragged (flat buffer + offsets; one long clip costs its own length only).  The same
    clips through a dense (N, Lmax) ``DeviceCollate`` and through the ragged bank's ``row_offsets`` form give bit-identical
    frame and sequence batches (same seeds: same draws, same counter-based noise)."""
    from types import SimpleNamespace
    from howl_amd.data.collate import DeviceCollate
    from howl_amd.data.common.tokenizer import WakeWordTokenizer
    from howl_amd.data.common.vocab import Vocab
    from howl_amd.data.transform.batchifier import AudioSequenceBatchifier, DeviceClip, WakeWordFrameBatchifier
    from howl_amd.training.data import WakeWordClipBank
    rng = np.random.default_rng(21)
    lens = [int(v) for v in rng.integers(3000, 30000, 24)] + [200001]          # one long negative
    clips, dense = _bank(lens, seed=5)
    meta = [SimpleNamespace(path=f"c{i}", transcription="hey fire fox" if i % 2 else "other words", end_timestamps=None)
            for i in range(len(lens))]
    bank = WakeWordClipBank([c.reshape(-1) for c in clips], meta, labeler=None, device=DEV)
    assert bank.flat.numel() < sum(lens) + 4 * len(lens) + 4 and bank.max_len == max(lens)
    assert all(torch.equal(bank.clip(i).cpu(), clips[i].reshape(-1)) for i in range(len(lens)))
    maps = [{float(100 + 37 * i): int(i % 3)} if i % 3 else {} for i in range(len(lens))]
    examples = [DeviceClip(i, L, m, meta[i].transcription) for i, (L, m) in enumerate(zip(lens, maps))]
    vocab = Vocab({"hey": 0, "fire": 1, "fox": 2}, oov_token_id=3)
    for seed in (0, 1, 2, 3):
        out = []
        for ragged in (False, True):
            if ragged:
                dc = DeviceCollate(bank.rows, bank.lengths, None, max_len=bank.max_len, seed=seed, row_offsets=bank.offsets)
            else:
                dc = DeviceCollate(dense.to(DEV), torch.tensor(lens), None, max_len=max(lens), seed=seed)
            fb = WakeWordFrameBatchifier(3, window_size_ms=500, rand=dc.rand)
            a = dc.frame_batch(examples, fb)
            b = dc.sequence_batch(examples, AudioSequenceBatchifier(3, WakeWordTokenizer(vocab, ignore_oov=False)))
            out.append((a.audio_data.cpu(), a.labels.cpu(), a.lengths, b.audio_data.cpu(), b.labels.cpu(), b.audio_lengths))
        for u, v in zip(*out):
            assert torch.equal(torch.as_tensor(u), torch.as_tensor(v))
