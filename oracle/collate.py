"""Oracle: collate-side waveform operations.  Test infrastructure only -- never imported by howl_amd/.

CPU restatement, on real tensors, of what the reference does to a list of clips before the hot path:

* ``truncate_length`` (``howl/data/transform/operator.py:73-74``), ``batchify`` (``:77-86``),
  ``tensorize_audio_data`` (``:89-109``), ``random_slice`` (``:60-70``);
* ``TimeshiftTransform`` (``howl/data/transform/transform.py:120-143``), ``NoiseTransform`` (``:168-196``) behind the
  ``AugmentModule.forward`` gate (``:90-97``);
* ``WakeWordFrameBatchifier.__call__`` (``howl/data/transform/batchifier.py:56-118``), quirks included: the negative
  branch builds its "positive intervals" from the *labels* (``timestamp_label_map.values()``) and slices the waveform
  with millisecond numbers as sample indices.

``rand`` is the ``random``-module-like generator the reference module would use (the global ``random`` module or a
``random.Random``); every function consumes draws in the reference's order.  Pinned against goldens G7b / G10
(``tests/test_oracle_golden.py``), which were captured from the reference classes themselves.
"""
import numpy as np
import torch

TIMESHIFT_DOMAIN, TIMESHIFT_IDX, TIMESHIFT_PROB = [0.25, 0.5, 0.75, 1], 0, 0.75
WHITE_DOMAIN, WHITE_IDX = [0.0001, 0.00025, 0.0005, 0.001, 0.002], 3
SP_DOMAIN, SP_IDX = [1 / 20000, 1 / 15000, 1 / 10000, 1 / 5000, 1 / 2500], 2
NOISE_PROB = 0.75


def truncate_length(clips, length=None):
    return [c[..., :length] for c in clips]


def timeshift(rand, clips, training=True, sr=16000):
    """One gate draw; when open: per clip a shift draw then a head/tail draw (transform.py:133-143)."""
    if not (rand.random() < TIMESHIFT_PROB and training):
        return list(clips)
    out = []
    for c in clips:
        n = c.size(-1)
        w = min(int(rand.random() * TIMESHIFT_DOMAIN[TIMESHIFT_IDX] * sr), int(0.5 * n))
        out.append(c[..., w:] if rand.random() < 0.5 else c[..., : n - w])
    return out


def noise(rand, clips, training=True, record=None):
    """Two parameters ("white", then "salt_pepper"), one gate draw each; when open one strength draw per clip; the noise
    samples themselves come from torch's global CPU generator (transform.py:180-196).  ``record`` (a dict) receives the
    per-clip sigma / probability lists."""
    clips = list(clips)
    for name, domain, idx in (("white", WHITE_DOMAIN, WHITE_IDX), ("salt_pepper", SP_DOMAIN, SP_IDX)):
        if not (rand.random() < NOISE_PROB and training):
            if record is not None:
                record[name] = [0.0] * len(clips)
            continue
        strengths = []
        for i, wf in enumerate(clips):
            s = domain[idx] * rand.random()
            strengths.append(s)
            if name == "white":
                mask = torch.empty_like(wf).normal_(0, s)
            else:
                mask = torch.empty_like(wf).bernoulli_(s / 2) - torch.empty_like(wf).bernoulli_(s / 2)
            mask.clamp_(-1, 1)
            clips[i] = (wf + mask).clamp_(-1, 1)
        if record is not None:
            record[name] = strengths
    return clips


def batchify(clips, labels=None):
    """Sort by length descending (stable, ``sorted(..., reverse=True)``), zero-pad right to the longest."""
    order = sorted(range(len(clips)), key=lambda k: clips[k].size(-1), reverse=True)
    lengths = torch.tensor([clips[k].size(-1) for k in order])
    lmax = int(lengths.max())
    audio = torch.stack([torch.cat((clips[k].reshape(-1), torch.zeros(lmax - clips[k].size(-1)))) for k in order])
    lab = None if labels is None else torch.tensor([labels[k] for k in order])
    return audio, lab, lengths, order


def tensorize_audio_data(rand, clips, max_length=None, rand_append=False, **extra):
    """operator.py:89-109: ``np.argsort(-lengths)`` order, zero padding after -- or, with ``rand_append``, on a side drawn
    per clip (one draw per clip, in sorted order) -- up to ``max_length`` (default: the longest clip)."""
    lengths = np.array([c.size(-1) for c in clips])
    order = np.argsort(-lengths)
    clips = [clips[k] for k in order]
    extra = {k: [v[j] for j in order] for k, v in extra.items()}
    if max_length is None:
        max_length = max(c.size(-1) for c in clips)
    rows = []
    for c in clips:
        flat = c.reshape(-1)
        pad = torch.zeros(max_length - c.size(-1))
        rows.append(torch.cat((pad, flat)) if (rand_append and rand.random() < 0.5) else torch.cat((flat, pad)))
    return torch.stack(rows), extra


def random_slice(rand, clip, max_window_size):
    n = clip.size(-1)
    if n < max_window_size:
        return clip
    a = rand.randint(0, n - max_window_size)
    return clip[..., a:a + max_window_size]


def frame_batchify(rand, clips, label_maps, negative_label, positive_sample_prob=0.5, window_size_ms=500,
                   sample_rate=16000, positive_delta_ms=150, eps_ms=20, pad_to_window=True):
    """``WakeWordFrameBatchifier.__call__``: ``label_maps[i]`` is clip i's ``timestamp_label_map`` (end ms -> label).
    Returns (audio (B, W), labels (B,), lengths (B,))."""
    picked = []
    for wf, tl in zip(clips, label_maps):
        if not tl:
            picked.append((negative_label, random_slice(rand, wf, int(sample_rate * window_size_ms / 1000))))
            continue
        select_negative = rand.random() > positive_sample_prob
        if not select_negative:
            end_ms, label = rand.choice(list(tl.items()))
            end_ms_rand = end_ms + (rand.random() * eps_ms)
            b = int((end_ms_rand / 1000) * sample_rate)
            a = max(b - int((window_size_ms / 1000) * sample_rate), 0)
            rand.random()                     # the reference's dead `if random.random() < 0:` still draws
            if b - a < 0:
                select_negative = True
            else:
                picked.append((label, wf[..., a:b]))
        if select_negative:
            spans = sorted(((v - positive_delta_ms, v + positive_delta_ms) for v in tl.values()), key=lambda s: s[0])
            gaps, last = [], 0
            for a, b in spans:
                if last < a:
                    gaps.append((last, a))
                last = b
            gaps.append((b, int(len(wf) / 16000 * 1000)))
            a, b = rand.choice(gaps)
            if b - a > window_size_ms:
                a = rand.randint(0, int(b - window_size_ms))
                b = a + window_size_ms
            picked.append((negative_label, wf[..., a:b]))       # millisecond numbers used as sample indices, as there
    max_length = int(window_size_ms / 1000 * sample_rate) if pad_to_window else None
    audio, extra = tensorize_audio_data(rand, [w for _, w in picked], rand_append=True, max_length=max_length,
                                        labels=[l for l, _ in picked], lengths=[w.size(-1) for _, w in picked])
    return audio, torch.tensor(extra["labels"]), torch.tensor(extra["lengths"])
