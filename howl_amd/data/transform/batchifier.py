"""``WakeWordFrameBatchifier`` / ``AudioSequenceBatchifier`` (``howl/data/transform/batchifier.py:13-118``) for clips that
live on the device.

The reference runs these in DataLoader workers on CPU tensors: a Python loop that slices one window per example and pads
it.  Here the examples are *descriptors* of rows of a device-resident clip bank (``DeviceClip``); the host only makes the
decisions -- in the reference's order and with the reference's draws from ``random`` -- and ONE ``howl_gather_windows``
launch cuts, pads and stacks all windows of the batch in HBM.  Decisions that look odd are the reference's and are kept
on purpose, because they decide what the model is trained on:

* the negative branch derives its "positive intervals" from ``timestamp_label_map.values()`` (the labels) and uses the
  resulting millisecond numbers as *sample* indices, so a negative taken from a positive clip is ``window_size_ms``
  samples long (``batchifier.py:88-106``);
* the dead ``if random.random() < 0`` still consumes a draw (``:74``);
* rows are ordered by ``np.argsort(-lengths)`` and padded before or after on a per-row draw (``operator.py:89-109``).
"""
import random
from dataclasses import dataclass, field
from typing import Dict, List, Sequence

import numpy as np
import torch

from howl_amd import ops
from howl_amd.data.common.batch import ClassificationBatch, SequenceBatch

__all__ = ["DeviceClip", "WakeWordFrameBatchifier", "AudioSequenceBatchifier", "WindowPlan"]


@dataclass
class DeviceClip:
    """What a ``WakeWordClipExample`` is to the device path: a row of the clip bank plus its frame labels."""
    clip_id: int
    num_samples: int
    timestamp_label_map: Dict[float, int] = field(default_factory=dict)     # word end (ms) -> label
    transcription: str = ""


@dataclass
class WindowPlan:
    """Host-side result of a batchifier call: one window per row, already in batch order."""
    clip_id: List[int]
    start: List[int]
    length: List[int]
    dst_off: List[int]
    labels: List[int]
    width: int
    source: List[int] = field(default_factory=list)      # index (into the examples passed in) each row came from


def _clamped(a: int, b: int, n: int):
    """(start, length) of the Python slice ``[a:b]`` of an n-sample clip."""
    lo, hi, _ = slice(a, b).indices(n)
    return lo, max(hi - lo, 0)


class WakeWordFrameBatchifier:
    def __init__(self, negative_label: int, positive_sample_prob: float = 0.5, window_size_ms: int = 500,
                 sample_rate: int = 16000, positive_delta_ms: int = 150, eps_ms: int = 20, pad_to_window: bool = True,
                 bank: torch.Tensor = None, rand=None):
        """``bank``: the (N, Lmax) device matrix the examples' ``clip_id`` index; ``rand``: a ``random.Random`` (default: the
        global ``random`` module, as the reference)."""
        self.negative_label = negative_label
        self.positive_sample_prob = positive_sample_prob
        self.window_size_ms = window_size_ms
        self.sample_rate = sample_rate
        self.positive_delta_ms = positive_delta_ms
        self.eps_ms = eps_ms
        self.pad_to_window = pad_to_window
        self.bank = bank
        self.rand = rand if rand is not None else random

    # ---- host decisions ---------------------------------------------------------------------------------------
    def _pick(self, ex: DeviceClip):
        """-> (label, start, length) of the window taken from one example."""
        rnd, n = self.rand, ex.num_samples
        window = int(self.sample_rate * self.window_size_ms / 1000)
        if not ex.timestamp_label_map:                       # no positive word: a random window of the clip
            if n < window:
                return self.negative_label, 0, n
            a = rnd.randint(0, n - window)
            return (self.negative_label,) + _clamped(a, a + window, n)
        take_negative = rnd.random() > self.positive_sample_prob
        if not take_negative:
            end_ms, label = rnd.choice(list(ex.timestamp_label_map.items()))
            jittered = end_ms + rnd.random() * self.eps_ms
            b = int((jittered / 1000) * self.sample_rate)
            a = max(b - int((self.window_size_ms / 1000) * self.sample_rate), 0)
            rnd.random()
            if b - a >= 0:
                return (label,) + _clamped(a, b, n)
        spans = sorted(((v - self.positive_delta_ms, v + self.positive_delta_ms)
                        for v in ex.timestamp_label_map.values()), key=lambda s: s[0])
        gaps, covered, hi = [], 0, 0
        for lo, hi in spans:
            if covered < lo:
                gaps.append((covered, lo))
            covered = hi
        gaps.append((hi, int(n / 16000 * 1000)))
        a, b = rnd.choice(gaps)
        if b - a > self.window_size_ms:
            a = rnd.randint(0, int(b - self.window_size_ms))
            b = a + self.window_size_ms
        return (self.negative_label,) + _clamped(a, b, n)

    def plan(self, examples: Sequence[DeviceClip]) -> WindowPlan:
        picks = [self._pick(ex) for ex in examples]
        lengths = np.array([p[2] for p in picks])
        order = np.argsort(-lengths)
        width = int(self.window_size_ms / 1000 * self.sample_rate) if self.pad_to_window else int(lengths.max())
        plan = WindowPlan([], [], [], [], [], width)
        for k in order:
            label, start, length = picks[k]
            if length > width:
                raise RuntimeError(f"window of {length} samples does not fit the padded width {width}")
            pad_front = self.rand.random() < 0.5
            plan.clip_id.append(examples[k].clip_id)
            plan.start.append(start)
            plan.length.append(length)
            plan.dst_off.append(width - length if pad_front else 0)
            plan.labels.append(label)
            plan.source.append(int(k))
        return plan

    # ---- device work ----------------------------------------------------------------------------------------------
    def __call__(self, examples: Sequence[DeviceClip]) -> ClassificationBatch:
        if self.bank is None:
            raise ValueError("WakeWordFrameBatchifier needs the device clip bank the examples index (bank=...)")
        plan = self.plan(examples)
        dev = self.bank.device
        i32 = lambda a: torch.tensor(a, dtype=torch.int32).to(dev, non_blocking=True)
        audio = ops.gather_windows(self.bank, i32(plan.clip_id), i32(plan.start), i32(plan.length), i32(plan.dst_off),
                                   plan.width)
        return ClassificationBatch(audio, torch.tensor(plan.labels).to(dev, non_blocking=True), torch.tensor(plan.length))


class AudioSequenceBatchifier:
    """CTC batches (``batchifier.py:13-34``): whole clips, longest first, zero padded on the right; labels are the
    tokenizer's ids of the transcription, padded with the negative label."""

    def __init__(self, negative_label: int, tokenizer, sample_rate: int = 16000, bank: torch.Tensor = None):
        self.negative_label, self.tokenizer, self.sample_rate, self.bank = negative_label, tokenizer, sample_rate, bank

    def order_and_labels(self, examples: Sequence[DeviceClip], lengths):
        """Batch order (``np.argsort(-lengths)``, ``operator.py:91-92``) and the padded label matrix in that order."""
        order = [int(k) for k in np.argsort(-np.asarray(lengths))]
        labels = [self.tokenizer.encode(examples[k].transcription) for k in order]
        smax = max(len(l) for l in labels)
        padded = torch.tensor([l + [self.negative_label] * (smax - len(l)) for l in labels], dtype=torch.long)
        return order, padded.reshape(len(order), smax), torch.tensor([len(l) for l in labels])

    def __call__(self, examples: Sequence[DeviceClip]) -> SequenceBatch:
        if self.bank is None:
            raise ValueError("AudioSequenceBatchifier needs the device clip bank the examples index (bank=...)")
        lengths = [ex.num_samples for ex in examples]
        order, padded, label_lengths = self.order_and_labels(examples, lengths)
        dev = self.bank.device
        i32 = lambda a: torch.tensor(a, dtype=torch.int32).to(dev, non_blocking=True)
        n = len(examples)
        audio = ops.gather_windows(self.bank, i32([examples[k].clip_id for k in order]), i32([0] * n),
                                   i32([lengths[k] for k in order]), i32([0] * n), max(lengths))
        return SequenceBatch(audio, padded, torch.tensor([lengths[k] for k in order]), label_lengths)
