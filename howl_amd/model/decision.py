"""Host-side decision logic of the inference engines, separated from the device work: temporal smoothing of class
probabilities and the wake-sequence matcher.  Behaviour follows ``howl/model/inference.py:91-161`` (label FSM over a
sliding time window, max-over-window smoothing with threshold and optional label colouring); the golden label histories
(tests/golden G8) pin it."""
from collections import deque
from typing import List, Optional, Sequence, Tuple

import numpy as np


def _drop_older_than(items: List[Tuple[float, object]], now: float, horizon_ms: float):
    """Removes the leading entries that are more than ``horizon_ms`` older than ``now`` (entries are time-ordered)."""
    keep_from = 0
    for stamp, _ in items:
        if now - stamp > horizon_ms:
            keep_from += 1
        else:
            break
    if keep_from:
        del items[:keep_from]


class ProbabilitySmoother:
    """Keeps (time, probability vector) frames of the last ``window_ms`` and turns them into one label per frame."""

    def __init__(self, window_ms: float, threshold: float, negative_label: int, color_map: Optional[dict] = None):
        self.window_ms, self.threshold = window_ms, threshold
        self.negative_label, self.color_map = negative_label, color_map
        self.frames: List[Tuple[float, np.ndarray]] = []

    def clear(self):
        self.frames = []

    def push(self, now: float, probs: np.ndarray) -> int:
        self.frames.append((now, probs))
        _drop_older_than(self.frames, now, self.window_ms)
        envelope = np.max(np.vstack([p for _, p in self.frames]), axis=0)   # per-class maximum over the window
        label = int(envelope.argmax())
        confident = envelope[label] >= self.threshold
        if self.color_map is not None:      # a colouring is configured (even an empty map recolours every label)
            label = self.color_map.get(label, self.negative_label)
        return label if confident else self.negative_label


class SequenceMatcher:
    """Looks for ``sequence`` (label indices, in order) in a time-stamped label history.  Repeats of the label matched last
    keep a partial match alive; any other label breaks it once ``tolerance_ms`` have passed since the last useful frame."""

    def __init__(self, sequence: Sequence[int], window_ms: float, tolerance_ms: float):
        self.sequence, self.window_ms, self.tolerance_ms = sequence, window_ms, tolerance_ms

    def present(self, history: List[Tuple[float, int]], now: float) -> bool:
        if not self.sequence:
            return False
        _drop_older_than(history, now, self.window_ms)
        matched = 0            # labels of the sequence seen so far
        anchor = 0.0           # time of the last frame that advanced or sustained the match
        holding = None         # label whose repetition sustains the match
        for stamp, label in history:
            if label == self.sequence[matched]:
                matched += 1
                if matched == len(self.sequence):
                    return True
                holding, anchor = label, stamp
            elif label == holding:
                anchor = stamp
            elif anchor + self.tolerance_ms < stamp:
                matched, anchor, holding = 0, 0.0, None
        return False
