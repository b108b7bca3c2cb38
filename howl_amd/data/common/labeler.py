"""Word-level frame labels (``howl/data/common/labeler.py:156-182``, ``label.py``)."""
from dataclasses import dataclass
from typing import Dict, List, Tuple

from .vocab import Vocab

__all__ = ["FrameLabelData", "WordFrameLabeler"]


@dataclass
class FrameLabelData:
    timestamp_label_map: Dict[float, int]
    start_timestamp: List[Tuple[int, float]]
    char_indices: List[Tuple[int, List[int]]]


class WordFrameLabeler:
    def __init__(self, vocab: Vocab):
        self.vocab = vocab

    def compute_frame_labels(self, metadata) -> FrameLabelData:
        """``metadata`` needs ``.transcription`` and ``.end_timestamps`` (per character, ms)."""
        frame_labels, start_timestamp, char_indices = dict(), [], []
        char_idx = 0
        for word in metadata.transcription.split():
            found, rest = self.vocab.trie.max_split(word)
            word_size = len(word.rstrip())
            if found and rest == "":
                label = self.vocab[word]
                frame_labels[metadata.end_timestamps[char_idx + word_size - 1]] = label
                char_indices.append((label, list(range(char_idx, char_idx + word_size))))
                start_timestamp.append((label, metadata.end_timestamps[char_idx - 1] if char_idx > 0 else 0.0))
            char_idx += word_size + 1
        return FrameLabelData(frame_labels, start_timestamp, char_indices)
