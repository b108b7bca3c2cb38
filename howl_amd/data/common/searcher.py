"""Transcript searchers: ``WordTranscriptSearcher`` (``howl/data/common/searcher.py:74-119``), ``LabelColoring`` and
``PhoneticTranscriptSearcher`` (``searcher.py:20-58, 122-144``)."""
import logging
import re
from typing import Dict, List, Optional

from howl_amd.settings import SETTINGS

from .phone import PhonePhrase
from .tokenizer import WakeWordTokenizer
from .vocab import Vocab

__all__ = ["LabelColoring", "PhoneticTranscriptSearcher", "WordTranscriptSearcher"]


class WordTranscriptSearcher:
    def __init__(self, vocab: Vocab):
        self.settings = SETTINGS.inference_engine
        self.vocab = vocab
        self.tokenizer = WakeWordTokenizer(self.vocab, False)
        self.inference_sequence_str = "".join(map(str, self.settings.inference_sequence))

    def search(self, item: str) -> bool:
        return self.inference_sequence_str in "".join(map(str, self.tokenizer.encode(item)))

    def contains_any(self, item: str) -> bool:
        return any(e != self.vocab.oov_token_id for e in self.tokenizer.encode(item))


class LabelColoring:
    """label -> colour: labels that share a colour are alternatives for the same position of the wake sequence (the phones
    of one word)."""

    def __init__(self):
        self.color_map: Dict[int, int] = {}
        self.color_counter = 0
        self.label_counter = 0

    def _take_color(self, color: Optional[int]) -> int:
        if color is None:
            color = self.color_counter
        else:
            self.color_counter = max(self.color_counter, color)
        self.color_counter += 1
        return color

    def append_label(self, label: int, color: int = None):
        known = self.color_map.get(label)
        if known is not None:
            if color is not None and color != known:
                raise RuntimeError(f"given label {label} is already registered with color {known} "
                                   f"which mismatches with the given color {color}")
            return
        self.color_map[label] = self._take_color(color)
        self.label_counter = max(self.label_counter, label + 1)

    def extend_sequence(self, size: int, color: int = None):
        color = self._take_color(color)
        for label in range(self.label_counter, self.label_counter + size):
            self.color_map[label] = color
        self.label_counter += size

    @classmethod
    def sequential_coloring(cls, num_labels: int) -> "LabelColoring":
        coloring = cls()
        for label in range(num_labels):
            coloring.append_label(label)
        return coloring


class PhoneticTranscriptSearcher:
    """Wake sequence as a regular expression over audible phone transcripts: one group of alternatives per colour, the
    groups of ``SETTINGS.inference_engine.inference_sequence`` in order, separated by single spaces."""

    def __init__(self, phrases: List[PhonePhrase], coloring: LabelColoring):
        self.settings = SETTINGS.inference_engine
        self.phrases = phrases
        by_color: Dict[int, List[str]] = {}
        for label, phrase in enumerate(phrases):
            by_color.setdefault(coloring.color_map[label], []).append(phrase.audible_transcript)
        groups = ["(" + "|".join(f"({t})" for t in by_color[color]) + ")" for color in sorted(by_color)]
        pattern = "^.*" + " ".join(groups[i] for i in self.settings.inference_sequence) + ".*$"
        logging.info(f"Using search pattern {pattern}")
        self.pattern = re.compile(pattern)

    def search(self, item: str) -> bool:
        return self.pattern.match(PhonePhrase.from_string(item).audible_transcript) is not None

    def contains_any(self, item: str) -> bool:
        transcript = PhonePhrase.from_string(item).audible_transcript
        return any(phrase.audible_transcript in transcript for phrase in self.phrases)
