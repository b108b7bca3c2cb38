"""Frame labels: word level (``howl/data/common/labeler.py:156-182``, ``label.py``) and phone level (``labeler.py:28-153``)."""
import string
from dataclasses import dataclass
from pathlib import Path
from typing import Dict, List, Optional, Tuple

from .phone import PhoneEnum, PhonePhrase, PronunciationDictionary
from .vocab import Vocab

__all__ = ["FrameLabelData", "WordFrameLabeler", "PhoneticFrameLabeler"]


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


# typographic quotes / dashes / a few accented vowels that show up in transcripts -> their dictionary spelling
_LOOKALIKES = str.maketrans("\u2018\u2019\u201d\u201c\u2014\u00e4\u00f6\u014d\u00e9\u00e0", "''\"\"-aooea")
_NO_PUNCTUATION = str.maketrans("", "", string.punctuation)


class PhoneticFrameLabeler:
    """Labels the frames at which one of ``phrases`` (the target phone sequences) ends.  A transcript word is looked up in the
    pronunciation dictionary as it is, then with look-alike characters replaced, then with punctuation removed; a word that
    is not an entry is split greedily into the longest entries (``labeler.py:49-91``)."""

    def __init__(self, phrases: List[PhonePhrase], pronounce_dict: Optional[PronunciationDictionary] = None):
        self.phrases = phrases
        if pronounce_dict is None:
            from howl_amd.settings import SETTINGS
            pronounce_dict = PronunciationDictionary.from_file(Path(SETTINGS.training.phone_dictionary))
        self.pronounce_dict = pronounce_dict
        self.punctuation_transforms = [None, _LOOKALIKES, _NO_PUNCTUATION]

    def transform(self, original_word: str) -> PhonePhrase:
        """Greedy split into the longest dictionary entries.  ``<unk>`` (when it is not an entry itself) contributes the
        inaudible ``spn`` -- and, exactly as in the reference's loop, scanning then resumes at its LAST character (``>``),
        which raises unless that is an entry: ``compute_frame_labels`` catches it and retries with punctuation removed."""
        out = PhonePhrase([])
        rest = original_word
        while rest:
            cut = len(rest)
            while cut > 0 and rest[:cut] not in self.pronounce_dict:
                cut -= 1
            if cut > 0:
                out.extend(self.pronounce_dict.encode(rest[:cut])[0])     # first pronunciation only, like the reference
                rest = rest[cut:]
            elif rest == "<unk>":
                out.extend(PhonePhrase.from_string(PhoneEnum.SPEECH_UNKNOWN.value))
                rest = rest[-1:]
            else:
                raise ValueError("word is not in the dictionary: ")
        return out

    def _phones_of(self, word: str) -> Optional[PhonePhrase]:
        for table in self.punctuation_transforms:
            if table is not None:
                word = word.translate(table)
                if not word:
                    return None
            try:
                return self.transform(word)
            except ValueError:
                continue
        print(f"Failed to find phonemes for {word} {[ord(ch) for ch in word]}")
        return None

    def compute_frame_labels(self, metadata) -> FrameLabelData:
        spoken = PhonePhrase([])
        for word in metadata.transcription.split():
            phones = self._phones_of(word)
            if phones:
                spoken.extend(phones)
        frame_labels: Dict[float, int] = {}
        for label, phrase in enumerate(self.phrases):
            at = 0
            while True:
                try:
                    at = spoken.audible_index(phrase, at)
                except ValueError:
                    break
                # as in the reference (labeler.py:137-150): the phrase's audible START index is used as the index into the
                # per-character end timestamps
                frame_labels[metadata.end_timestamps[at]] = label
                at += 1
        return FrameLabelData(frame_labels, [], [])
