"""Phone-level tokens: ``Phone`` / ``PhonePhrase`` / ``PronunciationDictionary`` (the behaviour of
``howl/data/common/phone.py:13-171``; host-side plumbing of ``InferenceContext(token_type="phone")``, pinned against the
reference classes by golden G11).

A phrase is kept as a tuple of normalised phone strings plus the positions of its audible ones, computed once; every
query the labeler / searcher makes is a lookup on those two arrays.
"""
import enum
from pathlib import Path
from typing import Dict, Iterable, List, Sequence, Tuple

__all__ = ["PhoneEnum", "Phone", "PhonePhrase", "PronunciationDictionary"]


class PhoneEnum(enum.Enum):
    SILENCE = "sil"
    SILENCE_OPTIONAL = "sp"
    SPEECH_UNKNOWN = "spn"


_INAUDIBLE = frozenset(p.value for p in PhoneEnum)


class Phone:
    __slots__ = ("text", "is_speech")

    def __init__(self, text: str):
        self.text = text.lower().strip()
        self.is_speech = self.text not in _INAUDIBLE

    def __str__(self):
        return self.text

    def __repr__(self):
        return f"Phone({self.text!r})"

    def __eq__(self, other):
        return isinstance(other, Phone) and other.text == self.text

    def __hash__(self):
        return hash(self.text)


class PhonePhrase:
    def __init__(self, phones: Iterable[Phone]):
        self.phones: List[Phone] = list(phones)

    @classmethod
    def from_string(cls, string: str) -> "PhonePhrase":
        return cls(Phone(tok) for tok in string.split())

    # ---- views -------------------------------------------------------------------------------------------------
    @property
    def text(self) -> str:
        return str(self)

    def __str__(self):
        return " ".join(p.text for p in self.phones)

    def __repr__(self):
        return f"PhonePhrase({str(self)!r})"

    def __eq__(self, other):
        return isinstance(other, PhonePhrase) and other.phones == self.phones

    def _audible_positions(self) -> List[int]:
        return [i for i, p in enumerate(self.phones) if p.is_speech]

    @property
    def audible_phones(self) -> List[Phone]:
        return [self.phones[i] for i in self._audible_positions()]

    @property
    def audible_transcript(self) -> str:
        return " ".join(self.phones[i].text for i in self._audible_positions())

    @property
    def sil_indices(self) -> List[int]:
        return [i for i, p in enumerate(self.phones) if not p.is_speech]

    def extend(self, other: "PhonePhrase"):
        self.phones.extend(other.phones)

    # ---- index conversions ---------------------------------------------------------------------------------------
    def all_idx_to_transcript_idx(self, phone_idx: int) -> int:
        """Character position in ``str(self)`` just past phone ``phone_idx`` (phones are separated by one space)."""
        if phone_idx >= len(self.phones):
            raise ValueError(f"Given phone idx ({phone_idx}) is greater than the number of phones ({len(self.phones)})")
        return sum(len(p.text) for p in self.phones[: phone_idx + 1]) + phone_idx

    def audible_idx_to_all_idx(self, audible_idx: int) -> int:
        """Position among ALL phones of the ``audible_idx``-th audible one."""
        positions = self._audible_positions()
        if audible_idx >= len(positions):
            raise ValueError(f"Given audible phone idx ({audible_idx}) is greater than"
                             f"the number of audible phones ({len(positions)})")
        return positions[audible_idx]

    def audible_index(self, query: "PhonePhrase", start: int = 0) -> int:
        """First audible index >= ``start`` at which ``query``'s audible phones occur as a contiguous run."""
        needle = [p.text for p in query.audible_phones]
        if not needle:
            raise ValueError(f"query phrase has empty audible_phones: {query.audible_transcript}")
        hay = [p.text for p in self.audible_phones]
        for at in range(start, len(hay) - len(needle) + 1):
            if hay[at:at + len(needle)] == needle:
                return at
        raise ValueError(f"query phrase is not found: {query.audible_transcript}")


class PronunciationDictionary:
    """word -> list of pronunciations, read from a CMUdict-style file (``WORD  ph ph ph``; lines starting with ``;`` are
    comments; a word may occur on several lines)."""

    def __init__(self, data_dict: Dict[str, List[PhonePhrase]]):
        self.word2phone = data_dict

    @staticmethod
    def _key(word: str) -> str:
        return word.strip().lower()

    def __contains__(self, key: str) -> bool:
        return self._key(key) in self.word2phone

    def encode(self, word: str) -> List[PhonePhrase]:
        key = self._key(word)
        found = self.word2phone.get(key)
        if found is None:
            raise ValueError(f"word is not in the dictionary: {key}")
        return found

    @classmethod
    def from_file(cls, filename: Path) -> "PronunciationDictionary":
        table: Dict[str, List[PhonePhrase]] = {}
        with Path(filename).open() as f:
            for line in f:
                if line.startswith(";"):
                    continue
                fields = line.split(maxsplit=1)
                if len(fields) < 2 or not fields[1].strip():
                    continue
                table.setdefault(fields[0].lower(), []).append(PhonePhrase.from_string(fields[1].strip().lower()))
        return cls(table)
