"""CTC target encoding for the sequence objective (``train.py:250-256`` feeds ``CTCLoss`` with these ids).

Same observable behaviour as ``howl/data/common/tokenizer.py``'s word tokenizer -- a transcript word contributes its label when
it is, as a whole, a vocabulary word, otherwise the OOV label (or nothing under ``ignore_oov``) -- built differently: the
vocabulary is flattened once into a lower-cased word -> id table, so encoding a transcript is one dictionary lookup per word
(the batchifier encodes every transcript of every batch on the host while the device is busy)."""
from enum import Enum, unique
from typing import Dict, Iterable, List, Optional

from .vocab import Vocab

__all__ = ["TokenType", "WakeWordTokenizer"]


@unique
class TokenType(str, Enum):
    PHONE = "phone"
    WORD = "word"


class WakeWordTokenizer:
    def __init__(self, vocab: Vocab, ignore_oov: bool = True):
        self.vocab = vocab
        self.ignore_oov = ignore_oov
        self._table: Dict[str, int] = dict(vocab.word2idx)
        self._fallback: Optional[int] = None if ignore_oov else vocab.oov_token_id
        self._strict = not ignore_oov and vocab.oov_token_id is None

    def _lookup(self, words: Iterable[str]) -> Iterable[int]:
        table, fallback = self._table, self._fallback
        for word in words:
            label = table.get(word, fallback)
            if label is not None:
                yield label
            elif self._strict:
                raise ValueError("label for oov word is not specified")

    def encode(self, transcript: str) -> List[int]:
        return list(self._lookup(transcript.lower().split()))

    def encode_batch(self, transcripts: Iterable[str]) -> List[List[int]]:
        return [self.encode(t) for t in transcripts]

    def decode(self, ids: Iterable[int]) -> str:
        return " ".join(self.vocab.word_of(i) for i in ids)
