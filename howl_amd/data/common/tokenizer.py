"""``howl/data/common/tokenizer.py``: token types and the word-level wake-word tokenizer (CTC targets)."""
from enum import Enum, unique
from typing import List

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

    def encode(self, transcript: str) -> List[int]:
        out = []
        for word in transcript.lower().split():
            found, rest = self.vocab.trie.max_split(word)
            if found and rest == "":
                out.append(self.vocab[word])
            elif not self.ignore_oov:
                if self.vocab.oov_token_id is None:
                    raise ValueError("label for oov word is not specified")
                out.append(self.vocab.oov_token_id)
        return out

    def decode(self, ids: List[int]) -> str:
        return " ".join(self.vocab[i] for i in ids)
