"""``WordTranscriptSearcher`` (``howl/data/common/searcher.py:74-119``)."""
from howl_amd.settings import SETTINGS

from .tokenizer import WakeWordTokenizer
from .vocab import Vocab

__all__ = ["WordTranscriptSearcher"]


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

    def count_vocab(self, item: str, ignore_oov: bool = True) -> dict:
        counter = dict((self.vocab[i], 0) for i in range(len(self.vocab)))
        for e in self.tokenizer.encode(item):
            if ignore_oov and e == self.vocab.oov_token_id:
                continue
            counter[self.vocab[e]] += 1
        return counter
