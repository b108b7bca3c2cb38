"""``InferenceContext`` (``howl/context.py:14-125``): vocab -> label ids; sizes the model head.

Word-level tokens are implemented (all BASELINE configs use ``TOKEN_TYPE='word'``); the phone-level branch needs the
pronunciation-dictionary / phone plumbing that SURVEY 2 lists as out of scope, and raises.
"""
import logging
from typing import List

from howl_amd.data.common.labeler import WordFrameLabeler
from howl_amd.data.common.searcher import WordTranscriptSearcher
from howl_amd.data.common.tokenizer import TokenType
from howl_amd.data.common.vocab import Vocab

__all__ = ["InferenceContext"]


class InferenceContext:
    def __init__(self, vocab: List[str], sequence: List[int] = None, token_type: str = TokenType.PHONE,
                 phone_dictionary_path: str = None, seed: int = 0, use_blank: bool = False):
        self.seed = seed
        self.sequence = sequence if sequence is not None else range(len(vocab))
        self.phone_dictionary_path = phone_dictionary_path
        self.coloring = None
        self.adjusted_vocab = []
        self.num_labels = 0
        self.token_type = token_type
        self.pronounce_dict = None
        if token_type != "word":
            raise NotImplementedError("InferenceContext: only token_type='word' is on the MI355X hot path "
                                      "(context.py:52-61 phone branch needs the pronunciation dictionary stack)")
        self.add_vocab(vocab)
        self.negative_label = len(self.adjusted_vocab)
        self.vocab = Vocab({word: idx for idx, word in enumerate(self.adjusted_vocab)},
                           oov_token_id=self.negative_label)
        self.labeler = WordFrameLabeler(self.vocab)
        self.add_vocab(["[OOV]"])
        self.searcher = WordTranscriptSearcher(self.vocab)
        self.blank_label = -1
        if use_blank:
            self.blank_label = len(self.adjusted_vocab)
            self.add_vocab(["[BLANK]"])
        for idx, word in enumerate(self.adjusted_vocab):
            logging.info(f"target {word:10} is assigned to label {idx}")

    def add_vocab(self, vocabs: List[str]):
        for v in vocabs:
            self.adjusted_vocab.append(v)
        self.num_labels += len(vocabs)

    @property
    def wake_word(self):
        return self.vocab.wakeword(self.sequence)
