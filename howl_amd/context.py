"""``InferenceContext`` (``howl/context.py:14-125``): vocab -> label ids; sizes the model head.

Both token types of the reference: ``word`` (all BASELINE configs) and ``phone`` (``context.py:52-61``: every vocab word is
replaced by the phones of its first dictionary pronunciation, one colour per word).
"""
import logging
from pathlib import Path
from typing import List

from howl_amd.data.common.labeler import PhoneticFrameLabeler, WordFrameLabeler
from howl_amd.data.common.phone import PhonePhrase, PronunciationDictionary
from howl_amd.data.common.searcher import LabelColoring, PhoneticTranscriptSearcher, WordTranscriptSearcher
from howl_amd.settings import SETTINGS
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
        if token_type not in ("word", "phone"):
            raise ValueError(f"InferenceContext: unknown token_type {token_type!r} (word | phone)")
        phone = token_type == "phone"
        if phone:
            self.pronounce_dict = PronunciationDictionary.from_file(Path(SETTINGS.training.phone_dictionary))
            self.coloring = LabelColoring()
            for word in vocab:
                phrase = self.pronounce_dict.encode(word)[0]       # single pronunciation, as the reference
                logging.info(f"Word {word: <10} has phonemes of {str(phrase)}")
                self.add_vocab([str(ph) for ph in phrase.phones])
        else:
            self.add_vocab(vocab)
        self.negative_label = len(self.adjusted_vocab)
        self.vocab = Vocab({word: idx for idx, word in enumerate(self.adjusted_vocab)},
                           oov_token_id=self.negative_label)
        # the labeler sees the targets only: built before the [OOV] / [BLANK] labels are appended
        if phone:
            phrases = [PhonePhrase.from_string(x) for x in self.adjusted_vocab]
            self.labeler = PhoneticFrameLabeler(phrases, self.pronounce_dict)
        else:
            self.labeler = WordFrameLabeler(self.vocab)
        self.add_vocab(["[OOV]"])
        self.searcher = PhoneticTranscriptSearcher(phrases, self.coloring) if phone else WordTranscriptSearcher(self.vocab)
        self.blank_label = -1
        if use_blank:
            self.blank_label = len(self.adjusted_vocab)
            self.add_vocab(["[BLANK]"])
        for idx, word in enumerate(self.adjusted_vocab):
            logging.info(f"target {word:10} is assigned to label {idx}")

    def add_vocab(self, vocabs: List[str]):
        self.adjusted_vocab.extend(vocabs)
        if self.coloring:
            self.coloring.extend_sequence(len(vocabs))
        self.num_labels += len(vocabs)

    @property
    def wake_word(self):
        return self.vocab.wakeword(self.sequence)
