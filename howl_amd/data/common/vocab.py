"""Word vocabulary with the trie-based longest-match split of ``howl/data/common/vocab.py:6-102``."""
from typing import List, Mapping, Tuple, Union

__all__ = ["Vocab", "VocabTrie"]


class VocabTrie:
    class Node:
        def __init__(self, terminal: bool = True):
            self.terminal = terminal
            self.children = {}

    def __init__(self):
        self.root = VocabTrie.Node(terminal=False)

    def _nearest_node(self, word: str, node):
        while word and word[0] in node.children:
            node, word = node.children[word[0]], word[1:]
        return node, word

    def add_word(self, word: str):
        node, left = self._nearest_node(word.lower(), self.root)
        for ch in left:
            nxt = VocabTrie.Node(terminal=False)
            node.children[ch] = nxt
            node = nxt
        node.terminal = True

    def max_split(self, tokens: str) -> Tuple[str, str]:
        """Longest prefix that walks the trie; empty prefix unless it ends on a terminal node (``vocab.py:50-61``)."""
        node, counter = self.root, 0
        for tok in tokens.lower():
            if tok not in node.children:
                break
            node = node.children[tok]
            counter += 1
        if not node.terminal:
            counter = 0
        return tokens[:counter], tokens[counter:]


class Vocab:
    """Word <-> label-id table behind ``InferenceContext`` (interface of ``howl/data/common/vocab.py:64-102``:
    ``len()``, ``vocab[word]`` / ``vocab[idx]``, ``.trie``, ``.oov_token_id``, ``.wakeword()``).

    Lookups are case-insensitive on the word side.  An unknown word maps to ``oov_token_id`` (``ValueError`` when no
    OOV id was configured); an unknown id renders as ``oov_word_repr``."""

    def __init__(self, word2idx: Union[Mapping[str, int], List[str]], oov_token_id: int = None,
                 oov_word_repr: str = "[OOV]"):
        pairs = list(word2idx.items()) if isinstance(word2idx, Mapping) else [(w, i) for i, w in enumerate(word2idx)]
        self.oov_token_id, self.oov_word_repr = oov_token_id, oov_word_repr
        self.word2idx, self.idx2word, self.trie = {}, {}, VocabTrie()
        for word, idx in pairs:
            self.word2idx[word.lower()] = idx
            self.idx2word[idx] = word            # ids render with the caller's spelling
            self.trie.add_word(word)

    def __len__(self):
        return len(self.word2idx)

    def id_of(self, word: str) -> int:
        idx = self.word2idx.get(word.lower(), self.oov_token_id)
        if idx is None:
            raise ValueError(f"couldn't find token for {word}")
        return idx

    def word_of(self, idx: int) -> str:
        return self.idx2word.get(idx, self.oov_word_repr)

    def __getitem__(self, item: Union[str, int]) -> Union[str, int]:
        return self.id_of(item) if isinstance(item, str) else self.word_of(item)

    def wakeword(self, sequence: List[int], separator: str = " ") -> str:
        return separator.join(self.word_of(i) for i in sequence)

    def __repr__(self):
        return repr(self.idx2word)
