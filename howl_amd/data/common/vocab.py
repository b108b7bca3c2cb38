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
    def __init__(self, word2idx: Union[Mapping[str, int], List[str]], oov_token_id: int = None,
                 oov_word_repr: str = "[OOV]"):
        if isinstance(word2idx, list):
            word2idx = {word: idx for idx, word in enumerate(word2idx)}
        self.word2idx = {k.lower(): v for k, v in word2idx.items()}
        self.idx2word = {v: k for k, v in word2idx.items()}
        self.oov_token_id = oov_token_id
        self.oov_word_repr = oov_word_repr
        self.trie = VocabTrie()
        for word in self.word2idx:
            self.trie.add_word(word.lower())

    def __len__(self):
        return len(self.word2idx)

    def __getitem__(self, item: Union[str, int]) -> Union[str, int]:
        ret = self.word2idx.get(item.lower(), self.oov_token_id) if isinstance(item, str) else \
            self.idx2word.get(item, self.oov_word_repr)
        if ret is None:
            raise ValueError(f"couldn't find token for {item}")
        return ret

    def wakeword(self, sequence: List[int], separator: str = " "):
        return separator.join([self[i] for i in sequence])

    def __repr__(self):
        return str(self.idx2word)
