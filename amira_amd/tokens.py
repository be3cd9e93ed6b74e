"""Gene strings <-> integer tokens.

The reference orders gene-mers by the signed 256-bit hash of each gene,
``int(sha256(pickle.dumps(name)).hexdigest(), 16) * strand`` (construct_gene.py:5-10,
:91-93).  Hash every distinct gene name ONCE on the host with the same hashlib / pickle
calls, rank the names by hash (ascending) and set

    token('+', g) = V + rank(g)        token('-', g) = V - 1 - rank(g)

Then integer order of tokens == order of the signed hashes, strand flip ==
``2V - 1 - token`` and the reverse complement of a k-tuple is the reversed tuple of
flipped tokens — which is all the device needs.
"""
import hashlib
import pickle
from itertools import chain

import numpy as np


def name_hash(name):
    """construct_gene.py:5-10 on a gene name."""
    return int(hashlib.sha256(pickle.dumps(name)).hexdigest(), 16)


def split_gene(gene):
    """construct_gene.py:49-65: strand = first character, name = rest with ' ' -> '_'."""
    assert gene.replace(" ", "") != "", "Gene information is missing"
    sign, name = gene[0], gene[1:].replace(" ", "_")
    assert sign == "-" or sign == "+", "Strand information missing for: " + gene
    assert name != "", "Gene name information missing for: " + gene
    return name, (1 if sign == "+" else -1)


class Vocabulary:
    """The distinct gene names of a read set, ranked by the reference's gene hash."""

    def __init__(self, names):
        names = list(dict.fromkeys(names))
        hashes = {n: name_hash(n) for n in names}
        self.names = sorted(names, key=hashes.__getitem__)  # rank -> name
        self.hashes = [hashes[n] for n in self.names]       # rank -> 256-bit int
        self.rank = {n: i for i, n in enumerate(self.names)}
        self.V = len(self.names)
        self.two_v = 2 * max(self.V, 1)
        self._tok = {}
        V = max(self.V, 1)
        for n, r in self.rank.items():
            self._tok["+" + n] = V + r
            self._tok["-" + n] = V - 1 - r

    @classmethod
    def from_ranked(cls, names, digests):
        """names already in rank order with their 32-byte sha256 digests (native loader)."""
        self = cls.__new__(cls)
        self.names = list(names)
        self.hashes = [int.from_bytes(bytes(d), "big") for d in digests]
        self.rank = {n: i for i, n in enumerate(self.names)}
        self.V = len(self.names)
        self.two_v = 2 * max(self.V, 1)
        V = max(self.V, 1)
        self._tok = {}
        for n, r in self.rank.items():
            self._tok["+" + n] = V + r
            self._tok["-" + n] = V - 1 - r
        return self

    # ---- encoding
    def token(self, gene):
        t = self._tok.get(gene)
        if t is None:  # spaces / malformed strings take the slow, asserting path
            name, strand = split_gene(gene)
            V = max(self.V, 1)
            t = V + self.rank[name] if strand == 1 else V - 1 - self.rank[name]
        return t

    def gene(self, token):
        V = max(self.V, 1)
        return ("+" + self.names[token - V]) if token >= V else ("-" + self.names[V - 1 - token])

    def signed_hash(self, token):
        V = max(self.V, 1)
        return self.hashes[token - V] if token >= V else -self.hashes[V - 1 - token]

    def strand(self, token):
        return 1 if token >= max(self.V, 1) else -1

    def name_of(self, token):
        V = max(self.V, 1)
        return self.names[token - V] if token >= V else self.names[V - 1 - token]

    def flip(self, token):
        return self.two_v - 1 - token

    def decode(self, tokens):
        """strings of a token sequence: one table lookup per token, built once (token -> "+name" / "-name")"""
        table = getattr(self, "_strings", None)
        if table is None:
            V = max(self.V, 1)
            table = self._strings = (["-" + n for n in reversed(self.names)] + [""] * (V - self.V)
                                     + ["+" + n for n in self.names])
        if hasattr(tokens, "tolist"):
            tokens = tokens.tolist()
        return [table[t] for t in tokens]


def tokenize(read_dict):
    """dict[read_id -> ["+geneA", "-geneB", ...]] -> (Vocabulary, tokens int32, read_offsets
    int64, read_ids list) in the dict's iteration order (== the reference's build order,
    construct_graph.py:45)."""
    lists = list(read_dict.values())
    seen = set(chain.from_iterable(lists))          # distinct "+name" / "-name" strings (C loops:
    names = {g: split_gene(g)[0] for g in seen}    # the per-gene work stays out of the interpreter)
    vocab = Vocabulary(names.values())
    table = {g: vocab.token(g) for g in seen}
    read_ids = list(read_dict.keys())
    offs = np.zeros(len(read_ids) + 1, dtype=np.int64)
    np.cumsum(list(map(len, lists)), out=offs[1:])
    toks = np.fromiter(map(table.__getitem__, chain.from_iterable(lists)), dtype=np.int32,
                       count=int(offs[-1]))
    return vocab, toks, offs, read_ids


def synthetic_vocabulary(names):
    return Vocabulary(names)
