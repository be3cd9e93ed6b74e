"""Import-only stand-in for the PyPI package ``suffix-tree==0.1.2`` (absent here, no network).

Used ONLY by tests/golden/gen_goldens.py in the build container to import the
reference; it never travels into the product or the oracle.  It implements the two
calls the reference makes (Tree(dict) and Tree.find_all(seq)) by brute force with
the published semantics: find_all yields (id, path) for every occurrence of seq in
every stored sequence, and str(path) is the space-joined suffix from the match
start followed by the end marker "$".
"""


class _Path:
    __slots__ = ("_items",)

    def __init__(self, items):
        self._items = items

    def __str__(self):
        return " ".join([str(x) for x in self._items] + ["$"])

    def __len__(self):
        return len(self._items) + 1


class Tree:
    def __init__(self, data=None):
        self._seqs = {}
        if data:
            for key, seq in data.items():
                self._seqs[key] = list(seq)

    def add(self, key, seq):
        self._seqs[key] = list(seq)

    def find_all(self, query):
        query = list(query)
        n = len(query)
        out = []
        for key, seq in self._seqs.items():
            for i in range(len(seq) - n + 1):
                if seq[i : i + n] == query:
                    out.append((key, _Path(seq[i:])))
        return out

    def find(self, query):
        return len(self.find_all(query)) > 0
