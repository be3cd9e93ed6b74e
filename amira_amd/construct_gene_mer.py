"""GeneMer value object — drop-in for amira/construct_gene_mer.py (reference v0.11.0).

The canonical orientation rule (lexicographically smaller list of signed gene hashes,
construct_gene_mer.py:15-39) is what kernel k_node_upsert evaluates on integer tokens;
this class is the host-side object view of one window / node.
"""
from .construct_gene import Gene, hashlib_hash


def define_rc_geneMer(geneMer):
    """reverse complement: reversed order, every strand flipped (:4-12)."""
    assert all(isinstance(gene, Gene) for gene in geneMer)
    return [gene.reverse_gene() for gene in geneMer[::-1]]


def sort_geneMers(geneMer, rcGeneMer):
    """hash lists of both orientations and their sorted pair (:15-28)."""
    fwd = [g.__hash__() for g in geneMer]
    rev = [g.__hash__() for g in rcGeneMer]
    assert not (fwd == rev), "Gene-mer and reverse complement gene-mer are identical"
    return fwd, rev, sorted([fwd, rev])


def choose_canonical_geneMer(geneMer, geneMerHashes, rcGeneMer, rcGeneMerHashes, sortedGeneMerhashes):
    """(canonical, other) by position in the sorted pair (:31-39)."""
    if sortedGeneMerhashes[0] == geneMerHashes and sortedGeneMerhashes[1] == rcGeneMerHashes:
        return geneMer, rcGeneMer
    if sortedGeneMerhashes[0] == rcGeneMerHashes and sortedGeneMerhashes[1] == geneMerHashes:
        return rcGeneMer, geneMer


def define_geneMer(geneMer):
    """(:42-56)"""
    assert isinstance(geneMer, list), "Gene-mer is not a list of Gene objects"
    assert not geneMer == [], "Gene-mer is empty"
    rc = define_rc_geneMer(geneMer)
    fwd_h, rc_h, ordered = sort_geneMers(geneMer, rc)
    return choose_canonical_geneMer(geneMer, fwd_h, rc, rc_h, ordered)


class GeneMer:
    def __init__(self, geneMer):
        self.canonicalGeneMer, self.rcGeneMer = define_geneMer(geneMer)
        self.geneMerSize = len(self.canonicalGeneMer)
        self.geneMerDirection = 1 if self.canonicalGeneMer == geneMer else -1
        self._hash = None

    @classmethod
    def _from_parts(cls, canonical, rc, direction, node_hash=None):
        """Object view of a device node: canonical / rc gene lists are already known."""
        self = cls.__new__(cls)
        self.canonicalGeneMer, self.rcGeneMer = canonical, rc
        self.geneMerSize = len(canonical)
        self.geneMerDirection = direction
        self._hash = node_hash
        return self

    def get_canonical_geneMer(self):
        return self.canonicalGeneMer

    def get_rc_geneMer(self):
        return self.rcGeneMer

    def get_geneMerDirection(self):
        return self.geneMerDirection

    def get_geneMer_size(self):
        return self.geneMerSize

    def __eq__(self, otherGeneMer):
        return (self.canonicalGeneMer == otherGeneMer.get_canonical_geneMer()
                and self.rcGeneMer == otherGeneMer.get_rc_geneMer())

    def __hash__(self):
        """sha256 over the pickled tuple of signed canonical gene hashes (:94-97)."""
        if self._hash is None:
            self._hash = hashlib_hash(tuple([g.__hash__() for g in self.canonicalGeneMer]))
        return self._hash
