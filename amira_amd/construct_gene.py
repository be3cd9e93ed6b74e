"""Gene value object — drop-in for amira/construct_gene.py (reference v0.11.0).

Host-side only: a gene's 256-bit hash is computed here with hashlib/pickle exactly as
the reference does (construct_gene.py:5-10); the device never hashes strings, it works
on integer tokens whose order equals the order of these hashes (amira_amd.tokens).
"""
import hashlib
import pickle

_SIGN = {"+": 1, "-": -1}
_CHAR = {1: "+", -1: "-"}


def hashlib_hash(value):
    """int(sha256(pickle.dumps(value)).hexdigest(), 16) — construct_gene.py:5-10."""
    return int.from_bytes(hashlib.sha256(pickle.dumps(value)).digest(), "big")


def convert_string_strand_to_int(stringStrand):
    assert stringStrand == "+" or stringStrand == "-"
    return _SIGN[stringStrand]


def reverse_strand(geneStrand):
    assert geneStrand == -1 or geneStrand == 1
    return -geneStrand


def convert_int_strand_to_string(intStrand):
    assert intStrand == -1 or intStrand == 1
    return _CHAR[intStrand]


class Gene:
    """"+name" / "-name" -> (name, strand); hash = sha256-of-name * strand (:47-93)."""

    __slots__ = ("name", "strand")

    def __init__(self, gene):
        assert gene.replace(" ", "") != "", "Gene information is missing"
        strand_char = gene[0]
        name = gene[1:].replace(" ", "_")
        assert strand_char == "-" or strand_char == "+", "Strand information missing for: " + gene
        assert name != "", "Gene name information missing for: " + gene
        self.name = name
        self.strand = convert_string_strand_to_int(strand_char)

    def get_name(self):
        return self.name

    def get_strand(self):
        return self.strand

    def reverse_gene(self):
        return Gene(_CHAR[reverse_strand(self.strand)] + self.name)

    def __str__(self):
        return _CHAR[self.strand] + self.name

    def __eq__(self, otherGene):
        return self.strand == otherGene.get_strand() and self.name == otherGene.get_name()

    def __hash__(self):
        return hashlib_hash(self.name) * self.strand
