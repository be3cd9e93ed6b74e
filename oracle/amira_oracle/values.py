"""Oracle value objects.  TEST INFRASTRUCTURE ONLY (oracle/README.md).

Restates, in our own words, the value types of the reference:
  hashlib_hash / Gene      amira/construct_gene.py:5-10, 47-93
  GeneMer                  amira/construct_gene_mer.py:4-97
  Read                     amira/construct_read.py:11-59
  Node                     amira/construct_node.py:4-154
  Edge                     amira/construct_edge.py:31-124
Accessor names follow the reference so one dump routine serves reference,
oracle and product.
"""
import hashlib
import pickle


def hashlib_hash(value):
    # construct_gene.py:5-10 — sha256 over the pickle of the value, as a Python int
    return int(hashlib.sha256(pickle.dumps(value)).hexdigest(), 16)


# The reference re-hashes on every __hash__ call (no memoisation).  Tests keep the
# memo on for speed; bench.py's cpu_baseline leg switches it off so the timed work
# is the reference's (28.6 sha256+pickle per gene-mer).
CACHE_HASHES = True

_STR2INT = {"+": 1, "-": -1}
_INT2STR = {1: "+", -1: "-"}


class Gene:
    # construct_gene.py:47-93
    __slots__ = ("name", "strand", "_h")

    def __init__(self, gene):
        assert gene.replace(" ", "") != "", "Gene information is missing"
        sign, name = gene[0], gene[1:].replace(" ", "_")
        assert sign in _STR2INT, "Strand information missing for: " + gene
        assert name != "", "Gene name information missing for: " + gene
        self.name, self.strand, self._h = name, _STR2INT[sign], None

    def get_name(self):
        return self.name

    def get_strand(self):
        return self.strand

    def reverse_gene(self):
        return Gene(_INT2STR[-self.strand] + self.name)

    def as_string(self):
        return _INT2STR[self.strand] + self.name

    def __eq__(self, other):
        return self.strand == other.get_strand() and self.name == other.get_name()

    def __hash__(self):
        if not CACHE_HASHES:
            return hashlib_hash(self.name) * self.strand
        if self._h is None:
            self._h = hashlib_hash(self.name)
        return self._h * self.strand


class GeneMer:
    # construct_gene_mer.py:42-97 (define_geneMer + class GeneMer)
    def __init__(self, genes):
        assert isinstance(genes, list), "Gene-mer is not a list of Gene objects"
        assert genes != [], "Gene-mer is empty"
        assert all(isinstance(g, Gene) for g in genes)
        rc = [g.reverse_gene() for g in reversed(genes)]  # :4-12
        fwd_h = [g.__hash__() for g in genes]  # :15-28
        rc_h = [g.__hash__() for g in rc]
        assert fwd_h != rc_h, "Gene-mer and reverse complement gene-mer are identical"
        if fwd_h < rc_h:  # :31-39 — lexicographically smaller hash list is canonical
            self.canonicalGeneMer, self.rcGeneMer, self.geneMerDirection = genes, rc, 1
        else:
            self.canonicalGeneMer, self.rcGeneMer, self.geneMerDirection = rc, genes, -1
        self.geneMerSize = len(genes)
        self._h = None

    def get_canonical_geneMer(self):
        return self.canonicalGeneMer

    def get_rc_geneMer(self):
        return self.rcGeneMer

    def get_geneMerDirection(self):
        return self.geneMerDirection

    def get_geneMer_size(self):
        return self.geneMerSize

    def __eq__(self, other):
        return (
            self.canonicalGeneMer == other.get_canonical_geneMer()
            and self.rcGeneMer == other.get_rc_geneMer()
        )

    def __hash__(self):
        # :94-97 — sha256 of the pickled tuple of signed canonical gene hashes
        if not CACHE_HASHES:
            return hashlib_hash(tuple([g.__hash__() for g in self.canonicalGeneMer]))
        if self._h is None:
            self._h = hashlib_hash(tuple(g.__hash__() for g in self.canonicalGeneMer))
        return self._h


class Read:
    # construct_read.py:11-59
    def __init__(self, readId, annotatedGenes, annotatedGenePositions=None):
        self.readId = readId
        self._annotatedGenes = annotatedGenes
        self._annotatedGenePositions = annotatedGenePositions
        self.listOfGenes = [Gene(g) for g in annotatedGenes]
        self.numberOfGenes = len(annotatedGenes)

    def get_readId(self):
        return self.readId

    def get_genes(self):
        return self.listOfGenes

    def get_number_of_genes(self):
        return self.numberOfGenes

    def get_annotatedGenes(self):
        return self._annotatedGenes

    def get_annotatedGenePositions(self):
        return self._annotatedGenePositions

    def get_geneMers(self, k):
        # :37-59 — every window i..i+k; position = (start of first gene, end of last gene)
        mers, spans = [], []
        pos = self._annotatedGenePositions
        for i in range(self.numberOfGenes - k + 1):
            if pos:
                spans.append((pos[i][0], pos[i + k - 1][1]))
            else:
                spans.append(None)
            mers.append(GeneMer(self.listOfGenes[i : i + k]))
        return mers, spans


class Node:
    # construct_node.py:4-154
    def __init__(self, geneMer):
        self.geneMer = geneMer
        self.canonicalGeneMer = geneMer.get_canonical_geneMer()
        self.reverseGeneMer = geneMer.get_rc_geneMer()
        self.geneMerHash = geneMer.__hash__()
        self.nodeCoverage = 0
        self.listOfReads = []
        self.forwardEdgeHashes = []
        self.backwardEdgeHashes = []
        self._component_ID = None
        self._color = None

    def get_geneMer(self):
        return self.geneMer

    def get_canonical_geneMer(self):
        return self.canonicalGeneMer

    def get_reverse_geneMer(self):
        return self.reverseGeneMer

    def get_node_coverage(self):
        return self.nodeCoverage

    def increment_node_coverage(self):
        self.nodeCoverage += 1
        return self.nodeCoverage

    def extend_node_coverage(self, value):
        self.nodeCoverage += value
        return self.nodeCoverage

    def get_list_of_reads(self):
        return self.listOfReads

    def get_reads(self):
        yield from self.listOfReads

    def get_component(self):
        return self._component_ID

    def set_component(self, c):
        self._component_ID = int(c)
        return self._component_ID

    def get_color(self):
        return self._color

    def add_read(self, read):
        if read not in self.listOfReads:  # :64-67 ordered, de-duplicated
            self.listOfReads.append(read)

    def remove_read(self, read):
        assert read in self.listOfReads
        self.listOfReads.remove(read)

    def get_forward_edge_hashes(self):
        return self.forwardEdgeHashes

    def get_backward_edge_hashes(self):
        return self.backwardEdgeHashes

    def add_forward_edge_hash(self, h):
        if h not in self.forwardEdgeHashes:
            self.forwardEdgeHashes.append(h)
        return self

    def add_backward_edge_hash(self, h):
        if h not in self.backwardEdgeHashes:
            self.backwardEdgeHashes.append(h)
        return self

    def remove_forward_edge_hash(self, h):
        assert h in self.forwardEdgeHashes, "This edge hash is not in the list of forward edge hashes"
        self.forwardEdgeHashes.remove(h)

    def remove_backward_edge_hash(self, h):
        assert h in self.backwardEdgeHashes, "This edge hash is not in the list of backward edge hashes"
        self.backwardEdgeHashes.remove(h)

    def assign_node_Id(self, i):
        self._nodeId = i
        return i

    def get_node_Id(self):
        return self._nodeId

    def __eq__(self, other):
        return (
            self.__hash__() == other.__hash__()
            and self.get_node_coverage() == other.get_node_coverage()
        )

    def __hash__(self):
        return self.geneMerHash


class Edge:
    # construct_edge.py:31-124
    def __init__(self, sourceNode, targetNode, sourceNodeDirection, targetNodeDirection):
        self.sourceNode = sourceNode
        self.targetNode = targetNode
        self.sourceNodeDirection = sourceNodeDirection
        self.targetNodeDirection = targetNodeDirection
        self.edgeCoverage = 0

    def get_sourceNode(self):
        return self.sourceNode

    def get_targetNode(self):
        return self.targetNode

    def set_sourceNode(self, n):
        self.sourceNode = n
        return n

    def set_targetNode(self, n):
        self.targetNode = n
        return n

    def get_sourceNodeDirection(self):
        return self.sourceNodeDirection

    def get_targetNodeDirection(self):
        return self.targetNodeDirection

    def set_sourceNodeDirection(self, d):
        self.sourceNodeDirection = d
        return d

    def set_targetNodeDirection(self, d):
        self.targetNodeDirection = d
        return d

    def get_edge_coverage(self):
        return self.edgeCoverage

    def increment_edge_coverage(self):
        self.edgeCoverage += 1
        return self.edgeCoverage

    def extend_edge_coverage(self, v):
        self.edgeCoverage += v
        return self.edgeCoverage

    def reduce_edge_coverage(self):
        self.edgeCoverage -= 1
        return self.edgeCoverage

    def __eq__(self, other):
        mine = sorted([self.sourceNode.__hash__(), self.targetNode.__hash__()])
        theirs = sorted([other.get_sourceNode().__hash__(), other.get_targetNode().__hash__()])
        return mine == theirs

    def __hash__(self):
        # :104-124 — min over the two sign variants of sha256(pickle((hS*dS, hT*dT)))
        a = self.sourceNode.__hash__() * self.sourceNodeDirection
        b = self.targetNode.__hash__() * self.targetNodeDirection
        return min(hashlib_hash((a, b)), hashlib_hash((-a, -b)))
