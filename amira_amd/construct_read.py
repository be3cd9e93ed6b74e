"""Read value object — drop-in for amira/construct_read.py (reference v0.11.0).

get_geneMers is the sliding-window extraction that k_node_upsert performs from the CSR
token layout (one window per token index); this class serves single reads on the host.
"""
from .construct_gene import Gene
from .construct_gene_mer import GeneMer


def convert_genes(annotatedGenes):
    return [Gene(g) for g in annotatedGenes]


class Read:
    def __init__(self, readId, annotatedGenes, annotatedGenePositions=None):
        self.readId = readId
        self.numberOfGenes = len(annotatedGenes)
        self.listOfGenes = convert_genes(annotatedGenes)
        self._annotatedGenes = annotatedGenes
        self._annotatedGenePositions = annotatedGenePositions

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

    def get_geneMers(self, kmerSize):
        """all windows of kmerSize genes and their (start, end) spans (:37-59)."""
        n_windows = self.numberOfGenes - kmerSize + 1
        positions = self._annotatedGenePositions
        geneMers = [GeneMer(self.listOfGenes[i:i + kmerSize]) for i in range(max(n_windows, 0))]
        if positions:
            spans = [(positions[i][0], positions[i + kmerSize - 1][1]) for i in range(max(n_windows, 0))]
        else:
            spans = [None] * max(n_windows, 0)
        return geneMers, spans
