"""Node — drop-in for amira/construct_node.py (reference v0.11.0).

Inside a GeneMerGraph these objects are VIEWS materialised from the device's node table
(coverage counter, ordered de-duplicated read list, ordered forward/backward edge lists,
component id, first-seen GeneMer); standalone they behave like the reference's class.
"""
from .construct_read import GeneMer, Read  # noqa: F401  (same re-exports as the reference)


class Node:
    def __init__(self, geneMer):
        self.geneMer = geneMer
        self.canonicalGeneMer = geneMer.get_canonical_geneMer()
        self.reverseGeneMer = geneMer.get_rc_geneMer()
        self.geneMerHash = geneMer.__hash__()
        self.nodeCoverage = 0
        self.listOfReads = []
        self.forwardEdgeHashes = []
        self.backwardEdgeHashes = []
        self._color = None
        self._component_ID = None

    # ---- gene-mer
    def get_geneMer(self):
        return self.geneMer

    def get_canonical_geneMer(self):
        return self.canonicalGeneMer

    def get_reverse_geneMer(self):
        return self.reverseGeneMer

    # ---- coverage
    def get_node_coverage(self):
        return self.nodeCoverage

    def increment_node_coverage(self):
        self.nodeCoverage += 1
        return self.nodeCoverage

    def extend_node_coverage(self, value):
        self.nodeCoverage += value
        return self.nodeCoverage

    # ---- reads (ordered, de-duplicated: construct_node.py:64-67)
    def get_list_of_reads(self):
        return self.listOfReads

    def get_reads(self):
        for read in self.listOfReads:
            yield read

    def add_read(self, read):
        if read not in self.listOfReads:
            self.listOfReads.append(read)

    def remove_read(self, read):
        assert read in self.listOfReads, "This node does not contain the read: " + read.get_readId()
        del self.listOfReads[self.listOfReads.index(read)]

    # ---- component / colour
    def get_color(self):
        return self._color

    def get_component(self):
        return self._component_ID

    def set_component(self, new_component_ID):
        self._component_ID = int(new_component_ID)
        return self._component_ID

    # ---- edge hash lists (ordered, append-if-absent: :79-101)
    def get_forward_edge_hashes(self):
        return self.forwardEdgeHashes

    def get_backward_edge_hashes(self):
        return self.backwardEdgeHashes

    def add_forward_edge_hash(self, forwardEdgeHash):
        if forwardEdgeHash not in self.forwardEdgeHashes:
            self.forwardEdgeHashes.append(forwardEdgeHash)
        return self

    def add_backward_edge_hash(self, backwardEdgeHash):
        if backwardEdgeHash not in self.backwardEdgeHashes:
            self.backwardEdgeHashes.append(backwardEdgeHash)
        return self

    def remove_forward_edge_hash(self, edgeHash):
        assert edgeHash in self.forwardEdgeHashes, "This edge hash is not in the list of forward edge hashes"
        del self.forwardEdgeHashes[self.forwardEdgeHashes.index(edgeHash)]

    def remove_backward_edge_hash(self, edgeHash):
        assert edgeHash in self.backwardEdgeHashes, "This edge hash is not in the list of backward edge hashes"
        del self.backwardEdgeHashes[self.backwardEdgeHashes.index(edgeHash)]

    # ---- ids
    def assign_node_Id(self, nodeId):
        self._nodeId = nodeId
        return self._nodeId

    def get_node_Id(self):
        return self._nodeId

    def __eq__(self, otherNode):
        return (self.__hash__() == otherNode.__hash__()
                and self.get_node_coverage() == otherNode.get_node_coverage())

    def __hash__(self):
        return self.geneMerHash

    def color_node(self, listOfAMRGenes):
        """0: no AMR gene, 1: AMR gene off a junction, 2: AMR gene at a junction (:134-154)."""
        names = [g.get_name() for g in self.canonicalGeneMer]
        if not any(g in listOfAMRGenes for g in names):
            self._color = 0
        elif len(self.forwardEdgeHashes) + len(self.backwardEdgeHashes) > 2:
            self._color = 2
        else:
            self._color = 1
