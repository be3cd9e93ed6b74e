"""Edge — drop-in for amira/construct_edge.py (reference v0.11.0).

The edge identity class used on the device, (source, target, sourceDir * targetDir), is
exactly the equivalence class of Edge.__hash__ below (min over the two sign variants).
"""
from .construct_gene import hashlib_hash
from .construct_node import Node  # noqa: F401


def extract_node_hashes(firstNode, secondNode):
    return firstNode.__hash__(), secondNode.__hash__()


def sort_node_hashes(firstNodeHash, secondNodeHash):
    lo, hi = sorted([firstNodeHash, secondNodeHash])
    return lo, hi


def define_source_and_target(firstNode, secondNode):
    return sort_node_hashes(*extract_node_hashes(firstNode, secondNode))


class Edge:
    def __init__(self, sourceNode, targetNode, sourceNodeDirection, targetNodeDirection):
        self.sourceNode = sourceNode
        self.targetNode = targetNode
        self.edgeCoverage = 0
        self.sourceNodeDirection = sourceNodeDirection
        self.targetNodeDirection = targetNodeDirection

    def get_sourceNode(self):
        return self.sourceNode

    def get_targetNode(self):
        return self.targetNode

    def set_sourceNode(self, new_sourceNode):
        self.sourceNode = new_sourceNode
        return self.sourceNode

    def set_targetNode(self, new_targetNode):
        self.targetNode = new_targetNode
        return self.targetNode

    def set_sourceNodeDirection(self, sourceDirection):
        self.sourceNodeDirection = sourceDirection
        return self.sourceNodeDirection

    def get_sourceNodeDirection(self):
        return self.sourceNodeDirection

    def set_targetNodeDirection(self, targetDirection):
        self.targetNodeDirection = targetDirection
        return self.targetNodeDirection

    def get_targetNodeDirection(self):
        return self.targetNodeDirection

    def get_edge_coverage(self):
        return self.edgeCoverage

    def increment_edge_coverage(self):
        self.edgeCoverage += 1
        return self.edgeCoverage

    def extend_edge_coverage(self, value):
        self.edgeCoverage += value
        return self.edgeCoverage

    def reduce_edge_coverage(self):
        self.edgeCoverage -= 1
        return self.edgeCoverage

    def __eq__(self, otherEdge):
        mine = sorted([self.sourceNode.__hash__(), self.targetNode.__hash__()])
        other = sorted([otherEdge.get_sourceNode().__hash__(), otherEdge.get_targetNode().__hash__()])
        return tuple(mine) == tuple(other)

    def __hash__(self):
        """min(sha256((hS*dS, hT*dT)), sha256((-hS*dS, -hT*dT))) — construct_edge.py:104-124."""
        s = self.sourceNode.__hash__() * self.sourceNodeDirection
        t = self.targetNode.__hash__() * self.targetNodeDirection
        return min(hashlib_hash((s, t)), hashlib_hash((-s, -t)))
