"""Oracle GeneMerGraph.  TEST INFRASTRUCTURE ONLY (oracle/README.md).

A dict/object restatement of amira/construct_graph.py (reference v0.11.0) for the
hot path: build (:31-102), accessors (:104-400), removal + coverage filter
(:402-540), unitig genes (:617-677), tip clipping (:679-861), components
(:911-958), per-read correction (:1123-1480, :1669-1691, :2292-2342) and
read-path clustering (:2360-2455, :2629-2959).  Quirks listed in SURVEY.md
Appendix A are kept on purpose; each method cites the lines it follows.
"""
import statistics
import sys
from collections import deque
from itertools import product

from . import paths as pf
from .bubbles import BubbleMixin
from .values import Edge, GeneMer, Node, Read, _INT2STR

sys.setrecursionlimit(50000)  # construct_graph.py:27


class GeneMerGraph(BubbleMixin):
    # ------------------------------------------------------------------ build
    def __init__(self, readDict, kmerSize, gene_positions=None):
        # construct_graph.py:31-102
        self._reads = readDict
        self._kmerSize = kmerSize
        self._minNodeCoverage = 1
        self._minEdgeCoverage = 1
        self._genePositions = gene_positions
        self._nodes = {}
        self._edges = {}
        self._readNodes = {}
        self._readNodeDirections = {}
        self._readNodePositions = {}
        self._shortReads = {}
        self._readsToCorrect = set()
        for rid in readDict:
            pos = gene_positions[rid] if gene_positions else None
            read = Read(rid, readDict[rid], pos)
            mers, spans = read.get_geneMers(kmerSize)
            if not mers:
                self._shortReads[rid] = read.get_annotatedGenes()  # :53-55
                continue
            last = len(mers) - 1
            for i, gm in enumerate(mers):
                node = self.add_node(gm, [rid])
                self.add_node_to_read(node, rid, gm.get_geneMerDirection(), spans[i])
                node.increment_node_coverage()
                if i < last:
                    self.add_node(mers[i + 1], [rid])
                    e1, e2 = self.add_edge(gm, mers[i + 1])
                    e1.increment_edge_coverage()
                    e2.increment_edge_coverage()
        self.assign_component_ids()

    # -------------------------------------------------------------- accessors
    def get_reads(self):
        return self._reads

    def get_short_read_annotations(self):
        return self._shortReads

    def get_gene_positions(self):
        return self._genePositions

    def get_short_read_gene_positions(self):
        return {r: self._genePositions[r] for r in self._shortReads}

    def get_readNodes(self):
        return self._readNodes

    def get_readNodeDirections(self):
        return self._readNodeDirections

    def get_readNodePositions(self):
        return self._readNodePositions

    def get_kmerSize(self):
        return self._kmerSize

    def get_minEdgeCoverage(self):
        return self._minEdgeCoverage

    def get_minNodeCoverage(self):
        return self._minNodeCoverage

    def set_minNodeCoverage(self, v):
        self._minNodeCoverage = v
        return v

    def set_minEdgeCoverage(self, v):
        self._minEdgeCoverage = v
        return v

    def get_nodes(self):
        return self._nodes

    def get_edges(self):
        return self._edges

    def get_reads_to_correct(self):
        return self._readsToCorrect

    def all_nodes(self):
        for h in self._nodes:
            yield self._nodes[h]

    def get_reads_for_nodes(self, hashes):
        out = set()
        for h in hashes:
            out.update(self._nodes[h].get_list_of_reads())
        return out

    def get_total_number_of_nodes(self):
        return len(self._nodes)

    def get_total_number_of_edges(self):
        return len(self._edges)

    def get_total_number_of_reads(self):
        return len(self._reads)

    # ------------------------------------------------------------- node/edge insertion
    def add_node_to_read(self, node, readId, direction, position=None):
        # :165-178
        if readId not in self._readNodes:
            self._readNodes[readId] = []
            self._readNodeDirections[readId] = []
            self._readNodePositions[readId] = []
        self._readNodes[readId].append(node.__hash__())
        self._readNodeDirections[readId].append(direction)
        self._readNodePositions[readId].append(position)
        return self._readNodes[readId]

    def get_nodes_containing_read(self, readId):
        return [self._nodes[h] for h in self._readNodes[readId] if h in self._nodes]

    def add_node_to_nodes(self, node, h):
        self._nodes[h] = node

    def get_node_by_hash(self, h):
        return self._nodes[h]

    def add_node(self, geneMer, reads):
        # :196-212 — insert-or-find; the FIRST GeneMer object seen stays on the node
        h = geneMer.__hash__()
        node = self._nodes.get(h)
        if node is None:
            node = Node(geneMer)
            self._nodes[h] = node
        for r in reads:
            node.add_read(r)
        return node

    def get_node(self, geneMer):
        h = geneMer.__hash__()
        assert h in self._nodes, "This gene-mer is not in the graph"
        return self._nodes[h]

    def get_nodes_containing(self, gene):
        # :223-244
        assert not (gene[0] == "+" or gene[0] == "-"), (
            "Strand information cannot be present for any specified genes"
        )
        assert isinstance(gene, str)
        return [
            n
            for n in self.all_nodes()
            if gene in [g.get_name() for g in n.get_canonical_geneMer()]
        ]

    def create_edges(self, src, tgt, sdir, tdir):
        # :246-262 — E1 = (A,B,dA,dB), E2 = (B,A,-dB,-dA)
        return Edge(src, tgt, sdir, tdir), Edge(tgt, src, -tdir, -sdir)

    def get_edge_by_hash(self, h):
        return self._edges[h]

    def add_edge_to_edges(self, edge):
        # :268-277 — first-seen Edge object is the one stored
        h = edge.__hash__()
        if h not in self._edges:
            self._edges[h] = edge
        return self._edges[h]

    def add_edges_to_graph(self, e1, e2):
        return self.add_edge_to_edges(e1), self.add_edge_to_edges(e2)

    def add_edge_to_node(self, node, edge):
        # :287-298 — list chosen by the STORED edge's source direction
        if edge.get_sourceNodeDirection() == 1:
            node.add_forward_edge_hash(edge.__hash__())
        if edge.get_sourceNodeDirection() == -1:
            node.add_backward_edge_hash(edge.__hash__())
        return node

    def add_edge(self, srcMer, tgtMer):
        # :300-324
        src = self.add_node(srcMer, [])
        tgt = self.add_node(tgtMer, [])
        e1, e2 = self.create_edges(
            src, tgt, srcMer.get_geneMerDirection(), tgtMer.get_geneMerDirection()
        )
        e1, e2 = self.add_edges_to_graph(e1, e2)
        self.add_edge_to_node(src, e1)
        self.add_edge_to_node(tgt, e2)
        return e1, e2

    # ---------------------------------------------------------------- topology queries
    def get_degree(self, node):
        return len(node.get_forward_edge_hashes()) + len(node.get_backward_edge_hashes())

    def get_forward_edges(self, node):
        return [self._edges[h] for h in node.get_forward_edge_hashes()]

    def get_backward_edges(self, node):
        return [self._edges[h] for h in node.get_backward_edge_hashes()]

    def get_forward_neighbors(self, node):
        return [e.get_targetNode() for e in self.get_forward_edges(node)]

    def get_backward_neighbors(self, node):
        return [e.get_targetNode() for e in self.get_backward_edges(node)]

    def get_all_neighbors(self, node):
        return self.get_forward_neighbors(node) + self.get_backward_neighbors(node)

    def get_all_neighbor_hashes(self, node):
        return {n.__hash__() for n in self.get_all_neighbors(node)}

    def check_if_nodes_are_adjacent(self, a, b):
        return b.__hash__() in self.get_all_neighbor_hashes(
            a
        ) and a.__hash__() in self.get_all_neighbor_hashes(b)

    def get_edge_hashes_between_nodes(self, a, b):
        # :364-386 — scalar pair normally, pair of LISTS in the multi-edge case (quirk kept)
        assert self.check_if_nodes_are_adjacent(a, b)
        ab = [
            e.__hash__()
            for e in self.get_forward_edges(a) + self.get_backward_edges(a)
            if e.get_targetNode() == b
        ]
        ba = [
            e.__hash__()
            for e in self.get_forward_edges(b) + self.get_backward_edges(b)
            if e.get_targetNode() == a
        ]
        if len(ab) > 1 or len(ba) > 1:
            return (ab, ba)
        return (ab[0], ba[0])

    def get_edges_between_nodes(self, a, b):
        x, y = self.get_edge_hashes_between_nodes(a, b)
        if isinstance(x, list) or isinstance(y, list):
            return [self._edges[h] for h in x], [self._edges[h] for h in y]
        return self._edges[x], self._edges[y]

    # --------------------------------------------------------------- removal + filter
    def remove_edge_from_edges(self, h):
        del self._edges[h]

    def remove_edge(self, h):
        # :409-428 — silently ignores unknown hashes
        if h not in self._edges:
            return
        edge = self._edges[h]
        src = edge.get_sourceNode()
        if edge.get_sourceNodeDirection() == 1:
            src.remove_forward_edge_hash(h)
        if edge.get_sourceNodeDirection() == -1:
            src.remove_backward_edge_hash(h)
        del self._edges[h]

    def remove_node_from_reads(self, node):
        # :442-461 — every occurrence becomes None in the three per-read lists
        h = node.__hash__()
        for rid in node.get_reads():
            keep = [x != h for x in self._readNodes[rid]]
            self._readNodes[rid] = [
                x if k else None for x, k in zip(self._readNodes[rid], keep)
            ]
            self._readNodeDirections[rid] = [
                x if k else None for x, k in zip(self._readNodeDirections[rid], keep)
            ]
            self._readNodePositions[rid] = [
                x if k else None for x, k in zip(self._readNodePositions[rid], keep)
            ]
            self._readsToCorrect.add(rid)

    def remove_node(self, node):
        # :463-484
        h = node.__hash__()
        assert h in self._nodes, "This node is not in the graph"
        assert node == self._nodes[h]
        self.remove_node_from_reads(node)
        for eh in set(node.get_forward_edge_hashes() + node.get_backward_edge_hashes()):
            tgt = self._edges[eh].get_targetNode()
            for e in self.get_edge_hashes_between_nodes(node, tgt):
                self.remove_edge(e)
        del self._nodes[h]

    def list_nodes_to_remove(self, minNodeCoverage):
        # :496-503
        return {n for n in self._nodes.values() if not n.get_node_coverage() > minNodeCoverage - 1}

    def list_edges_to_remove(self, minEdgeCoverage, nodesToRemove):
        # :505-521
        out = set()
        for h, e in self._edges.items():
            if not e.get_edge_coverage() > minEdgeCoverage - 1:
                out.add(h)
            if e.get_sourceNode() in nodesToRemove or e.get_targetNode() in nodesToRemove:
                out.add(h)
        return out

    def filter_graph(self, minNodeCoverage, minEdgeCoverage):
        # :523-540 — edges first, then nodes
        self._minNodeCoverage = minNodeCoverage
        self._minEdgeCoverage = minEdgeCoverage
        doomed_nodes = self.list_nodes_to_remove(minNodeCoverage)
        for h in self.list_edges_to_remove(minEdgeCoverage, doomed_nodes):
            self.remove_edge(h)
        for n in doomed_nodes:
            self.remove_node(n)
        return self

    # ------------------------------------------------------------------ gene strings
    def get_gene_mer_genes(self, node):
        return [_INT2STR[g.get_strand()] + g.get_name() for g in node.get_canonical_geneMer()]

    def get_reverse_gene_mer_genes(self, node):
        return [_INT2STR[g.get_strand()] + g.get_name() for g in node.get_reverse_geneMer()]

    def get_gene_mer_label(self, node):
        return "~~~".join(self.get_gene_mer_genes(node))

    def get_nodes_with_degree(self, degree):
        assert isinstance(degree, int)
        return [n for n in self.all_nodes() if self.get_degree(n) == degree]

    def reverse_list_of_genes(self, genes):
        return [("-" if g[0] == "+" else "+") + g[1:] for g in reversed(genes)]

    def get_genes_in_unitig(self, hashes):
        # :617-677 — append mode first, whole-procedure restart in prepend mode on mismatch
        if len(hashes) == 1:
            return self.get_gene_mer_genes(self._nodes[hashes[0]])
        k1 = self._kmerSize - 1

        def seed(i):
            a, b = self._nodes[hashes[i]], self._nodes[hashes[i + 1]]
            edge = self._edges[self.get_edge_hashes_between_nodes(a, b)[0]]
            if edge.get_sourceNodeDirection() == 1:
                return self.get_gene_mer_genes(a)
            return self.get_reverse_gene_mer_genes(a)

        out, failed = [], False
        for i in range(len(hashes) - 1):
            if i == 0:
                out += seed(0)
            else:
                seed(i)  # the reference looks the edge up every step (can assert)
            tgt = self._nodes[hashes[i + 1]]
            fw, bw = self.get_gene_mer_genes(tgt), self.get_reverse_gene_mer_genes(tgt)
            tail = out[-k1:] if k1 else out[0:]
            if fw[:-1] == tail:
                out.append(fw[-1])
            elif bw[:-1] == tail:
                out.append(bw[-1])
            else:
                failed = True
                break
        if not failed:
            return out
        out = []
        for i in range(len(hashes) - 1):
            if i == 0:
                out += seed(0)
            else:
                seed(i)
            tgt = self._nodes[hashes[i + 1]]
            fw, bw = self.get_gene_mer_genes(tgt), self.get_reverse_gene_mer_genes(tgt)
            head = out[:k1]
            if fw[1:] == head:
                out.insert(0, fw[0])
            elif bw[1:] == head:
                out.insert(0, bw[0])
            else:
                raise ValueError("Gene sequences do not match in alternative path.")
        return out

    # ------------------------------------------------------------------ GML output (:542-586, :873-909)
    def write_node_entry(self, node_id, node_string, node_coverage, reads, component_ID, nodeColor):
        rows = ["\tnode\t[", f"\t\tid\t{node_id}", f'\t\tlabel\t"{node_string}"',
                f"\t\tcoverage\t{node_coverage}"]
        if component_ID:
            rows.append(f"\t\tcomponent\t{component_ID}")
        rows.append('\t\treads\t"' + ",".join(reads) + '"')
        if nodeColor:
            rows.append(f'\t\tcolor\t"{nodeColor}"')
        rows.append("\t]")
        return "\n".join(rows)

    def write_edge_entry(self, source_node, target_node, source_edge_direction, target_edge_direction,
                         edge_coverage):
        return "\n".join(["\tedge\t[", f"\t\tsource\t{source_node}", f"\t\ttarget\t{target_node}",
                          f"\t\tsource_direction\t{source_edge_direction}",
                          f"\t\ttarget_direction\t{target_edge_direction}",
                          f"\t\tweight\t{edge_coverage}", "\t]"])

    def assign_Id_to_nodes(self):
        for i, node in enumerate(self.all_nodes()):
            assert node.assign_node_Id(i) == i, "This node was assigned an incorrect ID"

    def write_gml_to_file(self, output_file, gml_content):
        import os
        folder = os.path.dirname(output_file)
        if folder != "" and not os.path.exists(folder):
            os.mkdir(folder)
        with open(output_file + ".gml", "w") as fh:
            fh.write("\n".join(gml_content))

    def generate_gml(self, output_file, geneMerSize, min_node_coverage, min_edge_coverage):
        """nodes in dict order, each followed by its forward then backward edges (:873-909)"""
        graph_data = ["graph\t[", "multigraph 1"]
        self.assign_Id_to_nodes()
        for node in self.all_nodes():
            graph_data.append(self.write_node_entry(node.get_node_Id(), self.get_gene_mer_label(node),
                                                    node.get_node_coverage(), [r for r in node.get_reads()],
                                                    node.get_component(), node.get_color()))
            for edge in self.get_forward_edges(node) + self.get_backward_edges(node):
                if edge.get_edge_coverage() == 0:
                    continue
                graph_data.append(self.write_edge_entry(node.get_node_Id(), edge.get_targetNode().get_node_Id(),
                                                        edge.get_sourceNodeDirection(),
                                                        edge.get_targetNodeDirection(),
                                                        edge.get_edge_coverage()))
        graph_data.append("]")
        self.write_gml_to_file(".".join([output_file, str(geneMerSize), str(min_node_coverage),
                                         str(min_edge_coverage)]), graph_data)
        return graph_data

    # ------------------------------------------------------------------ linear paths
    def _step(self, node, use_forward):
        # get_forward_node_from_node :722-741 / get_backward_node_from_node :781-802.
        # Forward needs EXACTLY one forward edge; backward takes the FIRST backward edge.
        if use_forward:
            hs = node.get_forward_edge_hashes()
            if len(hs) != 1:
                return False, None, None
        else:
            hs = node.get_backward_edge_hashes()
            if len(hs) == 0:
                return False, None, None
        edge = self._edges[hs[0]]
        tgt = edge.get_targetNode()
        ok = self.get_degree(tgt) in (1, 2) and tgt != node
        return ok, tgt, edge.get_targetNodeDirection()

    def get_forward_node_from_node(self, node):
        return self._step(node, True)

    def get_backward_node_from_node(self, node):
        return self._step(node, False)

    def get_forward_path_from_node(self, node, startDirection, wantBranchedNode=False):
        # :743-779
        path = [node.__hash__()]
        ext, nxt, d = self._step(node, startDirection == 1)
        while ext:
            if path[0] == nxt.__hash__():
                break
            path.append(nxt.__hash__())
            ext, nxt, d = self._step(nxt, d == 1)
        if wantBranchedNode and nxt:
            path.append(nxt.__hash__())
        return path

    def get_backward_path_from_node(self, node, startDirection, wantBranchedNode=False):
        # :804-847
        path = [node.__hash__()]
        ext, nxt, d = self._step(node, startDirection != -1)
        while ext:
            if path[-1] == nxt.__hash__():
                break
            path.insert(0, nxt.__hash__())
            ext, nxt, d = self._step(nxt, d != -1)
        if wantBranchedNode and nxt:
            path.insert(0, nxt.__hash__())
        return path

    def get_linear_path_for_node(self, node, wantBranchedNode=False):
        # :849-861 — orientation = direction of the node's FIRST occurrence
        d0 = node.get_geneMer().get_geneMerDirection()
        back = self.get_backward_path_from_node(node, -1 * d0, wantBranchedNode)
        assert back[-1] == node.__hash__()
        fwd = self.get_forward_path_from_node(node, d0, wantBranchedNode)
        assert fwd[0] == node.__hash__()
        return back[:-1] + [node.__hash__()] + fwd[1:]

    def get_all_node_coverages(self):
        return [n.get_node_coverage() for n in self.all_nodes()]

    def get_mean_node_coverage(self):
        return statistics.mean(self.get_all_node_coverages())

    def calculate_mean_node_coverage(self):
        return statistics.mean(self.get_all_node_coverages())  # :2565-2569

    def get_AMR_nodes(self, genes):
        out = {}
        for g in genes:
            for n in self.get_nodes_containing(g):
                out[n.__hash__()] = n
        return out

    def remove_short_linear_paths(self, min_length, sample_genesOfInterest={}):
        # :679-720
        by_component = {}
        for node in self.all_nodes():
            if self.get_degree(node) != 1:
                continue
            path = self.get_linear_path_for_node(node)
            if not (0 < len(path) < min_length):
                continue
            if all(
                self._nodes[h].get_node_coverage() > self.get_mean_node_coverage() * 1.5
                for h in path
            ):
                continue
            by_component.setdefault(node.get_component(), []).append(path)
        amr = self.get_AMR_nodes(sample_genesOfInterest)
        removed = set()
        for comp, plist in by_component.items():
            members = (
                {n.__hash__() for n in self.get_nodes_in_component(comp)}
                if comp is not None
                else []
            )
            for path in plist:
                if comp is not None and len(members.intersection(path)) == len(members):
                    continue  # the tip is the whole component
                for h in path:
                    if h in amr or h in removed:
                        continue
                    self.remove_node(self._nodes[h])
                    removed.add(h)
        return list(removed)

    # ------------------------------------------------------------------ components
    def dfs_component(self, start, cid, visited=None):
        # :911-918
        if visited is None:
            visited = set()
        visited.add(start.__hash__())
        start.set_component(cid)
        for nb in self.get_all_neighbors(start):
            if nb.__hash__() not in visited:
                self.dfs_component(nb, cid, visited)

    def assign_component_ids(self):
        # :920-927
        visited, cid = set(), 1
        for h in self._nodes:
            if h not in visited:
                self.dfs_component(self._nodes[h], cid, visited)
                cid += 1

    def get_nodes_in_component(self, component):
        return [n for n in self._nodes.values() if n.get_component() == int(component)]

    def components(self):
        return sorted({n.get_component() for n in self._nodes.values()})

    def get_number_of_component(self):
        return len(self.components())

    def remove_low_coverage_components(self, min_component_coverage):
        # :950-958
        for cid in self.components():
            members = self.get_nodes_in_component(cid)
            if all(n.get_node_coverage() < min_component_coverage for n in members):
                for n in members:
                    self.remove_node(n)

    # ------------------------------------------------------------------ read correction
    def correct_reads(self, fastq_data):
        # :1123-1134
        genes_out, pos_out = {}, {}
        for rid in self._readNodes:
            genes = self.correct_single_read(rid, self._readNodes, fastq_data)
            if len(genes) > 0:
                genes_out[rid] = genes
                if self._genePositions:
                    pos_out[rid] = self._genePositions[rid]
        return genes_out, pos_out

    def correct_single_read(self, rid, readNodes, fastq_data):
        # :1136-1151
        if rid not in self._readsToCorrect:
            return self._reads[rid]
        if all(n is None for n in readNodes[rid]):
            return []
        start, end = self.find_read_boundaries(readNodes[rid])
        new = self.process_read_correction(rid, readNodes, start, end, fastq_data)
        if self._genePositions:
            assert len(new) == len(self._genePositions[rid])
        return new

    def find_read_boundaries(self, nodes):
        # :1153-1164 — truthiness, not "is None"
        start, end = 0, len(nodes) - 1
        for i, n in enumerate(nodes):
            if n:
                start = i
                break
        for i, n in enumerate(reversed(nodes)):
            if n:
                end = len(nodes) - 1 - i
                break
        return start, end

    def identify_path_terminals(self, nodes, start, end):
        # :1375-1386 — path_start persists across a run of None
        out = []
        for i in range(len(nodes)):
            if start <= i <= end and not nodes[i]:
                if nodes[i - 1]:
                    path_start = i - 1
                if nodes[i + 1]:
                    out.append((path_start, i + 1))
        return out

    def new_find_paths_between_nodes(
        self, start_hash, end_hash, distance, direction, path=None, seen=None
    ):
        # :2292-2342 — simple paths, <= distance nodes incl. endpoints, list order
        path = [] if path is None else path
        seen = set() if seen is None else seen
        path.append((start_hash, direction))
        seen.add(start_hash)
        if (end_hash and start_hash == end_hash and len(path) <= distance) or (
            end_hash is None and len(path) - 1 == distance
        ):
            return [list(path)]
        if len(path) - 1 > distance:
            return []
        node = self._nodes[start_hash]
        if direction == 1:
            hops = node.get_forward_edge_hashes()
        elif direction == -1:
            hops = node.get_backward_edge_hashes()
        else:
            hops = []
        found = []
        for eh in hops:
            e = self._edges[eh]
            nh = e.get_targetNode().__hash__()
            if nh in seen:
                continue
            found.extend(
                self.new_find_paths_between_nodes(
                    nh, end_hash, distance, e.get_targetNodeDirection(), list(path), seen | {nh}
                )
            )
        return found

    def generate_replacement_dict(self, tagged, pair):
        # :1388-1396
        a, b = pair
        return {
            pair: self.new_find_paths_between_nodes(
                tagged[a][0], tagged[b][0], self._kmerSize * 2, tagged[a][1]
            )
        }

    def insert_elements(self, base, inserts):
        # :1166-1203
        if len(inserts) == 0:
            return [base]
        keyed = [[(key, p) for p in plist] for key, plist in inserts.items()]
        out = []
        for combo in product(*keyed):
            cur, shift = base[:], 0
            for (s, e), p in combo:
                cur[s + shift : e + shift + 1] = p
                shift += len(p) - (e - s + 1)
            out.append(cur)
        return out

    def get_possible_paths(self, tagged, inserts, start, end):
        # :1205-1263 — upstream/downstream extension is disabled in the reference
        out = []
        for cand in self.insert_elements(tagged, inserts):
            out.append(([n for n, _ in cand if n], [d for n, d in cand if n]))
        return out

    def get_coverage_of_path(self, path):
        return statistics.mean([self._nodes[h].get_node_coverage() for h in path])

    def get_annotation_for_read(self, nodes, dirs, rid):
        # :1331-1373
        assert len(nodes) == len(dirs)
        if not dirs:
            dirs = self._readNodeDirections[rid]
        if len(nodes) == 1:
            if dirs[0] == 1:
                return self.get_gene_mer_genes(self._nodes[nodes[0]])
            if dirs[0] == -1:
                return self.get_reverse_gene_mer_genes(self._nodes[nodes[0]])
            raise ValueError(f"Gene-mer direction for a node with 1 read cannot be {dirs[0]}")
        out = []
        for i, h in enumerate(nodes):
            node, d = self._nodes[h], dirs[i]
            if i == 0:
                seed = (
                    self.get_gene_mer_genes(node) if d == 1 else self.get_reverse_gene_mer_genes(node)
                )
                out += seed[:-1]
            if d:
                g = self.get_gene_mer_genes(node) if d == 1 else self.get_reverse_gene_mer_genes(node)
                out.append(g[-1])
        assert None not in out
        return out

    def process_read_correction(self, rid, readNodes, start, end, fastq_data):
        # :1269-1329
        k = self._kmerSize
        tagged = list(zip(readNodes[rid], self._readNodeDirections[rid]))
        terminals = self.identify_path_terminals(readNodes[rid], start, end)
        if not terminals:
            if self._genePositions:
                self._genePositions[rid] = self._genePositions[rid][start : end + k]
            return self.get_annotation_for_read(
                [n for n, _ in tagged[start : end + 1]], [d for _, d in tagged[start : end + 1]], rid
            )
        inserts = {}
        for pair in terminals:
            inserts.update(self.generate_replacement_dict(tagged, pair))
        options = self.get_possible_paths(tagged, inserts, start, end)
        if options == []:
            return self._reads[rid]
        best_shared, best_cov = 0, 0
        for nodes, dirs in options:
            cov = self.get_coverage_of_path(nodes)
            genes = self.get_annotation_for_read(nodes, dirs, rid)
            shared = len(set(genes).intersection(self._reads[rid]))
            if shared > best_shared or (shared == best_shared and cov > best_cov):
                closest, best_shared, best_cov = genes, shared, cov
        new_pos, cur = [], 0
        for a, b in self.needleman_wunsch(closest, self._reads[rid]):
            if a != "*":
                if b != a:
                    new_pos.append((None, None))
                else:
                    new_pos.append(self._genePositions[rid][cur])
                    cur += 1
            else:
                cur += 1
        self._genePositions[rid] = self.replace_invalid_gene_positions(new_pos, fastq_data, rid)
        return closest

    def score(self, a, b):
        return int(a == b)

    def needleman_wunsch(self, x, y):
        # :1433-1480 — gap -1, match 1, mismatch 0; first-column/row init is -index;
        # ties resolved by max over (score, pointer) tuples: UP > LEFT > DIAG
        N, M = len(x), len(y)
        DIAG, LEFT, UP = (-1, -1), (-1, 0), (0, -1)
        F, P = {(-1, -1): 0}, {}
        for i in range(N):
            F[i, -1] = -i
        for j in range(M):
            F[-1, j] = -j
        for i in range(N):
            for j in range(M):
                F[i, j], P[i, j] = max(
                    (F[i - 1, j - 1] + int(x[i] == y[j]), DIAG),
                    (F[i - 1, j] - 1, LEFT),
                    (F[i, j - 1] - 1, UP),
                )
        out = deque()
        i, j = N - 1, M - 1
        while i >= 0 and j >= 0:
            step = P[i, j]
            if step == DIAG:
                out.appendleft((x[i], y[j]))
            elif step == LEFT:
                out.appendleft((x[i], "*"))
            else:
                out.appendleft(("*", y[j]))
            i, j = i + step[0], j + step[1]
        while i >= 0:
            out.appendleft((x[i], "*"))
            i -= 1
        while j >= 0:
            out.appendleft(("*", y[j]))
            j -= 1
        return list(out)

    def replace_invalid_gene_positions(self, positions, fastq_data, rid):
        # :1669-1691
        prev_end = 0
        for i, (s, e) in enumerate(positions):
            if e is not None:
                prev_end = e
            if s is None and e is None:
                nxt = None
                for j in range(i + 1, len(positions)):
                    if positions[j][0] is not None:
                        nxt = positions[j][0]
                        break
                if prev_end is not None and nxt is not None:
                    positions[i] = (prev_end, nxt)
                elif nxt is None and prev_end is not None:
                    positions[i] = (prev_end, len(fastq_data[rid]["sequence"]) - 1)
                else:
                    raise AttributeError("Could not find a valid gene start or end position.")
                assert None not in list(positions[i]), positions
        return positions

    def remove_junk_reads(self, error_rate):
        # :1398-1420 — Python round() (banker's rounding)
        keep, keep_pos, drop, drop_pos = {}, {}, {}, {}
        for rid, nodes in self._readNodes.items():
            allowed = round(len(nodes) * (1 - error_rate))
            bad = sum(1 for n in nodes if n is None)
            if bad <= allowed:
                keep[rid], keep_pos[rid] = self._reads[rid], self._genePositions[rid]
            else:
                drop[rid], drop_pos[rid] = self._reads[rid], self._genePositions[rid]
        return keep, keep_pos, drop, drop_pos

    def get_valid_reads_only(self):
        # :1422-1427
        return {r: g for r, g in self._reads.items() if r not in self._readsToCorrect}

    def collect_reads_in_path(self, path):
        # :1497-1504
        out = set()
        for h in list(path):
            if h in self._nodes:
                out.update(self._nodes[h].get_reads())
        return out

    def remove_non_AMR_associated_nodes(self, genesOfInterest):
        # :2941-2959
        wanted = set()
        for g in genesOfInterest:
            for n in self.get_nodes_containing(g):
                wanted.update(n.get_reads())
        doomed = [
            n for n in self._nodes.values() if not wanted.intersection(n.get_list_of_reads())
        ]
        for n in doomed:
            self.remove_node(n)

    # ------------------------------------------------------------------ clustering
    def find_sublist_indices(self, main, sub):
        return pf.find_sublist_indices(main, sub)

    def is_sublist(self, long_list, sub_list):
        return pf.is_sublist(long_list, sub_list)

    def get_AMR_anchors(self, amr):
        # :2629-2691 (bw_non_self is built from FORWARD neighbours — reference quirk)
        anchors, terminals = set(), {}
        for h in amr:
            terminals[h] = []
            node = self._nodes[h]
            is_anchor, singles = False, []
            fw = [n for n in self.get_forward_neighbors(node) if n.__hash__() != h]
            bw = fw
            if len(fw) == 0 or len(bw) == 0:
                anchors.add(h)
            for r in node.get_reads():
                rn = self._readNodes[r]
                if len(rn) == 1 and rn[0] == h:
                    singles.append(True)
                    terminals[h].append(True)
                    break
                singles.append(False)
                flags = [1 if n in amr else 0 for n in rn]
                for idx in [i for i, n in enumerate(rn) if n == h]:
                    if idx != 0 and idx != len(rn) - 1:
                        if flags[idx - 1] == 0 or flags[idx + 1] == 0:
                            is_anchor = True
                            break
                        terminals[h].append(False)
                    else:
                        terminals[h].append(True)
                if is_anchor:
                    anchors.add(h)
                    break
            if all(singles) or all(terminals[h]):
                f_amr = [n for n in self.get_forward_neighbors(node) if n.__hash__() in amr]
                b_amr = [n for n in self.get_backward_neighbors(node) if n.__hash__() in amr]
                if len(b_amr) == 0 or len(f_amr) == 0:
                    anchors.add(h)
        for h, flags in terminals.items():
            if len(flags) > 0 and flags.count(True) / len(flags) > 0.3:
                anchors.add(h)
        return anchors

    def get_singleton_paths(self, seen, anchors, final_paths, final_cov):
        # :2693-2701
        for a in anchors:
            if a not in seen:
                key = tuple(self.get_genes_in_unitig([a]))
                final_paths[key] = len(set(self._nodes[a].get_list_of_reads()))
                final_cov[key] = [self._nodes[a].get_node_coverage()]

    def get_all_sublists(self, lst, gene_call_subset, threshold, gene, cores):
        # :2711-2723 (the reference fans the i-loop out to a Pool; order of results is i order)
        out = {}
        for i in range(1, len(lst) + 1):
            res = pf.process_combinations_for_i((i, threshold, gene, lst, gene_call_subset))
            for sub in res:
                if sub:
                    out[sub] = res[sub]
        return out

    def get_full_paths(self, node_tree, reads, anchors, threshold, gene_call_subset, gene, cores):
        # :2725-2782
        full_blocks = {}
        for a1 in anchors:
            suffixes = pf.get_suffixes_from_initial_tree(node_tree, a1)
            sub_tree = pf.Tree({r: list(reversed(s)) for r, s in suffixes.items()})
            pf.process_anchors(sub_tree, anchors, a1, full_blocks, reads, node_tree, threshold)
        gene_blocks = {}
        for f in full_blocks:
            subs = self.get_all_sublists(
                self.get_genes_in_unitig(f), gene_call_subset, threshold, gene, cores
            )
            if len(subs) > 0:
                gene_blocks[f] = subs
        kept = pf.filter_blocks({f: full_blocks[f] for f in gene_blocks})
        final_paths, final_cov, seen = {}, {}, set()
        for f1 in kept:
            seen.update(f1)
            if f1 not in gene_blocks:
                continue
            diff = set()
            for o1 in gene_blocks[f1]:
                clash = False
                for f2 in kept:
                    if f1 == f2:
                        continue
                    g2 = self.get_genes_in_unitig(list(f2))
                    if pf.is_sublist(g2, list(o1)) or pf.is_sublist(
                        g2, self.reverse_list_of_genes(list(o1))
                    ):
                        clash = True
                        break
                if not clash:
                    diff.add(o1)
            if len(diff) > 0:
                pick = sorted(
                    list(diff),
                    key=lambda x: (
                        x.count(f"+{gene}") + x.count(f"-{gene}"),
                        gene_blocks[f1][x],
                        len(x),
                    ),
                    reverse=True,
                )[0]
                final_paths[pick] = gene_blocks[f1][pick]
                final_cov[pick] = [self._nodes[n].get_node_coverage() for n in list(f1)]
        return final_paths, seen, final_cov

    def get_paths_for_gene(self, node_tree, gene_call_subset, amr_hashes, threshold, gene, cores):
        # :2809-2829
        anchors = self.get_AMR_anchors(amr_hashes)
        final_paths, seen, final_cov = self.get_full_paths(
            node_tree, self._readNodes, anchors, threshold, gene_call_subset, gene, cores
        )
        self.get_singleton_paths(seen, anchors, final_paths, final_cov)
        return final_paths, final_cov

    def split_into_subpaths(self, gene, paths, path_cov, path_reads, mean_node_coverage=None):
        # :2360-2455
        counter = 1
        clusters, tracking = {}, {}
        if mean_node_coverage is None:
            mean_node_coverage = self.get_mean_node_coverage()
        for path in paths:
            fwd = list(path)
            rev = self.reverse_list_of_genes(fwd)
            tagged = list(path)
            fw_idx, rv_idx = {}, {}
            for g, name in enumerate(fwd):
                if name[1:] == gene:
                    allele = f"{gene}_{counter}"
                    fw_idx[g] = allele
                    rv_idx[len(fwd) - g - 1] = allele
                    clusters[allele], tracking[allele] = [], set()
                    tagged[g] = f"{name[0]}{allele}"
                    counter += 1
            tagged = tuple(tagged)
            for rid, genes in self._reads.items():
                hits = pf.find_sublist_indices(genes, fwd)
                idx = fw_idx
                if not hits:
                    hits = pf.find_sublist_indices(genes, rev)
                    idx = rv_idx
                    if not hits:
                        continue
                if len(hits) != 1:
                    continue
                path_reads.setdefault(tagged, set()).add(rid)
                ps = hits[0][0]
                for gi in idx:
                    assert genes[ps + gi][1:] == gene
                    s, e = self._genePositions[rid][ps + gi]
                    clusters[idx[gi]].append(f"{rid}_{s}_{e}")
                    tracking[idx[gi]].add(f"{rid}_{s}_{e}")
        order = sorted(tracking, key=lambda a: len(tracking[a]), reverse=True)
        doomed = set()
        for i, a1 in enumerate(order):
            if a1 in doomed:
                continue
            for a2 in order[i + 1 :]:
                if a1 != a2 and len(tracking[a1] & tracking[a2]) > 0:
                    doomed.add(a2)
        for d in doomed:
            del clusters[d]
        return clusters, path_reads

    def assign_final_alleles_to_components(self, alleles, clustered, allele_counts, gene):
        # :2784-2807 ('component' is deliberately left to leak across iterations)
        for allele in alleles:
            for entry in alleles[allele]:
                for h in self._readNodes["_".join(entry.split("_")[:-2])]:
                    component = self._nodes[h].get_component()
                    break
                break
            name = "_".join(allele.split("_")[:-1])
            if name not in allele_counts:
                allele_counts[name] = 1
            clustered.setdefault(component, {}).setdefault(gene, {})[
                f"{name}_{allele_counts[name]}"
            ] = alleles[allele]
            allele_counts[name] += 1

    def collect_component_missed_genes(self, by_comp, clustered, allele_counts, gene, path_reads):
        # :2831-2878
        for comp, hashes in by_comp.items():
            clustered.setdefault(comp, {}).setdefault(gene, {})
            if len(clustered[comp][gene]) != 0:
                continue
            if gene not in allele_counts:
                allele_counts[gene] = 1
            name = f"{gene}_{allele_counts[gene]}"
            key = (f"+{name}",)
            clustered[comp][gene][name] = []
            for rid in self.collect_reads_in_path(hashes):
                genes = self._reads[rid]
                for i in [i for i, g in enumerate(genes) if g[1:] == gene]:
                    s, e = self._genePositions[rid][i]
                    clustered[comp][gene][name].append(f"{rid}_{s}_{e}")
                path_reads.setdefault(key, set()).add(rid)
            allele_counts[gene] += 1

    def assign_reads_to_genes(
        self, listOfGenes, cores, allele_counts={}, mean_node_coverage=None, path_threshold=5
    ):
        # :2880-2939
        clustered, path_reads = {}, {}
        if mean_node_coverage is None:
            mean_node_coverage = self.get_mean_node_coverage()
        for gene in listOfGenes:
            amr_hashes = [n.__hash__() for n in self.get_nodes_containing(gene)]
            reads_with_gene = self.collect_reads_in_path(amr_hashes)
            node_tree = pf.construct_suffix_tree({r: self._readNodes[r] for r in reads_with_gene})
            subset = {r: self._reads[r] for r in reads_with_gene}
            rc = {r + "_reverse": self.reverse_list_of_genes(subset[r]) for r in subset}
            subset.update(rc)
            paths, cov = self.get_paths_for_gene(
                node_tree, subset, amr_hashes, mean_node_coverage / 20, gene, cores
            )
            alleles, path_reads = self.split_into_subpaths(
                gene, paths, cov, path_reads, mean_node_coverage
            )
            self.assign_final_alleles_to_components(alleles, clustered, allele_counts, gene)
            by_comp = {}
            for h in amr_hashes:
                by_comp.setdefault(self._nodes[h].get_component(), set()).add(h)
            self.collect_component_missed_genes(by_comp, clustered, allele_counts, gene, path_reads)
        return clustered, path_reads
