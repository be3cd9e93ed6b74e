"""Bubble popping — the tail of every cleaning iteration, drop-in for
amira/construct_graph.py (reference v0.11.0): correct_low_coverage_paths :2196-2250 and its
callees (:1482-1485, :1515-1667, :1693-1955, :1977-2014, :2066-2194, :2252-2265), plus
get_unitigs_in_graph :2961-2975 (row f4).

What grows with the data runs on the device, on the device's own node ids (amira_amd/csrc/amg_bubbles.hip):
  * the search for the paths between junctions — ONE search per start junction that notes every junction it arrives at,
    where the reference searches once per (start, stop) pair (`Engine.junction_paths`);
  * the sketches: the reads' bases are uploaded once per cleaning run (`Sequences`), a node's sketch is hashed straight
    from them under every window that sits on the node, the paths' sketches are united and compared on the device
    (`Engine.path_sketch_overlaps`) — the host sees sketch sizes and overlaps, never a hash.
What decides with those numbers — which path of a bubble is the better one, which reads are rewritten and how — is
host-side orchestration over a handful of paths, as in the reference.  A graph edited on the host (add_node, ...) and
the cases the device path leaves alone (a gene-mer size beyond 16: 4 k > 64 levels of search; two nodes joined by
several edges at the end of a path, where the reference fails; AMG_BUBBLES_BY_OBJECTS=1, the A/B and test switch) go
through the reference-shaped methods below, object by object, with `MinHash` (sourmash's three calls over amg_minhash).
"""
import os
import statistics
import sys
from collections import Counter, defaultdict

import numpy as np

from . import _ffi

_SEQS = {}   # (id(fastq_data), device) -> (fastq_data, Sequences, {read id: row}, number of reads)


def _sequences_for(fastq_data, device):
    """the bases of fastq_data's reads on the device: uploaded when a cleaning run first asks, kept until another
    fastq_data comes along (release_sequences() lets go at once).  fastq_data is taken as read-only, as the reference
    treats it."""
    from .engine import Sequences
    key = (id(fastq_data), int(device))
    got = _SEQS.get(key)
    if got is None or got[0] is not fastq_data or got[3] != len(fastq_data):
        release_sequences()
        ids = list(fastq_data)
        got = _SEQS[key] = (fastq_data, Sequences([fastq_data[r]["sequence"] for r in ids], device),
                            {r: i for i, r in enumerate(ids)}, len(ids))
    return got


def release_sequences():
    for entry in _SEQS.values():
        entry[1].close()
    _SEQS.clear()


class _PathOverlaps:
    """stands in for path_minimizers ({path: [node sketches]}) where the device has united and compared the sketches:
    sizes of the paths' sketches and, for every pair of paths between the same terminals, the hashes they share"""

    def __init__(self, index_of, size, common_of):
        self._index_of, self._size, self._common_of = index_of, size, common_of

    def compare(self, high_nodes, low_nodes):
        """(len(high sketch), len(low sketch), len(high sketch & low sketch))"""
        i, j = self._index_of[tuple(high_nodes)], self._index_of[tuple(low_nodes)]
        return self._size[i], self._size[j], self._common_of[(i, j)]


class MinHash:
    """sourmash.MinHash(n=0, ksize, scaled) as the reference uses it: add_sequence(seq, force=True)
    collects segments, `.hashes` sketches them on the device (lazily, all at once)."""

    def __init__(self, n=0, ksize=21, scaled=0, engine=None):
        assert n == 0 and scaled >= 1, "scaled sketches only"
        self.ksize, self.scaled, self._engine = ksize, scaled, engine
        self._pending, self._set = [], set()

    def add_sequence(self, sequence, force=False):
        self._pending.append(sequence)

    def _flush(self):
        if self._pending:
            eng = self._engine
            if eng is None:
                from .engine import Engine
                eng = self._engine = Engine(0)
            self._set |= eng.minhash(self._pending, [0] * len(self._pending), self.ksize, self.scaled)[0]
            self._pending = []

    def _adopt(self, hashes):
        self._set |= hashes
        self._pending = []

    @property
    def hashes(self):
        self._flush()
        return dict.fromkeys(sorted(self._set), 1)

    def __len__(self):
        self._flush()
        return len(self._set)

    def contained_by(self, other):
        a, b = set(self.hashes), set(other.hashes)
        return len(a & b) / len(a) if a else 0.0


class BubblePopping:
    """mixin of GeneMerGraph (construct_graph.py)"""

    # ------------------------------------------------------------------ alignment / path helpers
    def calculate_path_coverage(self, path):
        if not self._host_edits and not os.environ.get("AMG_BUBBLES_BY_OBJECTS"):   # (the coverages as the device holds them)
            v = self._v()
            coverage, id_of = v.arrays["nodes"]["coverage"], v.node_of_hash
            return statistics.mean([int(coverage[id_of[n[0]]]) for n in path[1:-1]])
        return statistics.mean([self.get_node_by_hash(n[0]).get_node_coverage() for n in path[1:-1]])

    def get_direction_between_two_nodes(self, source_node_hash, target_node_hash):
        forward, _ = self.get_edges_between_nodes(self.get_node_by_hash(source_node_hash),
                                                  self.get_node_by_hash(target_node_hash))
        return forward.get_targetNodeDirection() * -1

    def reverse_gene(self, gene):
        return {"+": "-" + gene[1:], "-": "+" + gene[1:], "*": "*"}.get(gene[0])

    def reverse_gene_alignment(self, alignment):
        return [(self.reverse_gene(col[0]), self.reverse_gene(col[1])) for col in alignment[::-1]]

    def count_snps_in_alignment(self, aln):
        return len([c for c in aln if c[0] != c[1] and "*" not in (c[0], c[1])])

    def count_indels_in_alignment(self, aln):
        return len([c for c in aln if c[0] != c[1] and "*" in (c[0], c[1])])

    def get_gene_mer_strings(self, genes_on_read):
        k = self.get_kmerSize()
        return [tuple(genes_on_read[i:i + k]) for i in range(len(genes_on_read) - (k - 1))]

    def get_path_to_alignment_mapping(self, alignment):
        higher_mapping, lower_mapping = {}, {}
        for column, (high, low) in enumerate(alignment):
            if low != "*":
                lower_mapping[len(lower_mapping)] = column
            if high != "*":
                higher_mapping[len(higher_mapping)] = column
        return higher_mapping, lower_mapping

    def longest_common_sublist(self, a, b):
        """longest contiguous run shared by a and b; the first longest one in (i, j) order"""
        longest = end_a = end_b = 0
        run = {}
        for i, x in enumerate(a):
            nxt = {}
            for j, y in enumerate(b):
                if x == y:
                    length = run.get(j - 1, 0) + 1
                    nxt[j] = length
                    if length > longest:
                        longest, end_a, end_b = length, i + 1, j + 1
            run = nxt
        return a[end_a - longest:end_a], (end_a - longest, end_a - 1), (end_b - longest, end_b - 1)

    def compare_paths(self, lower_coverage_genes, fw_higher_coverage_genes):
        fw_alignment = self.needleman_wunsch(fw_higher_coverage_genes, lower_coverage_genes)
        return (fw_alignment, self.reverse_gene_alignment(fw_alignment),
                self.count_snps_in_alignment(fw_alignment), self.count_indels_in_alignment(fw_alignment))

    def reorient_alignment(self, gene_mers_on_read, fw_genes_in_path_counter, bw_genes_in_path_counter,
                           fw_alignment, rv_alignment):
        on_read = Counter(gene_mers_on_read)
        fw_count = len(on_read & fw_genes_in_path_counter)
        rv_count = len(on_read & bw_genes_in_path_counter)
        if fw_count > rv_count:
            return fw_alignment
        if rv_count > fw_count:
            return rv_alignment
        return None  # equally close (or both strangers): the read is left alone

    # ------------------------------------------------------------------ re-writing one read
    def correct_genes_on_read(self, genes_on_read, first_shared_read_index, last_shared_read_index,
                              alignment_subset, read_id):
        core = [col[0] for col in alignment_subset if col[0] != "*"]
        self.get_reads()[read_id] = (genes_on_read[:first_shared_read_index] + core
                                     + genes_on_read[last_shared_read_index + 1:])
        return self.get_reads()[read_id]

    def get_gene_position_prefix(self, gene_positions, first_shared_read_index):
        return gene_positions[:first_shared_read_index]

    def get_gene_position_suffix(self, gene_positions, last_shared_read_index):
        return gene_positions[last_shared_read_index + 1:]

    def get_gene_position_core(self, gene_positions, first_shared_read_index, last_shared_read_index):
        return gene_positions[first_shared_read_index:last_shared_read_index + 1]

    def get_new_gene_position_core(self, alignment_subset, core_gene_positions):
        used, fresh = 0, []
        for high, low in alignment_subset:
            if high == "*":
                used += 1          # a gene of the read that the better path does not have
            elif low != high:
                fresh.append((None, None))   # a gene the read did not have: position inferred later
            else:
                fresh.append(core_gene_positions[used])
                used += 1
        return fresh

    def join_gene_position_ends_with_core(self, position_prefix, position_suffix, new_core_gene_positions):
        joined = new_core_gene_positions
        if len(position_prefix) != 0:
            joined = position_prefix + joined
        if len(position_suffix) != 0:
            joined = joined + position_suffix
        return joined

    def correct_gene_positions_on_read(self, first_shared_read_index, last_shared_read_index,
                                       alignment_subset, read_id, fastq_data):
        positions = self.get_gene_positions()[read_id][:]
        core = self.get_new_gene_position_core(
            alignment_subset,
            self.get_gene_position_core(positions, first_shared_read_index, last_shared_read_index))
        joined = self.join_gene_position_ends_with_core(
            self.get_gene_position_prefix(positions, first_shared_read_index),
            self.get_gene_position_suffix(positions, last_shared_read_index), core)
        self.get_gene_positions()[read_id] = self.replace_invalid_gene_positions(joined, fastq_data, read_id)
        n_genes, n_pos = len(self.get_reads()[read_id]), len(self.get_gene_positions()[read_id])
        assert n_genes == n_pos, f"{n_genes}/{n_pos}"
        return self.get_gene_positions()[read_id]

    def modify_alignment_subset(self, alignment_subset, genes_on_read):
        true_path = [col[0] for col in alignment_subset if col[0] != "*"]
        return alignment_subset if true_path == genes_on_read else self.needleman_wunsch(true_path, genes_on_read)

    # ------------------------------------------------------------------ sequences and sketches
    def get_read_sequence_for_path(self, read_id, path, fastq_data):
        nodes_on_read = self.get_readNodes()[read_id]
        spans = self.get_readNodePositions()[read_id]
        assert len(nodes_on_read) == len(spans)
        shared = [i for i, h in enumerate(nodes_on_read) if h in path]
        if not shared:
            return None
        sequence = fastq_data[read_id]["sequence"]
        start, end = spans[shared[0]][0], spans[shared[-1]][1]
        if start is None:
            assert shared[0] == 0, spans
            start = 0
        if end is None:
            assert shared[-1] == len(spans) - 1, spans
            end = len(sequence) - 1
        assert start >= 0
        assert end < len(sequence)
        return sequence[start:end + 1]

    def get_minhash_for_path(self, path, reads_in_path, fastq_data):
        sketch = MinHash(n=0, ksize=9, scaled=1, engine=self._engine)
        reads_with_positions = set()
        for read_id in reads_in_path:
            sequence = self.get_read_sequence_for_path(read_id, path, fastq_data)
            if sequence is not None:
                sketch.add_sequence(sequence, force=True)
                reads_with_positions.add(f"{read_id}")
        return sketch, reads_with_positions

    def _node_segments(self, node_hash, fastq_data):
        """the stretch of every read under every occurrence of the node (:2151-2157)"""
        segments = []
        for read in self.get_node_by_hash(node_hash).get_reads():
            sequence = fastq_data[read]["sequence"]
            spans = self.get_readNodePositions()[read]
            for i, h in enumerate(self.get_readNodes()[read]):
                if h == node_hash:
                    segments.append(sequence[spans[i][0]:spans[i][1] + 1])
        return segments

    def get_minhash_of_nodes(self, batch, node_minhashes, fastq_data):
        """one device launch sketches every node of the batch (ksize 11, scaled 10)"""
        batch = list(batch)
        segments, owners = [], []
        for index, node_hash in enumerate(batch):
            for s in self._node_segments(node_hash, fastq_data):
                segments.append(s)
                owners.append(index)
        sketches = self._engine.minhash(segments, owners, 11, 10) if segments else {}
        for index, node_hash in enumerate(batch):
            mh = MinHash(n=0, ksize=11, scaled=10, engine=self._engine)
            mh._adopt(sketches.get(index, set()))
            node_minhashes[node_hash] = mh

    def get_minhash_of_path(self, batch, path_minimizers, node_minhashes):
        for path_tuple in batch:
            path_minimizers[path_tuple].extend(node_minhashes[h] for h in path_tuple)

    def get_minhashes_for_paths(self, sorted_filtered_paths, fastq_data, cores):
        path_minimizers = defaultdict(set)
        node_minhashes = {}
        for path_tuple, _coverage in sorted_filtered_paths:
            hashes = tuple(p[0] for p in path_tuple)
            node_minhashes.update((h, None) for h in hashes if h not in node_minhashes)
            path_minimizers[hashes] = []
        self.get_minhash_of_nodes(list(node_minhashes), node_minhashes, fastq_data)
        self.get_minhash_of_path(list(path_minimizers), path_minimizers, node_minhashes)
        assert not any(v is None for v in path_minimizers.values())
        return path_minimizers

    def get_minimizers_from_minhashes(self, path, path_minimizers):
        union = set()
        for sketch in path_minimizers[tuple(path)]:
            union.update(sketch.hashes)
        return union

    # ------------------------------------------------------------------ junctions, paths, operations
    def identify_potential_bubble_starts(self):
        starts = {}
        for node in self.all_nodes():
            for hashes, direction in ((node.get_forward_edge_hashes(), 1), (node.get_backward_edge_hashes(), -1)):
                if len(hashes) > 1:
                    starts.setdefault(node.get_component(), []).append((node.__hash__(), direction))
        return starts

    def get_all_paths_between_junctions_in_component(self, potential_bubble_starts_component, max_distance,
                                                     cores=1):
        unique_paths = set()
        for start in potential_bubble_starts_component:
            for stop in potential_bubble_starts_component:
                if start[0] == stop[0]:
                    continue
                candidates = self.new_find_paths_between_nodes(start[0], stop[0], max_distance, start[1])
                # keep the paths that leave `start` and enter `stop` the way the junctions say
                valid = [p for p in candidates if p[0] == start
                         and (p[-1][0], self.get_direction_between_two_nodes(p[-2][0], p[-1][0])) == stop]
                if len(valid) > 1:   # a bubble needs two ways through
                    for p in valid:
                        mirrored = [(h, -d) for h, d in reversed(p)]
                        unique_paths.add(tuple(min(p, mirrored)))
        return list(unique_paths)

    def separate_paths_by_terminal_nodes(self, sorted_filtered_paths):
        by_terminals = {}
        for entry in sorted_filtered_paths:
            nodes = entry[0]
            by_terminals.setdefault(tuple(sorted([nodes[0][0], nodes[-1][0]])), []).append(entry)
        ranked = sorted(by_terminals.items(), key=lambda kv: max(len(e[0]) for e in kv[1]), reverse=True)
        return dict(ranked)

    def filter_paths_between_bubble_starts(self, unique_paths):
        """the paths that hold no other path (read either way along) as a run of their own nodes, shortest first, with
        their mean inner coverage (:2125-2146).  The reference asks a suffix tree over all paths for the paths that hold
        path i; here every (node, direction) item lists where it occurs, and a path is looked for where its first item
        occurs: the same hits, in time proportional to the occurrences instead of to all paths per question."""
        unique_paths = sorted(list(unique_paths), key=len)
        keys = [tuple(q) for q in unique_paths]
        occurrences = {}
        for j, q in enumerate(keys):
            for at, item in enumerate(q):
                occurrences.setdefault(item, []).append((j, at))
        filtered_paths, contained = [], set()
        for i, p in enumerate(unique_paths):
            if i in contained:
                continue
            m = len(p)
            for query in (keys[i], keys[i][::-1]):
                for j, at in occurrences.get(query[0], ()) if m else ():
                    if j != i and keys[j][at:at + m] == query:
                        contained.add(j)
            if m > 2:
                filtered_paths.append((p, self.calculate_path_coverage(p)))
        return filtered_paths

    def define_correction_operations(self, paths, path_coverages, reads_to_correct, correction_operations,
                                     path_minimizers, seen_nodes, threshold):
        path_coverages.extend(p[1] for p in paths)
        corrected_paths = set()
        for i, (high_entry, high_coverage) in enumerate(paths):
            high_nodes = [n[0] for n in high_entry]
            if tuple(high_nodes) in corrected_paths or any(n in seen_nodes for n in high_nodes):
                continue
            on_device = isinstance(path_minimizers, _PathOverlaps)
            high_sketch = None if on_device else self.get_minimizers_from_minhashes(high_nodes, path_minimizers)
            for low_entry, low_coverage in paths[i + 1:]:
                low_nodes = [n[0] for n in low_entry]
                if tuple(low_nodes) in corrected_paths or any(n in seen_nodes for n in low_nodes):
                    continue
                if on_device:
                    n_high, n_low, common = path_minimizers.compare(high_nodes, low_nodes)
                else:
                    low_sketch = self.get_minimizers_from_minhashes(low_nodes, path_minimizers)
                    n_high, n_low, common = len(high_sketch), len(low_sketch), len(high_sketch & low_sketch)
                if max(common / n_low, common / n_high) > threshold:
                    operation = (tuple(low_nodes), tuple(high_nodes), low_coverage, high_coverage)
                    correction_operations.add(operation)
                    corrected_paths.add(tuple(low_nodes))
                    for n in low_nodes:
                        if n not in high_nodes:
                            seen_nodes[n] = operation
        return path_coverages

    def get_path_reads_to_correct(self, reads_to_correct, seen_nodes):
        if not self._host_edits and seen_nodes and not os.environ.get("AMG_BUBBLES_BY_OBJECTS"):
            # the nodes' reads straight from the device's node -> reads lists
            v = self._v()
            off, rows, ids, id_of = v.arrays["node_reads_off"], v.arrays["node_reads"], self._read_ids, v.node_of_hash
            for node_hash, operation in seen_nodes.items():
                i = id_of[node_hash]
                for r in rows[off[i]:off[i + 1]].tolist():
                    reads_to_correct.setdefault(ids[r], operation)
            return
        for node_hash, operation in seen_nodes.items():
            for read in self.get_node_by_hash(node_hash).get_reads():
                reads_to_correct.setdefault(read, operation)

    def correct_bubble_paths(self, bubbles, fastq_data, path_minimizers, genesOfInterest, min_path_coverage,
                             threshold=0.80):
        seen_nodes, correction_operations, reads_to_correct, path_coverages = {}, set(), {}, []
        for terminals, entries in bubbles.items():
            if len(entries) > 1:
                by_coverage = sorted(list(entries), key=lambda e: e[1], reverse=True)
                path_coverages = self.define_correction_operations(
                    by_coverage, path_coverages, reads_to_correct, correction_operations, path_minimizers,
                    seen_nodes, threshold)
        self.get_path_reads_to_correct(reads_to_correct, seen_nodes)
        k = self.get_kmerSize()
        plans = {}   # operation -> (alignment, mirrored alignment, gene-mer counter, mirrored counter)
        spelled = {}   # (a better path is the better path of several worse ones)

        def genes_of(path):
            got = spelled.get(path)
            if got is None:
                got = spelled[path] = self.get_genes_in_unitig(list(path))
            return list(got)

        for operation in correction_operations:
            better = genes_of(operation[1])
            worse = genes_of(operation[0])
            fw_alignment, rv_alignment, _, _ = self.compare_paths(worse, better)
            if any(low[1:] in genesOfInterest and high[1:] not in genesOfInterest for high, low in fw_alignment):
                continue   # the correction would delete a gene of interest
            mers = [tuple(worse[i:i + k]) for i in range(len(worse) - (k - 1))]
            plans[operation] = (fw_alignment, rv_alignment, Counter(mers),
                                Counter(tuple(self.reverse_list_of_genes(list(m))) for m in mers))
        for read_id, operation in reads_to_correct.items():
            if operation not in plans:
                continue
            fw_alignment, rv_alignment, fw_counter, bw_counter = plans[operation]
            genes_on_read = self.get_reads()[read_id][:]
            read_alignment = self.reorient_alignment(self.get_gene_mer_strings(genes_on_read), fw_counter,
                                                     bw_counter, fw_alignment, rv_alignment)
            if read_alignment is None:
                continue
            _, lower_mapping = self.get_path_to_alignment_mapping(read_alignment)
            worse_on_alignment = [col[1] for col in read_alignment if col[1] != "*"]
            _, (start_path, end_path), (first_shared, last_shared) = self.longest_common_sublist(
                worse_on_alignment, genes_on_read)
            subset = read_alignment[lower_mapping[start_path]:lower_mapping[end_path] + 1]
            subset = self.modify_alignment_subset(subset, genes_on_read[first_shared:last_shared + 1])
            if len(subset) != 0:
                self.correct_genes_on_read(genes_on_read, first_shared, last_shared, subset, read_id)
                self.correct_gene_positions_on_read(first_shared, last_shared, subset, read_id, fastq_data)
        return path_coverages

    # ------------------------------------------------------------------ the device's share
    def _junction_paths_on_device(self):
        """{component: [path, ...]} with every path get_all_paths_between_junctions_in_component would add to its set,
        in the order it would add them, as lists of (node hash, direction) — and the components in order; None when the
        device path does not apply (module docstring)"""
        max_distance = self.get_kmerSize() * 4
        if self._host_edits or max_distance > 64 or max_distance < 2 or os.environ.get("AMG_BUBBLES_BY_OBJECTS"):
            return None
        found = self._engine.junction_paths(max_distance)
        if found["flags"]:
            return None
        v = self._v()
        nodes = v.arrays["nodes"]
        component_of = nodes["component"]
        components = np.unique(component_of[nodes["alive"] != 0]).tolist()   # (= sorted({n.get_component() for live n}))
        start_component = component_of[found["junction_node"]]
        by_component = {int(c): [] for c in np.unique(start_component).tolist()}
        off, ids, dirs = found["path_off"].tolist(), found["path_node"], found["path_dir"].tolist()
        v.ensure_hashes(np.unique(ids).tolist())
        hashes = [v.node_hash[i] for i in ids.tolist()]
        path_component = start_component[found["path_start"]].tolist()
        for p, c in enumerate(path_component):
            by_component[c].append(list(zip(hashes[off[p]:off[p + 1]], dirs[off[p]:off[p + 1]])))
        return components, by_component

    def _path_overlaps_on_device(self, bubbles, fastq_data):
        """what define_correction_operations asks of the sketches, for every pair of paths it can ask about: the paths
        between the same terminals, the one of higher coverage first (the order correct_bubble_paths sorts them in)"""
        if self._host_edits or os.environ.get("AMG_BUBBLES_BY_OBJECTS"):
            return None
        index_of, path_ids, groups = {}, [], []
        node_id = self._v().node_of_hash
        for entries in bubbles.values():
            if len(entries) < 2:
                continue
            group = []
            for entry, _coverage in sorted(list(entries), key=lambda e: e[1], reverse=True):
                key = tuple(n[0] for n in entry)
                if key not in index_of:
                    index_of[key] = len(path_ids)
                    path_ids.append([node_id[h] for h in key])
                group.append(index_of[key])
            groups.append(group)
        if not groups:
            return _PathOverlaps({}, [], {})
        size, common_of = [0] * len(path_ids), {}
        try:
            _, seqs, row_of, _ = _sequences_for(fastq_data, self._engine.device)
            rows = np.fromiter((row_of.get(r, -1) for r in self._read_ids), np.int32, len(self._read_ids))
            identity = len(rows) <= seqs.n and bool((rows == np.arange(len(rows), dtype=np.int32)).all())

            def compare(some):
                """the paths of these groups in one call; a call too big for the device's buffers is cut in two"""
                local, pair_a, pair_b = {}, [], []
                for group in some:
                    for p in group:
                        local.setdefault(p, len(local))
                    for a in range(len(group)):
                        for b in range(a + 1, len(group)):
                            pair_a.append(local[group[a]])
                            pair_b.append(local[group[b]])
                order = list(local)
                path_off = np.zeros(len(order) + 1, np.int64)
                np.cumsum([len(path_ids[p]) for p in order], out=path_off[1:])
                flat = np.fromiter((x for p in order for x in path_ids[p]), np.int32, int(path_off[-1]))
                try:
                    got_size, got_common = self._engine.path_sketch_overlaps(
                        seqs, None if identity else rows, 11, 10, path_off, flat, pair_a, pair_b)
                except _ffi.AmgError as err:
                    if err.code != _ffi.E_NOMEM or len(some) < 2:
                        raise
                    compare(some[:len(some) // 2])
                    compare(some[len(some) // 2:])
                    return
                for p, n in zip(order, got_size.tolist()):
                    size[p] = n
                for a, b, n in zip(pair_a, pair_b, got_common.tolist()):
                    common_of[(order[a], order[b])] = n

            compare(groups)
        except (_ffi.AmgError, KeyError, TypeError):
            return None   # (a read without a sequence, a position below zero, ...: the objects' way says what the reference says)
        return _PathOverlaps(index_of, size, common_of)

    def correct_low_coverage_paths(self, fastq_data, genesOfInterest, cores, min_path_coverage,
                                   components_to_skip, use_minimizers=False):
        """pop bubbles: between every pair of junctions, reads on the lower-coverage way through are
        re-written to the higher-coverage one when their sequences' MinHash containment exceeds 0.8"""
        assert self.get_gene_positions()
        max_distance = self.get_kmerSize() * 4
        on_device = self._junction_paths_on_device()
        if on_device is not None:
            components, paths_of = on_device
        else:
            starts = self.identify_potential_bubble_starts()
            components = self.components()
        path_coverages = []
        for component in components:
            sys.stderr.write(f"\n\tAmira: popping bubbles using 1 CPU for component {component} / {len(components)}\n")
            if on_device is not None:
                if component in components_to_skip or component not in paths_of:
                    continue
                unique_paths = set()
                for p in paths_of[component]:
                    unique_paths.add(tuple(sorted([p, [(h, -d) for h, d in reversed(p)]])[0]))
                unique_paths = list(unique_paths)
            else:
                if component in components_to_skip or component not in starts:
                    continue
                unique_paths = self.get_all_paths_between_junctions_in_component(starts[component], max_distance, cores)
            shortest_first = sorted(self.filter_paths_between_bubble_starts(unique_paths), key=lambda e: len(e[0]))
            bubbles = self.separate_paths_by_terminal_nodes(shortest_first)
            path_minimizers = None
            if use_minimizers:
                path_minimizers = self._path_overlaps_on_device(bubbles, fastq_data)
                if path_minimizers is None:
                    path_minimizers = self.get_minhashes_for_paths(shortest_first, fastq_data, cores)
            path_coverages += self.correct_bubble_paths(bubbles, fastq_data, path_minimizers, genesOfInterest,
                                                        min_path_coverage)
        return self.get_reads(), self.get_gene_positions(), path_coverages, min_path_coverage

    # ------------------------------------------------------------------ unitigs (row f4)
    def get_unitigs_in_graph(self, outfile):
        unitigs = set()
        for node in self.all_nodes():
            if len(self.get_all_neighbors(node)) > 2:
                continue
            path = self.get_linear_path_for_node(node, True)
            path = min(path, path[::-1])
            genes = self.get_genes_in_unitig(path)
            canonical = min(genes, self.reverse_list_of_genes(genes))
            unitigs.add((tuple(canonical), len(self.collect_reads_in_path(path))))
        with open(outfile, "w") as handle:
            handle.write("\n".join(f"{','.join(genes)}\t{support}" for genes, support in unitigs))
