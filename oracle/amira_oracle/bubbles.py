"""Bubble popping.  TEST INFRASTRUCTURE ONLY (oracle/README.md).

Restates the tail of every cleaning iteration, amira/construct_graph.py (reference v0.11.0):
correct_low_coverage_paths :2196-2250 and what it calls — junction discovery :2252-2265, simple
paths between junctions :2066-2098 (new_find_paths_between_nodes :2292-2342), containment filter
:2125-2146, MinHash per node / per path :2148-2194 (sourmash restated in minhash.py), pairing of
paths by their terminals :2100-2115, correction operations :1753-1824, read re-writing
:1833-1955 with its alignment / position helpers :1515-1751, :1977-2014.  Quirks are kept.
"""
import statistics
import sys
from collections import Counter, defaultdict

from . import paths as pf
from .minhash import MinHash


class BubbleMixin:
    # ------------------------------------------------------------------ small helpers
    def calculate_path_coverage(self, path):
        # :1482-1485 — interior nodes only
        return statistics.mean([self._nodes[n[0]].get_node_coverage() for n in path[1:-1]])

    def get_direction_between_two_nodes(self, source_hash, target_hash):
        # :1515-1522
        s2t, _ = self.get_edges_between_nodes(self._nodes[source_hash], self._nodes[target_hash])
        return s2t.get_targetNodeDirection() * -1

    def reverse_gene(self, gene):
        # :1524-1530
        if gene[0] == "+":
            return "-" + gene[1:]
        if gene[0] == "-":
            return "+" + gene[1:]
        if gene[0] == "*":
            return "*"

    def reverse_gene_alignment(self, alignment):
        # :1532-1539
        return [(self.reverse_gene(a), self.reverse_gene(b)) for a, b in reversed(alignment)]

    def count_snps_in_alignment(self, aln):
        return sum(1 for a, b in aln if a != b and a != "*" and b != "*")  # :1577

    def count_indels_in_alignment(self, aln):
        return sum(1 for a, b in aln if a != b and (a == "*" or b == "*"))  # :1580

    def get_gene_mer_strings(self, genes_on_read):
        # :1583-1589
        k = self._kmerSize
        return [tuple(genes_on_read[i:i + k]) for i in range(len(genes_on_read) - (k - 1))]

    def get_path_to_alignment_mapping(self, alignment):
        # :1977-1990
        higher, lower = {}, {}
        for i, (a, b) in enumerate(alignment):
            if b != "*":
                lower[len(lower)] = i
            if a != "*":
                higher[len(higher)] = i
        return higher, lower

    def longest_common_sublist(self, a, b):
        # :1992-2014 — first maximum in row-major order wins
        best, end_a, end_b = 0, 0, 0
        prev = [0] * (len(b) + 1)
        for i in range(1, len(a) + 1):
            cur = [0] * (len(b) + 1)
            for j in range(1, len(b) + 1):
                if a[i - 1] == b[j - 1]:
                    cur[j] = prev[j - 1] + 1
                    if cur[j] > best:
                        best, end_a, end_b = cur[j], i, j
            prev = cur
        return a[end_a - best:end_a], (end_a - best, end_a - 1), (end_b - best, end_b - 1)

    # ------------------------------------------------------------------ read sequences, MinHash
    def get_read_sequence_for_path(self, read_id, path, fastq_data):
        # :1541-1565
        assert len(self._readNodes[read_id]) == len(self._readNodePositions[read_id])
        shared = [i for i, h in enumerate(self._readNodes[read_id]) if h in path]
        if not shared:
            return None
        seq = fastq_data[read_id]["sequence"]
        first, last = shared[0], shared[-1]
        start = self._readNodePositions[read_id][first][0]
        if start is None:
            assert first == 0, self._readNodePositions[read_id]
            start = 0
        end = self._readNodePositions[read_id][last][1]
        if end is None:
            assert last == len(self._readNodePositions[read_id]) - 1, self._readNodePositions[read_id]
            end = len(seq) - 1
        assert start >= 0
        assert end < len(seq)
        return seq[start:end + 1]

    def get_minhash_for_path(self, path, reads_in_path, fastq_data):
        # :1567-1575
        mh = MinHash(n=0, ksize=9, scaled=1)
        with_positions = set()
        for read_id in reads_in_path:
            seq = self.get_read_sequence_for_path(read_id, path, fastq_data)
            if seq is not None:
                mh.add_sequence(seq, force=True)
                with_positions.add(f"{read_id}")
        return mh, with_positions

    def get_minhash_of_nodes(self, batch, node_minhashes, fastq_data):
        # :2148-2158 — every occurrence of the node on every read of its list
        for h in batch:
            mh = MinHash(n=0, ksize=11, scaled=10)
            for read in self._nodes[h].get_reads():
                seq = fastq_data[read]["sequence"]
                for i, n in enumerate(self._readNodes[read]):
                    if n == h:
                        p = self._readNodePositions[read][i]
                        mh.add_sequence(seq[p[0]:p[1] + 1], force=True)
            node_minhashes[h] = mh

    def get_minhash_of_path(self, batch, path_minimizers, node_minhashes):
        # :2160-2163
        for path_tuple in batch:
            for h in path_tuple:
                path_minimizers[path_tuple].append(node_minhashes[h])

    def get_minhashes_for_paths(self, sorted_filtered_paths, fastq_data, cores):
        # :2165-2194 (joblib threads in the reference; the result does not depend on them)
        path_minimizers = defaultdict(set)
        node_minhashes = {}
        for path_tuple, _ in sorted_filtered_paths:
            path = [p[0] for p in path_tuple]
            for h in path:
                node_minhashes.setdefault(h, None)
            path_minimizers[tuple(path)] = []
        self.get_minhash_of_nodes(list(node_minhashes), node_minhashes, fastq_data)
        self.get_minhash_of_path(list(path_minimizers), path_minimizers, node_minhashes)
        assert not any(v is None for v in path_minimizers.values())
        return path_minimizers

    def get_minimizers_from_minhashes(self, path, path_minimizers):
        # :1747-1751
        out = set()
        for mh in path_minimizers[tuple(path)]:
            out.update(mh.hashes)
        return out

    # ------------------------------------------------------------------ junctions and the paths between them
    def identify_potential_bubble_starts(self):
        # :2252-2265
        starts = {}
        for node in self.all_nodes():
            if len(node.get_forward_edge_hashes()) > 1:
                starts.setdefault(node.get_component(), []).append((node.__hash__(), 1))
            if len(node.get_backward_edge_hashes()) > 1:
                starts.setdefault(node.get_component(), []).append((node.__hash__(), -1))
        return starts

    def get_all_paths_between_junctions_in_component(self, starts, max_distance, cores=1):
        # :2066-2098 — ordered pairs; a pair contributes only when it has MORE than one valid path
        unique = set()
        for start_hash, start_dir in starts:
            for stop_hash, stop_dir in starts:
                if start_hash == stop_hash:
                    continue
                found = self.new_find_paths_between_nodes(start_hash, stop_hash, max_distance, start_dir)
                valid = [p for p in found
                         if p[0] == (start_hash, start_dir)
                         and (p[-1][0], self.get_direction_between_two_nodes(p[-2][0], p[-1][0]))
                         == (stop_hash, stop_dir)]
                if len(valid) > 1:
                    for p in valid:
                        flipped = list(reversed([(h, d * -1) for h, d in p]))
                        unique.add(tuple(sorted([p, flipped])[0]))
        return list(unique)

    def filter_paths_between_bubble_starts(self, unique_paths):
        # :2125-2146 — shortest first; a path that contains an earlier one (either way round) is dropped
        unique_paths = sorted(list(unique_paths), key=len)
        tree = pf.Tree({i: p for i, p in enumerate(unique_paths)})
        kept, contained = [], set()
        for i, p in enumerate(unique_paths):
            if i in contained:
                continue
            fwd, rev = list(p), list(reversed(list(p)))
            for j in [pid for pid, _ in tree.find_all(fwd)] + [pid for pid, _ in tree.find_all(rev)]:
                if i != j:
                    contained.add(j)
            if len(p) > 2:
                kept.append((p, self.calculate_path_coverage(p)))
        return kept

    def separate_paths_by_terminal_nodes(self, sorted_filtered_paths):
        # :2100-2115
        paired = {}
        for p in sorted_filtered_paths:
            key = tuple(sorted([p[0][0][0], p[0][-1][0]]))
            paired.setdefault(key, []).append(p)
        return dict(sorted(paired.items(), key=lambda kv: max(len(path[0]) for path in kv[1]), reverse=True))

    # ------------------------------------------------------------------ choosing what to correct
    def define_correction_operations(self, paths, path_coverages, reads_to_correct, correction_operations,
                                     path_minimizers, seen_nodes, threshold):
        # :1753-1824
        corrected = set()
        for p in paths:
            path_coverages.append(p[1])
        for i in range(len(paths)):
            high_path, high_cov = paths[i]
            high_path = [n[0] for n in high_path]
            high_set, high_tuple = set(high_path), tuple(high_path)
            if high_tuple in corrected:
                continue
            if any(n in seen_nodes for n in high_path):
                continue
            high_min = self.get_minimizers_from_minhashes(high_path, path_minimizers)
            for low_path, low_cov in paths[i + 1:]:
                low_path = [n[0] for n in low_path]
                low_tuple = tuple(low_path)
                if low_tuple in corrected:
                    continue
                if any(n in seen_nodes for n in low_path):
                    continue
                low_min = self.get_minimizers_from_minhashes(low_path, path_minimizers)
                shared = len(high_min & low_min)
                containment = max([shared / len(low_min), shared / len(high_min)])
                if containment > threshold:
                    operation = (low_tuple, high_tuple, low_cov, high_cov)
                    correction_operations.add(operation)
                    corrected.add(low_tuple)
                    for n in low_path:
                        if n not in high_set:
                            seen_nodes[n] = operation
        return path_coverages

    def get_path_reads_to_correct(self, reads_to_correct, seen_nodes):
        # :1826-1831
        for n, operation in seen_nodes.items():
            for read in self._nodes[n].get_reads():
                if read not in reads_to_correct:
                    reads_to_correct[read] = operation

    def compare_paths(self, lower_genes, fw_higher_genes):
        # :1737-1745
        fw = self.needleman_wunsch(fw_higher_genes, lower_genes)
        return fw, self.reverse_gene_alignment(fw), self.count_snps_in_alignment(fw), self.count_indels_in_alignment(fw)

    def reorient_alignment(self, gene_mers_on_read, fw_counter, bw_counter, fw_alignment, rv_alignment):
        # :1591-1612 — distinct shared gene-mers decide; ties (also 0 : 0) give None
        on_read = Counter(gene_mers_on_read)
        fw, rv = len(on_read & fw_counter), len(on_read & bw_counter)
        if fw > rv:
            return fw_alignment
        if rv > fw:
            return rv_alignment
        return None

    # ------------------------------------------------------------------ re-writing a read
    def correct_genes_on_read(self, genes_on_read, first_shared, last_shared, alignment_subset, read_id):
        # :1616-1628
        core = [c[0] for c in alignment_subset if c[0] != "*"]
        self._reads[read_id] = genes_on_read[:first_shared] + core + genes_on_read[last_shared + 1:]
        return self._reads[read_id]

    def get_gene_position_prefix(self, gene_positions, first_shared):
        return gene_positions[:first_shared]  # :1630

    def get_gene_position_suffix(self, gene_positions, last_shared):
        return gene_positions[last_shared + 1:]  # :1633

    def get_gene_position_core(self, gene_positions, first_shared, last_shared):
        return gene_positions[first_shared:last_shared + 1]  # :1636

    def get_new_gene_position_core(self, alignment_subset, core_positions):
        # :1641-1654
        at, out = 0, []
        for a, b in alignment_subset:
            if a != "*":
                if b != a:
                    out.append((None, None))
                else:
                    out.append(core_positions[at])
                    at += 1
            else:
                at += 1
        return out

    def join_gene_position_ends_with_core(self, prefix, suffix, core):
        # :1656-1667
        if len(prefix) != 0 and len(suffix) != 0:
            return prefix + core + suffix
        if len(prefix) != 0:
            return prefix + core
        if len(suffix) != 0:
            return core + suffix
        return core

    def correct_gene_positions_on_read(self, first_shared, last_shared, alignment_subset, read_id, fastq_data):
        # :1693-1729
        positions = self._genePositions[read_id][:]
        core = self.get_new_gene_position_core(
            alignment_subset, self.get_gene_position_core(positions, first_shared, last_shared))
        joined = self.join_gene_position_ends_with_core(
            self.get_gene_position_prefix(positions, first_shared),
            self.get_gene_position_suffix(positions, last_shared), core)
        self._genePositions[read_id] = self.replace_invalid_gene_positions(joined, fastq_data, read_id)
        assert len(self._reads[read_id]) == len(self._genePositions[read_id]), (
            str(len(self._reads[read_id])) + "/" + str(len(self._genePositions[read_id])))
        return self._genePositions[read_id]

    def modify_alignment_subset(self, alignment_subset, genes_on_read):
        # :1731-1735
        true_path = [c[0] for c in alignment_subset if c[0] != "*"]
        if true_path == genes_on_read:
            return alignment_subset
        return self.needleman_wunsch(true_path, genes_on_read)

    def correct_bubble_paths(self, bubbles, fastq_data, path_minimizers, genesOfInterest, min_path_coverage,
                             threshold=0.80):
        # :1833-1955
        seen_nodes, correction_operations, reads_to_correct, path_coverages = {}, set(), {}, []
        for pair in bubbles:
            if len(bubbles[pair]) > 1:
                paths = sorted(list(bubbles[pair]), key=lambda x: x[1], reverse=True)
                path_coverages = self.define_correction_operations(
                    paths, path_coverages, reads_to_correct, correction_operations, path_minimizers,
                    seen_nodes, threshold)
        self.get_path_reads_to_correct(reads_to_correct, seen_nodes)
        k = self._kmerSize
        fw_alignments, bw_alignments, fw_counters, bw_counters = {}, {}, {}, {}
        for operation in correction_operations:
            high_genes = self.get_genes_in_unitig(list(operation[1]))
            low_genes = self.get_genes_in_unitig(list(operation[0]))
            fw, rv, _, _ = self.compare_paths(low_genes, high_genes)
            # never delete a gene of interest (:1880-1884)
            if any(c[1][1:] in genesOfInterest and c[0][1:] not in genesOfInterest for c in fw):
                continue
            fw_alignments[operation], bw_alignments[operation] = fw, rv
            mers = [tuple(low_genes[i:i + k]) for i in range(len(low_genes) - (k - 1))]
            fw_counters[operation] = Counter(mers)
            bw_counters[operation] = Counter(tuple(self.reverse_list_of_genes(list(m))) for m in mers)
        for read_id, operation in reads_to_correct.items():
            if operation not in fw_alignments:
                continue
            genes_on_read = self._reads[read_id][:]
            read_alignment = self.reorient_alignment(
                self.get_gene_mer_strings(genes_on_read), fw_counters[operation], bw_counters[operation],
                fw_alignments[operation], bw_alignments[operation])
            if read_alignment is None:
                continue
            _, lower_mapping = self.get_path_to_alignment_mapping(read_alignment)
            low_on_alignment = [a[1] for a in read_alignment if not a[1] == "*"]
            _, (start_path, end_path), (first_shared, last_shared) = self.longest_common_sublist(
                low_on_alignment, genes_on_read)
            subset = read_alignment[lower_mapping[start_path]:lower_mapping[end_path] + 1]
            subset = self.modify_alignment_subset(subset, genes_on_read[first_shared:last_shared + 1])
            if len(subset) != 0:
                self.correct_genes_on_read(genes_on_read, first_shared, last_shared, subset, read_id)
                self.correct_gene_positions_on_read(first_shared, last_shared, subset, read_id, fastq_data)
        return path_coverages

    def correct_low_coverage_paths(self, fastq_data, genesOfInterest, cores, min_path_coverage,
                                   components_to_skip, use_minimizers=False):
        # :2196-2250
        assert self._genePositions
        starts = self.identify_potential_bubble_starts()
        max_distance = self._kmerSize * 4
        path_coverages = []
        for component in self.components():
            sys.stderr.write(f"\n\tAmira: popping bubbles using 1 CPU for component {component} / "
                             f"{len(self.components())}\n")
            if component in components_to_skip:
                continue
            if component not in starts:
                continue
            unique_paths = self.get_all_paths_between_junctions_in_component(starts[component], max_distance, cores)
            filtered = self.filter_paths_between_bubble_starts(unique_paths)
            ordered = sorted(filtered, key=lambda x: len(x[0]), reverse=False)
            path_minimizers = self.get_minhashes_for_paths(ordered, fastq_data, cores) if use_minimizers else None
            paired = self.separate_paths_by_terminal_nodes(ordered)
            path_coverages += self.correct_bubble_paths(paired, fastq_data, path_minimizers, genesOfInterest,
                                                        min_path_coverage)
        return self._reads, self._genePositions, path_coverages, min_path_coverage

    # ------------------------------------------------------------------ unitigs (row f4)
    def get_unitigs_in_graph(self, outfile):
        # :2961-2975
        unitigs = set()
        for node in self.all_nodes():
            if len(self.get_all_neighbors(node)) > 2:
                continue
            path = self.get_linear_path_for_node(node, True)
            path = sorted([path, list(reversed(path))])[0]
            genes = self.get_genes_in_unitig(path)
            canonical = sorted([genes, self.reverse_list_of_genes(genes)])[0]
            unitigs.add((tuple(canonical), len(self.collect_reads_in_path(path))))
        with open(outfile, "w") as fh:
            fh.write("\n".join([f"{','.join(u[0])}\t{u[1]}" for u in unitigs]))
