"""Oracle drivers.  TEST INFRASTRUCTURE ONLY (oracle/README.md).

Restates the parts of amira/graph_utils.py that sit on the hot path:
build_graph (:12-14), the cleaning sweep of iterative_bubble_popping (:127-181,
without the MinHash bubble-popping tail),
get_overall_mean_node_coverages (:299-313) and choose_kmer_size (:258-296).
build_multiprocessed_graph is modelled at cores=1, which is the only way the
pipeline calls it; its multi-core merge is not a parity target.
"""
import statistics

from .graph import GeneMerGraph


def build_graph(read_dict, kmer_size, gene_positions=None):
    return GeneMerGraph(read_dict, kmer_size, gene_positions)


def build_multiprocessed_graph(annotatedReads, geneMer_size, cores, gene_positions=None):
    assert cores == 1, "oracle models the single-graph result only"
    graph = GeneMerGraph(dict(annotatedReads), geneMer_size,
                         dict(gene_positions) if gene_positions is not None else None)
    return graph


class FakeFastq(dict):
    """fastq stand-in: only len(fastq[read]["sequence"]) is consulted (construct_graph.py:1685)."""

    def __init__(self, lengths):
        super().__init__({r: {"sequence": range(n)} for r, n in lengths.items()})


def correction_sweep(reads, positions, k, fastq, node_min_coverage=3, trace=None):
    """build -> filter_graph(n,1) -> correct_reads -> build -> remove_short_linear_paths(k)
    -> correct_reads -> build   (graph_utils.py:145-166; SURVEY Appendix C cfg-3 sweep)."""

    def note(tag, g):
        if trace is not None:
            trace.append((tag, g))

    g1 = build_graph(reads, k, positions)
    note("build1", g1)
    g1.filter_graph(node_min_coverage, 1)
    note("filtered1", g1)
    reads2, pos2 = g1.correct_reads(fastq)
    g2 = build_graph(reads2, k, pos2)
    note("build2", g2)
    removed = g2.remove_short_linear_paths(k)
    note("clipped2", g2)
    reads3, pos3 = g2.correct_reads(fastq)
    g3 = build_graph(reads3, k, pos3)
    note("build3", g3)
    return g3, reads3, pos3, removed


def get_overall_mean_node_coverages(graph):
    # graph_utils.py:299-313
    out = {}
    for k in range(3, 16, 2):
        covs = [
            sum(1 for r in n.get_reads() if len(graph.get_reads()[r]) >= k)
            for n in graph.all_nodes()
        ]
        out[k] = statistics.mean(covs) if covs else 0
    return out


def choose_kmer_size(mean_cov, reads, cores, positions, genes_of_interest):
    # graph_utils.py:258-296
    chosen = 3
    if mean_cov >= 20:
        for k in range(3, 16, 2):
            graph = build_graph(dict(reads), k, dict(positions))
            amr = {n.__hash__() for g in genes_of_interest for n in graph.get_nodes_containing(g)}

            def valid(c):
                members = [n.__hash__() for n in graph.get_nodes_in_component(c)]
                rs = graph.collect_reads_in_path([h for h in members if h in amr])
                lens = [len(graph.get_reads()[r]) for r in rs]
                if not lens:
                    return True
                return len([x for x in lens if x >= 2 * k - 1]) / len(lens) >= 0.8

            if all(valid(c) for c in graph.components()):
                chosen = k
            else:
                break
    return chosen


def iterative_bubble_popping(new_annotatedReads, new_gene_position_dict, cleaning_iterations, geneMer_size, cores,
                             short_reads, short_read_gene_positions, fastq_content, output_dir, node_min_coverage,
                             sample_genesOfInterest, min_path_coverage):
    # graph_utils.py:127-181 — the whole cleaning driver, bubble popping included
    prev_nodes, components_to_skip = 0, set()
    for _ in range(cleaning_iterations):
        g = build_multiprocessed_graph(new_annotatedReads, geneMer_size, 1, new_gene_position_dict)
        g.filter_graph(node_min_coverage, 1)
        new_annotatedReads, new_gene_position_dict = g.correct_reads(fastq_content)
        g = build_multiprocessed_graph(new_annotatedReads, geneMer_size, 1, new_gene_position_dict)
        if len(g.get_nodes()) == prev_nodes:
            break
        prev_nodes = len(g.get_nodes())
        short_reads.update(g.get_short_read_annotations())
        short_read_gene_positions.update(g.get_short_read_gene_positions())
        g.remove_short_linear_paths(geneMer_size)
        new_annotatedReads, new_gene_position_dict = g.correct_reads(fastq_content)
        g = build_multiprocessed_graph(new_annotatedReads, geneMer_size, 1, new_gene_position_dict)
        short_reads.update(g.get_short_read_annotations())
        short_read_gene_positions.update(g.get_short_read_gene_positions())
        new_annotatedReads, new_gene_position_dict, _, min_path_coverage = g.correct_low_coverage_paths(
            fastq_content, sample_genesOfInterest, cores, min_path_coverage, components_to_skip, True)
    return new_annotatedReads, new_gene_position_dict
