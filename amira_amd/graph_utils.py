"""Graph drivers — drop-in for the hot-path part of amira/graph_utils.py (reference v0.11.0):
build_graph (:12-14), merge_nodes / merge_edges / merge_reads / merge_graphs (:17-102),
build_multiprocessed_graph (:105-124), the cleaning loop of
iterative_bubble_popping (:127-181), choose_kmer_size (:258-296) and
get_overall_mean_node_coverages (:299-313).

The reference's only parallelism is host processes over read shards whose merge is both slow
(CHANGELOG "limit graph building to 1 CPU") and wrong for edge coverages
(graph_utils.py:71-75); every pipeline call passes cores=1.  Here one graph is always built
on the device and `cores` is accepted for signature compatibility only.
"""
import statistics
import os
import sys

import numpy as np

from .construct_graph import GeneMerGraph


def build_graph(read_dict, kmer_size, gene_positions=None):
    return GeneMerGraph(read_dict, kmer_size, gene_positions)


def _tokenized(reads, gene_positions):
    """(TokenizedReads, TokenizedPositions or None) of a {read: genes} / {read: [(start, end)]} pair"""
    from .io import TokenizedPositions, TokenizedReads
    from .tokens import tokenize
    if isinstance(reads, TokenizedReads):
        reads_t = reads.settled()
    else:
        reads_t = TokenizedReads(*tokenize(reads))
    if gene_positions is None:
        return reads_t, gene_positions
    if isinstance(gene_positions, TokenizedPositions):
        return reads_t, gene_positions.settled()
    offs, n = reads_t.read_offsets, int(reads_t.read_offsets[-1])
    gs, ge = np.empty(n, np.int64), np.empty(n, np.int64)
    for r, rid in enumerate(reads_t.read_ids):
        a, b = int(offs[r]), int(offs[r + 1])
        if b > a:
            p = gene_positions[rid]
            gs[a:b] = [x[0] for x in p[: b - a]]
            ge[a:b] = [x[1] for x in p[: b - a]]
    return reads_t, TokenizedPositions(reads_t.read_ids, offs, gs, ge)


def _own(mapping):
    """the graph's own copy of a {read: ...} mapping, as the reference's drivers make one (.copy(), dict
    comprehensions); array-backed mappings (amira_amd.io.TokenizedReads / TokenizedPositions) keep their arrays —
    copying them item by item would decode a million reads into lists only to tokenise them again"""
    from .io import TokenizedPositions, TokenizedReads
    if mapping is None or isinstance(mapping, TokenizedReads):
        return mapping
    if isinstance(mapping, TokenizedPositions):
        return mapping.copy()   # shares the arrays; correct_reads redirects changed reads in the COPY, as it would
    return {r: mapping[r] for r in mapping}


def build_filtered_graph(read_dict, kmer_size, gene_positions, min_node_coverage, min_edge_coverage=1):
    """build_graph(...) followed by filter_graph(min_node_coverage, min_edge_coverage) — the opening of every
    cleaning iteration (graph_utils.py:147-149) — in one device pass: an uncorrected graph is ~99 % nodes the filter
    deletes at once, and they are never ranked, stored or joined by edges.  The graph a caller gets is the one the
    two calls leave behind."""
    return GeneMerGraph(read_dict, kmer_size, gene_positions, _filter=(min_node_coverage, min_edge_coverage))


_CORES_NOTE = []


def build_multiprocessed_graph(annotatedReads, geneMer_size, cores, gene_positions=None):
    """graph_utils.py:105-124.  The reference shards the reads over `cores` host processes and merges their sub-graphs;
    its own pipeline always passes cores = 1 (CHANGELOG: multi-process builds were slower), and that single-graph result
    is what comes back here, built on ONE GPU whatever `cores` says — said once on stderr when cores > 1.  The read-
    sharded build across GPUs is a collective of its own, one process per GPU with the shard's reads as token arrays:
    amira_amd.dist.dist_build / amg_dist_merge."""
    if cores is not None and int(cores) > 1 and not _CORES_NOTE:
        _CORES_NOTE.append(True)
        sys.stderr.write(f"\nAmira (amira_amd): build_multiprocessed_graph builds on one GPU; cores={cores} is not used "
                         "(read-sharded builds over several GPUs: amira_amd.dist.dist_build)\n")
    return build_graph(_own(annotatedReads), geneMer_size, _own(gene_positions))


def merge_graphs(sub_graphs):
    """graph_utils.py:94-102: one graph from the sub-graphs of a read-sharded build.  The reference replays every
    sub-graph's windows into the first one object by object (merge_nodes :17-50) and then patches edges in
    (merge_edges :53-76, which doubles the coverage of every shared edge instead of adding the other side's: SURVEY
    section 5).  Here the merged graph is BUILT: the sub-graphs' reads, in sub-graph order — the order in which the
    reference's merge meets them — go through one device build, which is the single-graph result (what the
    reference's own pipeline always uses: it passes cores = 1).  Across GPUs the same result comes from
    amira_amd.dist.dist_build (read shards + key-owner table merge over RCCL)."""
    first = sub_graphs[0]
    reads, positions = {}, ({} if first.get_gene_positions() is not None else None)
    for g in sub_graphs:
        for r in g.get_reads():
            reads[r] = g.get_reads()[r]
            if positions is not None and g.get_gene_positions() is not None:
                positions[r] = g.get_gene_positions()[r]
    return GeneMerGraph(reads, first.get_kmerSize(), positions)


def merge_nodes(sub_graphs, fastq_data=None):
    """graph_utils.py:17-50.  Returns the merged graph like the reference; it is complete already (edges, reads and
    component ids included: see merge_graphs), so merge_edges / merge_reads have nothing left to do."""
    return merge_graphs(sub_graphs)


def merge_edges(sub_graphs, reference_graph):
    """graph_utils.py:53-76: nothing to do on a graph merge_nodes / merge_graphs returned (edges are part of the build)"""
    return None


def merge_reads(sub_graphs, reference_graph):
    """graph_utils.py:78-91: nothing to do on a graph merge_nodes / merge_graphs returned (it was built from all reads)"""
    return None


def cleaning_sweep(reads, gene_positions, geneMer_size, fastq_content, node_min_coverage=3):
    """one cleaning iteration without the bubble-popping tail (graph_utils.py:145-166):
    build -> filter_graph(n, 1) -> correct_reads -> build -> remove_short_linear_paths(k)
    -> correct_reads -> build.  Returns (graph, reads, positions)."""
    graph = build_filtered_graph(_own(reads), geneMer_size, _own(gene_positions), node_min_coverage, 1)
    reads, gene_positions = graph.correct_reads(fastq_content)
    graph = build_multiprocessed_graph(reads, geneMer_size, 1, gene_positions)
    graph.remove_short_linear_paths(geneMer_size, _lazy_hashes=True)
    reads, gene_positions = graph.correct_reads(fastq_content)
    graph = build_multiprocessed_graph(reads, geneMer_size, 1, gene_positions)
    return graph, reads, gene_positions


def iterative_bubble_popping(new_annotatedReads, new_gene_position_dict, cleaning_iterations,
                             geneMer_size, cores, short_reads, short_read_gene_positions,
                             fastq_content, output_dir, node_min_coverage, sample_genesOfInterest,
                             min_path_coverage):
    """same loop as the reference (graph_utils.py:127-181), bubble popping included; the first build of an
    iteration and the filter_graph that follows it run as one device pass (build_filtered_graph)."""
    prev_nodes = 0
    components_to_skip = set()
    # dicts in, dicts out — and arrays in between: the reads and their positions are tokenised ONCE, every build of
    # every iteration takes them over as arrays (device to device after a correction), the reads bubble popping
    # rewrites are spelled into them (TokenizedReads / TokenizedPositions .settled()).  Callers that hand the array-
    # backed mappings over (amira_amd.pre_processing.process_pandora_json) get them back.
    from .io import TokenizedReads
    as_dicts = not isinstance(new_annotatedReads, TokenizedReads)
    if as_dicts and new_gene_position_dict and len(new_gene_position_dict) >= len(new_annotatedReads) \
            and not os.environ.get("AMG_BUBBLES_BY_OBJECTS"):
        try:
            new_annotatedReads, new_gene_position_dict = _tokenized(new_annotatedReads, new_gene_position_dict)
        except (KeyError, TypeError, ValueError, IndexError, AssertionError):
            pass   # (positions that do not cover the reads, ...: the dicts go through as they are)
    # (the cyclic collector is paused for the run: the steps below make and drop millions of small containers — gene
    # lists, position pairs, path tuples — none of them in cycles, and every pass of the collector over a heap that
    # holds the reads of a whole sample costs more than the step that triggered it)
    import gc
    gc_was_on = gc.isenabled()
    gc.disable()
    try:
        for this_iteration in range(cleaning_iterations):
            sys.stderr.write(f"\nAmira: running graph cleaning iteration {this_iteration+1}\n")
            graph = build_filtered_graph(_own(new_annotatedReads), geneMer_size, _own(new_gene_position_dict),
                                         node_min_coverage, 1)
            new_annotatedReads, new_gene_position_dict = graph.correct_reads(fastq_content)
            graph = build_multiprocessed_graph(new_annotatedReads, geneMer_size, 1, new_gene_position_dict)
            if graph.get_total_number_of_nodes() == prev_nodes:
                sys.stderr.write(f"\n\tAmira: terminating cleaning at iteration {this_iteration+1}\n")
                break
            prev_nodes = graph.get_total_number_of_nodes()
            sys.stderr.write("\n\tAmira: removing dead ends\n")
            short_reads.update(graph.get_short_read_annotations())
            short_read_gene_positions.update(graph.get_short_read_gene_positions())
            graph.remove_short_linear_paths(geneMer_size, _lazy_hashes=True)
            new_annotatedReads, new_gene_position_dict = graph.correct_reads(fastq_content)
            graph = build_multiprocessed_graph(new_annotatedReads, geneMer_size, 1, new_gene_position_dict)
            short_reads.update(graph.get_short_read_annotations())
            short_read_gene_positions.update(graph.get_short_read_gene_positions())
            new_annotatedReads, new_gene_position_dict, path_coverages, min_path_coverage = (
                graph.correct_low_coverage_paths(fastq_content, sample_genesOfInterest, cores,
                                                 min_path_coverage, components_to_skip, True))
    finally:
        if gc_was_on:
            gc.enable()
        from .bubble_popping import release_sequences
        release_sequences()   # (the reads' bases went to the device once for this run: not kept beyond it)
    if as_dicts and isinstance(new_annotatedReads, TokenizedReads):
        new_annotatedReads = new_annotatedReads.to_dict()
        new_gene_position_dict = (new_gene_position_dict.to_dict() if hasattr(new_gene_position_dict, "to_dict")
                                  else {r: new_gene_position_dict[r] for r in new_gene_position_dict})
    return new_annotatedReads, new_gene_position_dict


def choose_kmer_size(overall_mean_node_coverage, new_annotatedReads, cores, new_gene_position_dict,
                     sample_genesOfInterest):
    """largest odd k in 3..15 such that, in every component, >= 80 % of the reads through
    AMR nodes have >= 2k-1 genes (graph_utils.py:258-296)."""
    geneMer_size = 3
    if overall_mean_node_coverage >= 20:
        # the seven graphs see the same reads: gene names are hashed and ranked, reads and positions flattened and
        # uploaded ONCE, and the seven builds share two passes over the token stream (GeneMerGraph.build_many ->
        # amg_build_multi) instead of making two each
        reads_t, pos_t = _tokenized(new_annotatedReads, new_gene_position_dict)
        graphs = GeneMerGraph.build_many(reads_t, list(range(3, 16, 2)), pos_t)
        for k, graph in zip(range(3, 16, 2), graphs):
            amr = {n.__hash__() for g in sample_genesOfInterest for n in graph.get_nodes_containing(g)}

            def is_component_valid(component):
                members = [n.__hash__() for n in graph.get_nodes_in_component(component)]
                reads = graph.collect_reads_in_path([h for h in members if h in amr])
                lengths = [len(graph.get_reads()[r]) for r in reads]
                if len(lengths) == 0:
                    return True
                return len([x for x in lengths if x >= (2 * k - 1)]) / len(lengths) >= 0.8

            if all(is_component_valid(c) for c in graph.components()):
                geneMer_size = k
            else:
                break
        for graph in reversed(graphs):
            graph.close()
    return geneMer_size


def get_overall_mean_node_coverages(graph):
    """mean over nodes of the number of the node's reads with >= k genes, k = 3,5,..,15
    (graph_utils.py:299-313) — from the device's node->reads CSR."""
    off, idx = graph._engine.node_reads()
    alive = graph._engine.nodes()["alive"] != 0
    read_len = np.diff(graph._read_off)
    out = {}
    for k in range(3, 16, 2):
        ok = (read_len[idx] >= k).astype(np.int64)
        csum = np.concatenate([[0], np.cumsum(ok)])
        per_node = (csum[off[1:]] - csum[off[:-1]])[alive].tolist()
        out[k] = statistics.mean(per_node) if per_node else 0
    return out
