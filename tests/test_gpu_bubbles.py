"""The device's share of bubble popping (row f1; amira_amd/csrc/amg_bubbles.hip) against the reference-shaped methods
of the same graph, which the goldens and the differential fuzzer pin to the reference and the oracle:
  * amg_junction_paths — one search per start junction — finds what get_all_paths_between_junctions_in_component's
    search per (start, stop) pair finds, in the order it adds them to its set (construct_graph.py:2066-2098);
  * amg_path_sketch_overlaps — sketches hashed from the resident bases, united and compared on the device — gives the
    sizes and overlaps of the sets the objects' way builds (:2148-2194, :1747-1786);
  * correct_low_coverage_paths is the same with and without AMG_BUBBLES_BY_OBJECTS."""
import numpy as np
import pytest

import procedures as P

pytestmark = pytest.mark.gpu

CASES = [(71, 300, 30, 120, 3, 0.04), (72, 400, 25, 90, 3, 0.05), (73, 300, 40, 150, 5, 0.04), (74, 200, 30, 60, 4, 0.05),
         (75, 500, 30, 200, 3, 0.03)]


def _graph(seed, N, L, V, k, err, min_cov=3):
    from amira_amd import GeneMerGraph, synth
    ids, sts = synth.loop_reads(seed, N, L, V, err, 0)
    calls = synth.to_read_dict(ids, sts, synth.gene_names(V, 0))
    pos = {r: [(80 * i, 80 * i + 59) for i in range(len(g))] for r, g in calls.items()}
    fq = P.synth_fastq(calls, pos, flank=40)
    try:
        g = GeneMerGraph(calls, k, pos)
        g.filter_graph(min_cov, 1)
        calls, pos = g.correct_reads(fq)
        return GeneMerGraph(calls, k, pos), fq
    except AssertionError:   # a palindromic gene-mer (even k)
        return None, None


@pytest.mark.parametrize("case", CASES)
def test_paths_between_junctions_one_search_per_start(case):
    g, _ = _graph(*case)
    if g is None:
        pytest.skip("palindromic gene-mer")
    found = g._junction_paths_on_device()
    assert found is not None
    components, paths_of = found
    assert components == g.components()
    starts = g.identify_potential_bubble_starts()
    assert sorted(paths_of) == sorted(starts)
    n_paths = 0
    for component, junctions in starts.items():
        want = g.get_all_paths_between_junctions_in_component(junctions, g.get_kmerSize() * 4, 1)
        got = set()
        for p in paths_of[component]:
            got.add(tuple(sorted([p, [(h, -d) for h, d in reversed(p)]])[0]))
        # the same paths, put into the set in the same order
        assert list(got) == want
        n_paths += len(want)
    assert n_paths > 0 or not starts


@pytest.mark.parametrize("case", CASES[:3])
def test_sketch_sizes_and_overlaps(case):
    g, fq = _graph(*case)
    if g is None:
        pytest.skip("palindromic gene-mer")
    starts = g.identify_potential_bubble_starts()
    compared = 0
    for component, junctions in starts.items():
        unique = g.get_all_paths_between_junctions_in_component(junctions, g.get_kmerSize() * 4, 1)
        shortest_first = sorted(g.filter_paths_between_bubble_starts(unique), key=lambda e: len(e[0]))
        bubbles = g.separate_paths_by_terminal_nodes(shortest_first)
        dev = g._path_overlaps_on_device(bubbles, fq)
        assert dev is not None
        sketches = g.get_minhashes_for_paths(shortest_first, fq, 1)
        for entries in bubbles.values():
            if len(entries) < 2:
                continue
            ranked = sorted(list(entries), key=lambda e: e[1], reverse=True)
            for i in range(len(ranked)):
                for j in range(i + 1, len(ranked)):
                    high, low = [n[0] for n in ranked[i][0]], [n[0] for n in ranked[j][0]]
                    a = g.get_minimizers_from_minhashes(high, sketches)
                    b = g.get_minimizers_from_minhashes(low, sketches)
                    assert dev.compare(high, low) == (len(a), len(b), len(a & b))
                    compared += 1
    assert compared > 0 or not starts


def test_sequences_in_another_order_than_the_reads():
    """the reads of a graph are a subset of fastq_data in an order of their own: row_to_seq"""
    g, fq = _graph(*CASES[0])
    if g is None:
        pytest.skip("palindromic gene-mer")
    shuffled = {r: fq[r] for r in sorted(fq, reverse=True)}
    shuffled["a read no graph has seen"] = {"sequence": "ACGT" * 50, "quality": "I" * 200}
    starts = g.identify_potential_bubble_starts()
    for component, junctions in starts.items():
        unique = g.get_all_paths_between_junctions_in_component(junctions, g.get_kmerSize() * 4, 1)
        shortest_first = sorted(g.filter_paths_between_bubble_starts(unique), key=lambda e: len(e[0]))
        bubbles = g.separate_paths_by_terminal_nodes(shortest_first)
        one, two = g._path_overlaps_on_device(bubbles, fq), g._path_overlaps_on_device(bubbles, shuffled)
        assert one is not None and two is not None
        assert one._size == two._size and one._common_of == two._common_of


@pytest.mark.parametrize("case", CASES + [(91, 2000, 60, 800, 5, 0.02)])   # (the last one: hundreds of junctions)
def test_whole_step_equals_the_objects_way(case, monkeypatch):
    def run(by_objects):
        if by_objects:
            monkeypatch.setenv("AMG_BUBBLES_BY_OBJECTS", "1")
        else:
            monkeypatch.delenv("AMG_BUBBLES_BY_OBJECTS", raising=False)
        g, fq = _graph(*case)
        if g is None:
            return None
        reads, pos, covs, _ = g.correct_low_coverage_paths(fq, set(), 1, 2, set(), True)
        return ({r: list(v) for r, v in reads.items()}, {r: [tuple(p) for p in v] for r, v in pos.items()},
                [float(c) for c in covs])
    a, b = run(False), run(True)
    if a is None:
        pytest.skip("palindromic gene-mer")
    assert a[2] == b[2]
    assert a[0] == b[0]
    assert a[1] == b[1]


def test_raw_calls_on_a_graph_without_junctions():
    from amira_amd import Engine, tokenize
    reads = {"r%d" % i: ["+g%d" % j for j in range(10)] for i in range(4)}
    vocab, toks, offs, _ = tokenize(reads)
    e = Engine(0)
    try:
        e.set_reads(toks, offs, vocab.two_v)
        e.build(3)
        found = e.junction_paths(12)
        assert len(found["junction_node"]) == 0 and len(found["path_start"]) == 0 and found["flags"] == 0
        assert found["path_off"].tolist() == [0]
    finally:
        e.close()


def _cleaning_inputs(seed, N, L, V, err):
    from amira_amd import synth
    ids, sts = synth.loop_reads(seed, N, L, V, err, 0)
    calls = synth.to_read_dict(ids, sts, synth.gene_names(V, 0))
    pos = {r: [(80 * i, 80 * i + 59) for i in range(len(g))] for r, g in calls.items()}
    return calls, pos, P.synth_fastq(calls, pos, flank=40)


@pytest.mark.parametrize("case", [(81, 400, 30, 150, 3, 0.04), (82, 600, 40, 300, 5, 0.03)])
def test_cleaning_run_from_arrays_from_dicts_and_by_objects(case, monkeypatch, tmp_path):
    """iterative_bubble_popping (graph_utils.py:127-181): array-backed mappings in and out, dicts in and out (tokenised
    once inside), and the reference-shaped way object by object give the same reads and positions"""
    from amira_amd import graph_utils as gu
    seed, N, L, V, k, err = case
    calls, pos, fq = _cleaning_inputs(seed, N, L, V, err)

    def run(reads, positions):
        short, short_pos = {}, {}
        r, p = gu.iterative_bubble_popping(reads, positions, 3, k, 1, short, short_pos, fq, str(tmp_path), 3, set(), 2)
        return {x: list(r[x]) for x in r}, {x: [tuple(q) for q in p[x]] for x in p}, sorted(short)

    copy = lambda: ({r: list(v) for r, v in calls.items()}, {r: list(v) for r, v in pos.items()})  # noqa: E731
    from_dicts = run(*copy())
    from_arrays = run(*gu._tokenized(*copy()))
    monkeypatch.setenv("AMG_BUBBLES_BY_OBJECTS", "1")
    by_objects = run(*copy())
    assert from_arrays == from_dicts
    assert by_objects == from_dicts
    assert sum(len(v) for v in from_dicts[0].values()) > 0


def test_sketch_calls_too_big_for_one_go_are_split(monkeypatch):
    """amg_path_sketch_overlaps refuses more (path, hash) pairs than its buffers are meant for (AMG_E_NOMEM); the
    caller then compares the groups of paths in halves — same sizes, same overlaps"""
    g, fq = _graph(*CASES[4])
    if g is None:
        pytest.skip("palindromic gene-mer")
    calls = []
    inner = g._engine.path_sketch_overlaps
    monkeypatch.setattr(g._engine, "path_sketch_overlaps", lambda *a, **k: (calls.append(1), inner(*a, **k))[1])
    starts = g.identify_potential_bubble_starts()
    split = 0
    for component, junctions in starts.items():
        unique = g.get_all_paths_between_junctions_in_component(junctions, g.get_kmerSize() * 4, 1)
        shortest_first = sorted(g.filter_paths_between_bubble_starts(unique), key=lambda e: len(e[0]))
        bubbles = g.separate_paths_by_terminal_nodes(shortest_first)
        monkeypatch.delenv("AMG_TEST_SKETCH_PAIRS", raising=False)
        whole = g._path_overlaps_on_device(bubbles, fq)
        assert whole is not None
        if sum(1 for e in bubbles.values() if len(e) > 1) < 2:
            continue
        cap = 64 * sum(whole._size) + 64
        while cap > 16:    # ever smaller buffers, until the one call no longer fits and halves of it do
            monkeypatch.setenv("AMG_TEST_SKETCH_PAIRS", str(cap))
            del calls[:]
            halves = g._path_overlaps_on_device(bubbles, fq)
            if halves is None:
                break      # (one group alone is beyond this cap)
            assert halves._size == whole._size and halves._common_of == whole._common_of
            if len(calls) > 1:
                split += 1
                break
            cap //= 2
    monkeypatch.delenv("AMG_TEST_SKETCH_PAIRS", raising=False)
    assert split > 0 or not starts


def test_long_sequences_cross_the_upload_buffers():
    """amg_seqs_create sends the bases through two 32 MB pinned buffers that take turns: segments that straddle the
    seams of the stream (and lie far into it) sketch as the same bases do on the host"""
    from amira_amd import Engine, tokenize
    from amira_amd.engine import Sequences
    rng = np.random.default_rng(7)
    seam = 32 << 20
    lengths = [seam + 4000, seam - 2500, 2 * seam + 77]            # seams fall inside read 0, read 1 | 2 and read 2
    seqs = [np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, n, dtype=np.uint8)].tobytes().decode() for n in lengths]
    genes = ["+g%d" % j for j in range(8)]
    reads = {"r%d" % i: list(genes) for i in range(3)}
    vocab, toks, offs, ids = tokenize(reads)
    k = 3
    # where the genes sit on each read: right across the seams of the concatenated stream
    stream_start = np.concatenate([[0], np.cumsum(lengths)[:-1]])
    around = [seam - 300, 2 * seam - 300 - int(stream_start[1]), 3 * seam - 300 - int(stream_start[2])]
    gs = np.concatenate([np.arange(8, dtype=np.int64) * 80 + a for a in around])
    ge = gs + 59
    e = Engine(0)
    s = Sequences(seqs, 0)
    try:
        e.set_reads(toks, offs, vocab.two_v)
        e.set_positions(gs, ge, np.asarray(lengths, np.int64))
        e.build(k)
        D = e.graph_sizes()[0]
        tok_node = e.read_node_ids()
        size, common = e.path_sketch_overlaps(s, None, 11, 1, np.arange(D + 1, dtype=np.int64), np.arange(D, dtype=np.int32),
                                              [0], [D - 1])
        want = []
        for node in range(D):
            segs = []
            for w in np.flatnonzero(tok_node == node).tolist():
                r = int(np.searchsorted(offs, w, side="right") - 1)
                segs.append(seqs[r][int(gs[w]):int(ge[w + k - 1]) + 1])
            want.append(e.minhash(segs, [0] * len(segs), 11, 1)[0])
        assert size.tolist() == [len(x) for x in want] and min(size.tolist()) > 100
        assert common.tolist() == [len(want[0] & want[D - 1])]
    finally:
        s.close()
        e.close()
