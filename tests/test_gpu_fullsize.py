"""Full-size checks on the GPU at BASELINE.json's bench configurations: exact comparison with
the sequential C token oracle (oracle/token_oracle.c, itself checked against the pinned Python
oracle) plus the size-independent invariants of SURVEY Appendix G."""
import os
import sys

import numpy as np
import pytest

import token_oracle

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def _invariants(eng, c, L, k):
    nodes, edges = eng.nodes(), eng.edges()
    assert int(nodes["coverage"].sum()) == c["n_windows"]                      # sum node cov == gene-mers
    reads_with_windows = c["n_reads"] - c["n_short_reads"]
    assert int(edges["coverage"].sum()) == 2 * (c["n_windows"] - reads_with_windows)
    loops = int((edges["src"] == edges["tgt"]).sum())
    assert c["n_edges"] == 2 * c["n_pairs"] - loops
    # first occurrences are strictly increasing in node id order (ids = insertion order)
    assert bool(np.all(np.diff(nodes["first_token"]) > 0))


@pytest.mark.parametrize("workload,k,exact", [("cfg2", None, 1), ("cfg3", None, 1), ("cfg3", 3, 1),
                                                  ("cfg2", 7, 0)])
def test_full_size_build_equals_c_oracle(workload, k, exact):
    """k = None: the configuration's own k (5: two-word exact keys); k = 3: one-word exact keys;
    k = 7 on the 5 000-gene vocabulary (98 bits): the verified-fingerprint path"""
    import bench
    from amira_amd import Engine
    w = dict(bench.WORKLOADS[workload])
    if k is not None:
        w["k"] = k
    vocab, toks, offs = bench.make_tokens(w, 0, w["N"])
    eng = Engine(0)
    eng.set_reads(toks, offs, vocab.two_v)
    eng.build(w["k"])
    c = eng.counts()
    assert c["exact_keys"] == exact
    _invariants(eng, c, w["L"], w["k"])
    want = token_oracle.build(toks, offs, w["k"], vocab.two_v)
    nodes, edges = eng.nodes(), eng.edges()
    tok_node, tok_dir = eng.read_nodes()
    assert c["n_windows"] == want["n_windows"] and c["n_short_reads"] == want["n_short"]
    assert np.array_equal(nodes["tokens"], want["tokens"])
    assert np.array_equal(nodes["coverage"], want["coverage"])
    assert np.array_equal(nodes["first_dir"], want["first_dir"])
    for a, b in (("src", "src"), ("tgt", "tgt"), ("sdir", "sdir"), ("tdir", "tdir"), ("coverage", "ecov")):
        assert np.array_equal(edges[a], want[b]), a
    assert np.array_equal(tok_node, want["tok_node"]) and np.array_equal(tok_dir, want["tok_dir"])
    # idempotence of the sweep's fixed point: rebuilding the same reads gives the same graph
    eng.build(w["k"])
    assert np.array_equal(eng.nodes()["coverage"], nodes["coverage"])
    eng.close()


def test_full_size_sweep_properties():
    """cfg3 sweep at full size: every corrected read threads through live nodes only, the
    corrected read set is a fixed point of a second filter+correct, counts are conserved."""
    import bench
    from amira_amd import Engine
    w = bench.WORKLOADS["cfg3-sweep"]
    N, L, k = 200_000, w["L"], w["k"]   # 200 k reads keep the C-side checks below in seconds
    vocab, toks, offs = bench.make_tokens(w, 0, N)
    gs = np.tile(np.arange(L, dtype=np.int64) * 1000, N)
    eng = Engine(0)
    eng.set_reads(toks, offs, vocab.two_v)
    eng.set_positions(gs, gs + 899, np.full(N, L * 1000 + 100, np.int64))
    eng.build(k)
    eng.filter(3, 1)
    marked = int(eng.reads_to_correct().sum())
    n_reads, n_tokens = eng.correct_reads()
    out = eng.corrected(n_reads, n_tokens, True)
    assert n_reads <= N and marked > 0
    lens = np.diff(out["read_offsets"])
    assert lens.min() >= k                                   # every kept read still has a node
    assert np.all(out["gene_start"] <= out["gene_end"])      # repaired positions are ordered
    unchanged = out["changed"] == 0
    src = out["orig_read"][unchanged]
    assert np.array_equal(lens[unchanged], np.diff(offs)[src])
    eng.adopt_corrected()
    eng.build(k)
    c2 = eng.counts()
    _invariants(eng, c2, L, k)
    # the rebuilt graph equals the C oracle on the corrected reads
    want = token_oracle.build(out["tokens"], out["read_offsets"], k, vocab.two_v)
    assert np.array_equal(eng.nodes()["coverage"], want["coverage"])
    assert np.array_equal(eng.read_nodes()[0], want["tok_node"])
    eng.close()


def _cfg3_inputs(N):
    import bench
    w = bench.WORKLOADS["cfg3-sweep"]
    vocab, toks, offs = bench.make_tokens(w, 0, N)
    L = w["L"]
    gs = np.tile(np.arange(L, dtype=np.int64) * 1000, N)
    return w, vocab, toks, offs, gs, gs + 899, np.full(N, L * 1000 + 100, np.int64)


def test_full_size_sweep_equals_c_oracle():
    """THE benchmarked workload, bit for bit: cfg3 sweep at 1 M reads x 60 genes (build ->
    filter_graph(3,1) -> correct_reads -> build -> remove_short_linear_paths(5) -> correct_reads
    -> build) against the sequential C restatement of the reference (oracle/token_sweep.c, pinned
    to the Python oracle in tests/test_token_oracle.py): every graph array after every stage,
    masked windows, reads to correct, corrected genes and positions, removed tips."""
    from amira_amd import Engine
    from helpers import compare_corrected, compare_engine_to_sweep
    N = 1_000_000
    w, vocab, toks, offs, gs, ge, rl = _cfg3_inputs(N)
    k = w["k"]
    eng = Engine(0)
    orc = token_oracle.Sweep(toks, offs, vocab.two_v, gs, ge, rl)
    try:
        eng.set_reads(toks, offs, vocab.two_v)
        eng.set_positions(gs, ge, rl)
        eng.build(k)
        orc.build(k)
        compare_engine_to_sweep(eng, orc, "build 1")
        eng.filter(3, 1)
        orc.filter(3, 1)
        compare_engine_to_sweep(eng, orc, "filter")
        out = compare_corrected(eng, orc, True, "correct 1")
        assert int(out["changed"].sum()) > N // 2          # most reads carry an error at 2 % x 60 genes
        eng.adopt_corrected()
        orc.adopt_corrected()
        eng.build(k)
        orc.build(k)
        compare_engine_to_sweep(eng, orc, "build 2")
        got = np.sort(eng.remove_short_linear_paths(k))
        want = np.sort(orc.remove_short_linear_paths(k))
        assert len(want) > 0 and np.array_equal(got, want)
        compare_engine_to_sweep(eng, orc, "clip")
        compare_corrected(eng, orc, True, "correct 2")
        eng.adopt_corrected()
        orc.adopt_corrected()
        eng.build(k)
        orc.build(k)
        compare_engine_to_sweep(eng, orc, "build 3")
        _invariants(eng, eng.counts(), w["L"], k)
    finally:
        eng.close()
        orc.close()
