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
