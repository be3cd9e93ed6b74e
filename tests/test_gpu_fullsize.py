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
                                                  ("cfg2", 7, 2), ("cfg3", 7, 2), ("cfg3-onecounter", None, 1)])
def test_full_size_build_equals_c_oracle(workload, k, exact, monkeypatch):
    """k = None: the configuration's own k (5: two-word exact keys); k = 3: one-word exact keys;
    k = 7 (98 / 112 bits): the 16-byte slots keyed by verified 94-bit fingerprints"""
    import bench
    from amira_amd import Engine
    if workload.endswith("-onecounter"):
        # the round-3 shape of the table passes: claims from ONE counter, every edge class through the table — and the
        # per-workgroup lists of the counting sweeps cut to 8 ids, so that they run over and the second launch of a
        # count falls back to a sweep of its own (k_count_ids)
        monkeypatch.setenv("AMG_CLAIM_SHARDS", "0")
        monkeypatch.setenv("AMG_EDGE_LONE", "0")
        monkeypatch.setenv("AMG_COUNT_LIST_SEG", "8")
        workload = workload[:-11]
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


def test_cfg5_emulated_8_ranks_equal_c_oracle():
    """BASELINE config 5 as far as one GPU can take it: the 8 M-read stream cut into 8 shards of
    1 M reads, one Engine per emulated rank on this GPU, every build merged by key owner through the
    loop-back exchange (the device phases bench.py --gpus 8 runs; only the wire is missing), the
    first one with filter_graph(3,1) fused in — against the sequential C restatement of the
    reference run on the WHOLE 8 M-read stream.  Every rank must hold the single-graph result."""
    import bench
    from amira_amd import Engine
    from amira_amd.dist import dist_build_loopback
    from helpers import compare_engine_to_sweep, live_arrays
    import time
    t_start = time.time()
    lap = (lambda what: print(f"[cfg5] {time.time() - t_start:7.1f} s  {what}", flush=True)) if os.environ.get("AMG_TEST_TIMES") else (lambda what: None)
    world, N = 8, 1_000_000
    w = bench.WORKLOADS["cfg3-sweep"]          # bench.py: rank r holds reads [r N, (r + 1) N) of this stream
    L, k = w["L"], w["k"]
    vocab, toks, offs = bench.make_tokens(w, 0, world * N)
    lap("reads made")
    gs = np.tile(np.arange(L, dtype=np.int64) * 1000, world * N)
    ge = gs + 899
    rl = np.full(world * N, L * 1000 + 100, np.int64)
    orc = token_oracle.Sweep(toks, offs, vocab.two_v, gs, ge, rl)
    lap("oracle loaded")
    engines = []
    try:
        for r in range(world):
            lo, hi = r * N, (r + 1) * N
            e = Engine(0)
            e.set_reads(toks[offs[lo]:offs[hi]], offs[lo:hi + 1] - offs[lo], vocab.two_v)
            e.set_positions(gs[offs[lo]:offs[hi]], ge[offs[lo]:offs[hi]], rl[lo:hi])
            engines.append(e)
        del gs, ge

        # ---- the whole stream through ONE engine's plain build (480 M tokens, 43 M nodes: claim ids from the shard
        # counters, the edge classes of coverage-1 nodes past the edge table, at eight times the benchmark's size)
        lap("engines loaded")
        orc.build(k)
        lap("oracle build 1")
        one = Engine(0)
        try:
            one.set_reads(toks, offs, vocab.two_v)
            one.build(k)
            assert one.counts()["exact_keys"] == 1
            compare_engine_to_sweep(one, orc, "one engine, 8 M reads")
        finally:
            one.close()
        lap("one engine compared")
        # ---- build 1, merged, filter_graph(3, 1) fused in == build + filter on the whole stream
        dist_build_loopback(engines, k, 3, 1)
        orc.filter(3, 1)
        want = live_arrays(orc)
        tok_lo = 0
        for r, e in enumerate(engines):
            got = live_arrays(e, with_adj=(r in (0, world - 1)))
            n_tok = len(got["tok_node"])
            for key in ("tokens", "coverage", "first_dir", "src", "tgt", "sdir", "tdir", "ecov"):
                assert np.array_equal(got[key], want[key]), (r, key)
            if "adj" in got:
                assert np.array_equal(got["adj_off"], want["adj_off"]) and np.array_equal(got["adj"], want["adj"]), r
            assert np.array_equal(got["tok_node"], want["tok_node"][tok_lo:tok_lo + n_tok]), r
            assert np.array_equal(got["tok_dir"], want["tok_dir"][tok_lo:tok_lo + n_tok]), r
            assert np.array_equal(got["to_correct"], want["to_correct"][r * N:(r + 1) * N]), r
            assert e.counts()["n_nodes"] == len(want["coverage"])       # only survivors were replicated
            tok_lo += n_tok
        del want

        def correct_all(stage):
            """correct_reads on every rank; the ranks' corrected reads, concatenated, equal the oracle's"""
            nr, nt = orc.correct_reads(threads=min(32, os.cpu_count() or 1))
            ref = orc.corrected(nr, nt, True)
            outs = [e.corrected(*e.correct_reads(), True) for e in engines]
            for key in ("tokens", "gene_start", "gene_end", "changed"):
                assert np.array_equal(np.concatenate([o[key] for o in outs]), ref[key]), (stage, key)
            assert np.array_equal(np.concatenate([np.diff(o["read_offsets"]) for o in outs]), np.diff(ref["read_offsets"]))
            base = np.cumsum([0] + [e.sizes()[0] for e in engines])
            assert np.array_equal(np.concatenate([o["orig_read"] + base[r] for r, o in enumerate(outs)]), ref["orig_read"])
            for e in engines:
                e.adopt_corrected()
            orc.adopt_corrected()

        def merged_build_equals(stage):
            dist_build_loopback(engines, k)
            orc.build(k)
            tn_o, td_o = orc.read_nodes()
            tok_lo = 0
            for r, e in enumerate(engines):
                n_tok = e.sizes()[1]
                if r in (0, world - 1):
                    # the whole replicated graph: compare through a view of the oracle cut to this shard
                    ne, no = e.nodes(), orc.nodes()
                    for key in ("tokens", "coverage", "first_dir", "component", "alive"):
                        assert np.array_equal(ne[key], no[key]), (stage, r, key)
                    ee, eo = e.edges(), orc.edges()
                    for key in ("src", "tgt", "sdir", "tdir", "coverage", "alive"):
                        assert np.array_equal(ee[key], eo[key]), (stage, r, key)
                    a, b = e.node_adj(), orc.node_adj()
                    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]), (stage, r)
                c, co = e.counts(), orc.counts()
                for key in ("n_nodes", "n_edges", "n_components"):
                    assert c[key] == co[key], (stage, r, key)
                tn, td = e.read_nodes()
                assert np.array_equal(tn, tn_o[tok_lo:tok_lo + n_tok]), (stage, r)
                assert np.array_equal(td, td_o[tok_lo:tok_lo + n_tok]), (stage, r)
                tok_lo += n_tok
            assert tok_lo == len(tn_o)

        lap("merged build 1 compared")
        correct_all("correct 1")
        lap("correct 1")
        merged_build_equals("build 2")
        lap("build 2")
        want_removed = np.sort(orc.remove_short_linear_paths(k))
        assert len(want_removed) > 0
        for e in engines:
            assert np.array_equal(np.sort(e.remove_short_linear_paths(k)), want_removed)
        lap("clip")
        correct_all("correct 2")
        lap("correct 2")
        merged_build_equals("build 3")
        lap("build 3")
    finally:
        for e in engines:
            e.close()
        orc.close()


def test_cfg4_full_size_build_and_planted_alleles():
    """BASELINE config 4 at full size (1 M error-free reads, 10 AMR genes planted in 2 - 3 genome
    contexts each): the device build equals the sequential C oracle, and the read-path clustering
    through the reference-shaped API finds exactly the planted alleles, each path's read set equal to a
    direct token-space computation (reads that hold the path exactly once, forward occurrences first —
    construct_graph.py:2401-2439 — found by sliding comparisons over the raw token array)."""
    import re
    import bench
    from amira_amd import Engine, GeneMerGraph, synth
    from amira_amd.io import TokenizedPositions, TokenizedReads
    w = bench.WORKLOADS["cfg4"]
    N, L, k = w["N"], w["L"], w["k"]
    vocab, toks, offs = bench.make_tokens(w, 0, N)
    # ---- build == C oracle
    eng = Engine(0)
    eng.set_reads(toks, offs, vocab.two_v)
    eng.build(k)
    want = token_oracle.build(toks, offs, k, vocab.two_v)
    nodes, edges = eng.nodes(), eng.edges()
    assert np.array_equal(nodes["tokens"], want["tokens"]) and np.array_equal(nodes["coverage"], want["coverage"])
    for a, b in (("src", "src"), ("tgt", "tgt"), ("sdir", "sdir"), ("tdir", "tdir"), ("coverage", "ecov")):
        assert np.array_equal(edges[a], want[b]), a
    assert np.array_equal(eng.read_nodes()[0], want["tok_node"])
    eng.close()
    # ---- clustering
    read_ids = synth.read_names(0, N)
    gs = np.tile(np.arange(L, dtype=np.int64) * 1000, N)
    g = GeneMerGraph(TokenizedReads(vocab, toks, offs, read_ids), k, TokenizedPositions(read_ids, offs, gs, gs + 899))
    genes = [f"amr{j}" for j in range(w["n_amr"])]
    clustered, path_reads = g.assign_reads_to_genes(genes, 1, {}, None)
    alleles = {gene: d for comp in clustered.values() for gene, d in comp.items()}
    assert [len(alleles[f"amr{j}"]) for j in range(10)] == [2 + (j % 2) for j in range(10)]   # the planted copies

    mat = toks.reshape(N, L)

    def occurrences(pattern):
        """(read, count) of the token pattern, by direct comparison of shifted columns"""
        m = len(pattern)
        hit = np.ones((N, L - m + 1), bool)
        for j, t in enumerate(pattern):
            hit &= mat[:, j:L - m + 1 + j] == t
        return hit.sum(axis=1)

    flip = vocab.two_v - 1
    checked = 0
    for named, reads in path_reads.items():
        genes_on_path = [re.sub(r"^([+-]amr\d+)_\d+$", r"\1", x) for x in named]
        fwd = [vocab.token(x) for x in genes_on_path]
        rev = [flip - t for t in reversed(fwd)]
        n_fw, n_rv = occurrences(fwd), occurrences(rev)
        holds = (n_fw == 1) | ((n_fw == 0) & (n_rv == 1))
        assert set(np.flatnonzero(holds).tolist()) == {int(r[1:]) for r in reads}, named
        checked += 1
    assert checked >= 25
    for gene, d in alleles.items():
        seen = set()
        token_p, token_m = vocab.token("+" + gene), vocab.token("-" + gene)
        carriers = set(np.flatnonzero(((mat == token_p) | (mat == token_m)).any(axis=1)).tolist())
        for allele, entries in d.items():
            rows = {int(e.split("_")[0][1:]) for e in entries}
            assert rows <= carriers and not (rows & seen), (gene, allele)   # alleles of a gene share no read
            assert 2000 < len(rows) < 4000                                    # ~3 000x depth per copy
            seen |= rows
    g.close()


def test_full_size_filtered_build_equals_c_oracle():
    """amg_build_filtered at the benchmark size: the first build of the cfg 3 sweep with filter_graph(3,1) applied on
    the way must leave what the C oracle's build + filter leave (live nodes, edges, coverages, list orders, masked
    windows, reads to correct) and the first correction must come out the same, genes and positions."""
    from amira_amd import Engine
    from helpers import compare_corrected, live_arrays
    N = 1_000_000
    w, vocab, toks, offs, gs, ge, rl = _cfg3_inputs(N)
    k = w["k"]
    eng = Engine(0)
    orc = token_oracle.Sweep(toks, offs, vocab.two_v, gs, ge, rl)
    try:
        eng.set_reads(toks, offs, vocab.two_v)
        eng.set_positions(gs, ge, rl)
        eng.build_filtered(k, 3, 1)
        orc.build(k)
        orc.filter(3, 1)
        got, want = live_arrays(eng), live_arrays(orc)
        for key in want:
            assert np.array_equal(got[key], want[key]), key
        assert eng.counts()["n_nodes"] == len(want["coverage"])      # only the survivors exist
        out = compare_corrected(eng, orc, True, "correct 1 after the filtered build")
        assert int(out["changed"].sum()) > N // 2
    finally:
        eng.close()
        orc.close()
