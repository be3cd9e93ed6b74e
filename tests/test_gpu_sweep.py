"""GPU parity of filter -> correct_reads -> rebuild -> clip -> correct_reads -> rebuild
(the cleaning sweep of graph_utils.py:145-166) against the CPU oracle, at the array level
of the C ABI."""
import numpy as np
import pytest

import procedures as P
from helpers import compare_engine_to_oracle, oracle_arrays

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from amira_amd import Engine
    e = Engine(0)
    yield e
    e.close()


def flat_positions(read_ids, reads, pos):
    gs = np.fromiter((p[0] for r in read_ids for p in pos[r]), dtype=np.int64)
    ge = np.fromiter((p[1] for r in read_ids for p in pos[r]), dtype=np.int64)
    assert len(gs) == sum(len(reads[r]) for r in read_ids)
    return gs, ge


def check_corrected(eng, vocab, read_ids, want_genes, want_pos):
    n_reads, n_tokens = eng.correct_reads()
    out = eng.corrected(n_reads, n_tokens, want_pos is not None)
    got_ids = [read_ids[i] for i in out["orig_read"]]
    assert got_ids == list(want_genes.keys())
    offs = out["read_offsets"]
    for i, rid in enumerate(got_ids):
        a, b = int(offs[i]), int(offs[i + 1])
        assert vocab.decode(out["tokens"][a:b]) == list(want_genes[rid]), rid
        if want_pos is not None:
            got = list(zip(out["gene_start"][a:b].tolist(), out["gene_end"][a:b].tolist()))
            assert got == [tuple(p) for p in want_pos[rid]], rid
    return got_ids, out


DERIVED = []   # per run_sweep: was the third graph made from the second one's live part (amg_derive.hip)?


def run_sweep(eng, reads, pos, fq, k, min_cov=3):
    from amira_amd import tokenize
    from amira_oracle import GeneMerGraph
    vocab, toks, offs, read_ids = tokenize(reads)
    eng.set_reads(toks, offs, vocab.two_v)
    gs, ge = flat_positions(read_ids, reads, pos)
    rl = np.asarray([len(fq[r]["sequence"]) for r in read_ids], dtype=np.int64)
    eng.set_positions(gs, ge, rl)
    pos = {r: list(v) for r, v in pos.items()}

    eng.build(k)
    g1 = GeneMerGraph(reads, k, pos)
    compare_engine_to_oracle(eng, oracle_arrays(g1, vocab, read_ids, offs, k))
    eng.filter(min_cov, 1)
    g1.filter_graph(min_cov, 1)
    compare_engine_to_oracle(eng, oracle_arrays(g1, vocab, read_ids, offs, k), live_only=True)
    r2, p2 = g1.correct_reads(fq)
    ids2, out2 = check_corrected(eng, vocab, read_ids, r2, p2)

    eng.adopt_corrected()
    eng.build(k)
    g2 = GeneMerGraph(r2, k, p2)
    compare_engine_to_oracle(eng, oracle_arrays(g2, vocab, ids2, out2["read_offsets"], k))
    removed = eng.remove_short_linear_paths(k)
    order = {h: i for i, h in enumerate(g2.get_nodes())}
    want_removed = sorted(order[h] for h in g2.remove_short_linear_paths(k))
    assert sorted(removed.tolist()) == want_removed
    compare_engine_to_oracle(eng, oracle_arrays(g2, vocab, ids2, out2["read_offsets"], k), live_only=True)
    r3, p3 = g2.correct_reads(fq)
    ids3, out3 = check_corrected(eng, vocab, ids2, r3, p3)

    eng.adopt_corrected()
    eng.build(k)
    g3 = GeneMerGraph(r3, k, p3)
    compare_engine_to_oracle(eng, oracle_arrays(g3, vocab, ids3, out3["read_offsets"], k))
    DERIVED.append(eng.counts().get("derived", 0))
    return len(want_removed)


@pytest.mark.parametrize("seed,N,L,V,k,err", [(7, 400, 30, 300, 5, 0.03), (11, 400, 24, 200, 3, 0.03),
                                              (13, 300, 40, 250, 7, 0.02), (17, 800, 40, 150, 5, 0.05),
                                              (29, 1500, 40, 1000, 5, 0.02), (31, 600, 60, 400, 5, 0.04)])
def test_sweep_synthetic(eng, seed, N, L, V, k, err):
    reads, pos, fq = P.synth_inputs(seed, N, L, V, err)
    run_sweep(eng, reads, pos, fq, k)


@pytest.mark.parametrize("seed,N,L,V,k,err", [(17, 800, 40, 150, 5, 0.05), (31, 600, 60, 400, 5, 0.04), (11, 400, 24, 200, 3, 0.03)])
def test_sweep_with_every_read_rethreaded_by_a_wave(eng, monkeypatch, seed, N, L, V, k, err):
    """AMG_NO_LEAN_GAPPED=1: no read takes the sixteen-lanes-per-read kernel (k_corr_gapped_lean, the default for reads
    whose questions have one answer each): the wave-per-read kernel alone gives the same corrected reads"""
    monkeypatch.setenv("AMG_NO_LEAN_GAPPED", "1")
    reads, pos, fq = P.synth_inputs(seed, N, L, V, err)
    run_sweep(eng, reads, pos, fq, k)


@pytest.mark.parametrize("name,k", [("nine", 3), ("nine", 5), ("three", 5), ("four", 5), ("six", 5)])
def test_sweep_fixture(eng, name, k):
    calls, pos = P.fixture(name)
    lengths = {r: (pos[r][-1][1] + 200 if pos[r] else 100) for r in pos}
    run_sweep(eng, calls, pos, P.FakeFastq(lengths), k)


def test_low_coverage_components(eng):
    from amira_amd import tokenize
    from amira_oracle import GeneMerGraph
    calls, _ = P.fixture("nine")
    vocab, toks, offs, read_ids = tokenize(calls)
    eng.set_reads(toks, offs, vocab.two_v)
    eng.build(3)
    g = GeneMerGraph(calls, 3)
    for m in (5, 40):
        eng.remove_low_coverage_components(m)
        g.remove_low_coverage_components(m)
        compare_engine_to_oracle(eng, oracle_arrays(g, vocab, read_ids, offs, 3), live_only=True)


@pytest.mark.parametrize("shift", [0, 1 << 33])
def test_corrected_positions_as_int32_when_they_fit(eng, shift):
    """Engine.corrected(pos32=True) — what amira_amd.io.DeviceCorrected fetches — hands the positions over as int32
    arrays gathered on the device, the same values as the 64-bit read-back; positions beyond 32 bits come as int64"""
    from amira_amd import tokenize
    reads, pos, fq = P.synth_inputs(17, 800, 40, 150, 0.05)
    vocab, toks, offs, read_ids = tokenize(reads)
    eng.set_reads(toks, offs, vocab.two_v)
    gs, ge = flat_positions(read_ids, reads, pos)
    eng.set_positions(gs + shift, ge + shift, np.asarray([len(fq[r]["sequence"]) for r in read_ids], dtype=np.int64) + shift)
    eng.build(5)
    eng.filter(3, 1)
    n = eng.correct_reads()
    wide, narrow = eng.corrected(*n, True), eng.corrected(*n, True, pos32=True)
    assert narrow["gene_start"].dtype == (np.int64 if shift else np.int32)
    for key in wide:
        assert np.array_equal(wide[key], narrow[key]), key


def test_third_graph_of_a_sweep_is_derived_from_the_second(eng, monkeypatch):
    """a correction that only drops and trims reads (what follows tip clipping) arms the rebuild that reuses the graph at
    hand: the sweeps of this module are compared with the oracle either way — here: the shortcut IS taken where it
    applies, is not with AMG_NO_DERIVE=1, and both give the oracle's graph"""
    reads, pos, fq = P.synth_inputs(7, 400, 30, 300, 0.03)
    del DERIVED[:]
    run_sweep(eng, reads, pos, fq, 5)
    assert DERIVED == [1]
    monkeypatch.setenv("AMG_NO_DERIVE", "1")
    run_sweep(eng, reads, pos, fq, 5)
    assert DERIVED == [1, 0]


@pytest.mark.parametrize("patch", [True, False])
def test_live_lists_after_nodes_died(eng, monkeypatch, patch):
    """filter -> correct_reads (the live forward / backward lists exist now) -> more nodes die (a component filter, then
    listed nodes) -> correct_reads again: the walkers must see the lists of the graph as it is NOW — brought up to date
    in place (k_lr_patch) or made again (AMG_NO_LADJ_PATCH=1) — against the oracle"""
    from amira_amd import tokenize
    from amira_oracle import GeneMerGraph
    if not patch:
        monkeypatch.setenv("AMG_NO_LADJ_PATCH", "1")
    reads, pos, fq = P.synth_inputs(17, 800, 40, 150, 0.05)
    k = 5
    vocab, toks, offs, read_ids = tokenize(reads)
    eng.set_reads(toks, offs, vocab.two_v)
    gs, ge = flat_positions(read_ids, reads, pos)
    eng.set_positions(gs, ge, np.asarray([len(fq[r]["sequence"]) for r in read_ids], dtype=np.int64))
    eng.build(k)
    g = GeneMerGraph(reads, k, {r: list(v) for r, v in pos.items()})
    order = list(g.get_nodes())                                          # engine node id = position here
    eng.filter(3, 1)
    g.filter_graph(3, 1)
    # the engine re-threads: its lists exist from here on (checked against an oracle graph of its own — the reference's
    # correct_reads rewrites the gene positions it was given, so `g` itself corrects only once, below)
    g0 = GeneMerGraph(reads, k, {r: list(v) for r, v in pos.items()})
    g0.filter_graph(3, 1)
    check_corrected(eng, vocab, read_ids, *g0.correct_reads(fq))
    eng.remove_low_coverage_components(12)
    g.remove_low_coverage_components(12)
    # every ninth live node of the middle of the graph goes too (remove_node, construct_graph.py:463-484)
    live = [i for i, h in enumerate(order) if h in g.get_nodes()]
    victims = live[len(live) // 4: 3 * len(live) // 4: 9]
    assert len(victims) > 5
    for i in victims:
        g.remove_node(g.get_node_by_hash(order[i]))
    eng.remove_nodes(victims)
    compare_engine_to_oracle(eng, oracle_arrays(g, vocab, read_ids, offs, k), live_only=True)
    check_corrected(eng, vocab, read_ids, *g.correct_reads(fq))


def test_borrowed_device_inputs(eng):
    """inputs handed over as borrowed device pointers (on_device = 2): same corrected reads as
    with copied inputs, and the caller's arrays are untouched after build / correct / adopt"""
    import torch
    from amira_amd import tokenize
    reads, pos, fq = P.synth_inputs(17, 800, 40, 150, 0.05)
    vocab, toks, offs, read_ids = tokenize(reads)
    gs, ge = flat_positions(read_ids, reads, pos)
    rl = np.asarray([len(fq[r]["sequence"]) for r in read_ids], dtype=np.int64)

    def sweep(borrow):
        d = [torch.from_numpy(np.ascontiguousarray(a)).cuda() for a in (toks, offs, gs, ge, rl)]
        keep = [t.clone() for t in d]
        eng.set_reads_device(d[0].data_ptr(), d[1].data_ptr(), len(offs) - 1, vocab.two_v, borrow=borrow)
        eng.set_positions_device(d[2].data_ptr(), d[3].data_ptr(), d[4].data_ptr(), borrow=borrow)
        outs = []
        for _ in range(2):
            eng.build(5)
            eng.filter(3, 1)
            nr, nt = eng.correct_reads()
            outs.append(eng.corrected(nr, nt, True))
            eng.adopt_corrected()
        eng.build(5)
        torch.cuda.synchronize()
        for a, b in zip(d, keep):
            assert torch.equal(a, b)
        return outs, eng.nodes()["coverage"].copy()

    a, cov_a = sweep(False)
    b, cov_b = sweep(True)
    for x, y in zip(a, b):
        for key in x:
            assert np.array_equal(x[key], y[key]), key
    assert np.array_equal(cov_a, cov_b)


def _tandem_reads(seed, n_reads, L, err):
    """reads over a genome with tandem gene arrays: corrected and original gene lists of such
    reads are near-periodic, so shifted alignments tie with the diagonal one (the position
    carry-over may then not take its equal-length shortcut)"""
    from amira_amd import synth
    rng = np.random.default_rng(seed)
    genome = [(1 if rng.random() < 0.5 else -1, f"g{i}") for i in range(60)]
    for at, name, n in ((12, "t0", 9), (33, "t1", 6), (50, "t2", 12)):
        genome[at:at] = [(1, name)] * n
    names = sorted({g for _, g in genome})
    reads = {}
    for r in range(n_reads):
        s0 = int(rng.integers(0, len(genome) - L + 1))
        seq = list(genome[s0:s0 + L])
        if rng.random() < 0.5:
            seq = [(-st, g) for st, g in reversed(seq)]
        out = []
        for st, g in seq:
            if rng.random() < err:
                g = names[int(rng.integers(0, len(names)))]
            out.append(("+" if st > 0 else "-") + g)
        reads[f"r{r:05d}"] = out
    return reads, synth.positions_for(reads), P.FakeFastq(synth.fake_fastq_lengths(reads))


@pytest.mark.parametrize("seed,k", [(3, 3), (4, 5), (5, 3)])
def test_sweep_tandem_repeats(eng, seed, k):
    reads, pos, fq = _tandem_reads(seed, 700, 26, 0.04)
    run_sweep(eng, reads, pos, fq, k)


def test_position_carry_over_tie_with_shifted_tandem_array(eng):
    """a read whose corrected and original gene lists have the same length, differ in two places
    and are shift-equal up to the second one (a tandem array moved by one gene): a gapped
    alignment ties with the diagonal one, the reference's tie order picks the gapped one, so
    the equal-length shortcut of the carry-over kernel must not fire (found by tools/fuzz_sweep.py)"""
    import json, lzma, os
    from amira_amd import synth
    path = os.path.join(os.path.dirname(__file__), "golden", "data", "nw_tie_case.json.xz")
    d = json.loads(lzma.open(path, "rt").read())
    reads = d["reads"]
    pos = synth.positions_for(reads)
    fq = P.FakeFastq(synth.fake_fastq_lengths(reads))
    run_sweep(eng, reads, pos, fq, d["k"], min_cov=d["min_cov"])


@pytest.mark.parametrize("env", [{"AMG_X_TIGHT_BITS": "1"}, {"AMG_X_GENERIC_K": "1"}, {"AMG_NODE_BUCKETS": "0"}, {"AMG_PLAIN_SYNC": "1"}, {"AMG_X_HEAD_TILES": "3"}, {"AMG_NO_FAST_GAPPED": "1"}, {"AMG_NO_FAST_NW": "1"},
                                 {"AMG_COUNT_INLINE": "1"}, {"AMG_KEY_MODE": "fp"}, {"AMG_X_RANK_SORT": "1"}, {"AMG_NO_FAST_GAPPED": "1", "AMG_NO_FAST_NW": "1"},
                                 {"AMG_NO_GAP_MEMO": "1"}, {"AMG_POS_COMPACT_MIN": "-1000000000"}, {"AMG_EDGE_HOME": "0"}, {"AMG_ADJ_SORT": "1"}, {"AMG_NW_NO_SHORTCUT": "1"}, {"AMG_TEST_NODE_BOUND": "300"},
                                 {"AMG_CLIP_COMPONENTS": "1"}])
def test_sweep_general_kernels(eng, monkeypatch, env):
    """the general (any-size) re-threading / alignment kernels, the inline-atomic counting path, the re-threading
    without its path memo, a position pool that is compacted at every amg_adopt_corrected and an edge pass without
    home slots must give the same
    results as the paths that normally take these reads"""
    for k_, v_ in env.items():
        monkeypatch.setenv(k_, v_)
    reads, pos, fq = P.synth_inputs(17, 800, 40, 150, 0.05)
    run_sweep(eng, reads, pos, fq, 5)
    reads, pos, fq = P.synth_inputs(13, 300, 40, 250, 0.02)
    run_sweep(eng, reads, pos, fq, 7)


@pytest.mark.parametrize("seed,N,L,V,k,err", [(7, 400, 30, 300, 5, 0.03), (17, 800, 40, 150, 5, 0.05)])
def test_positions_32_bit_at_the_boundary(seed, N, L, V, k, err):
    """amg_set_positions32 / amg_get_corrected32: int32 positions in, and out only the positions the carry-over
    produced — everything else named as a slice of the caller's own arrays; after TWO corrections (the second one's
    input positions live partly in the engine's pool) the reconstruction equals the 64-bit read-back"""
    from amira_amd import Engine, tokenize
    reads, pos, fq = P.synth_inputs(seed, N, L, V, err)
    vocab, toks, offs, read_ids = tokenize(reads)
    gs, ge = flat_positions(read_ids, reads, pos)
    rl = np.asarray([len(fq[r]["sequence"]) for r in read_ids], dtype=np.int64)
    wide, narrow = Engine(0), Engine(0)
    try:
        for e, (s_, e_) in ((wide, (gs, ge)), (narrow, (gs.astype(np.int32), ge.astype(np.int32)))):
            e.set_reads(toks, offs, vocab.two_v)
            e.set_positions(s_, e_, rl)
            e.build(k)
            e.filter(3, 1)
        for stage in range(2):
            nw, nn = wide.correct_reads(), narrow.correct_reads()
            assert nw == nn
            want = wide.corrected(*nw, True)
            got = narrow.corrected32(*nn)
            for key in ("tokens", "read_offsets", "orig_read", "changed"):
                assert np.array_equal(got[key], want[key]), key
            o = got["read_offsets"]
            n_own = 0
            for i in range(nn[0]):
                a, b = int(o[i]), int(o[i + 1])
                src = int(got["pos_src"][i])
                if src >= 0:      # a slice of the arrays this engine was given
                    s_, e_ = gs[src:src + b - a], ge[src:src + b - a]
                    n_own += 1
                else:
                    at = -1 - src
                    s_, e_ = got["new_start"][at:at + b - a], got["new_end"][at:at + b - a]
                assert np.array_equal(s_, want["gene_start"][a:b]) and np.array_equal(e_, want["gene_end"][a:b]), (stage, i)
            assert 0 < n_own < nn[0]
            if stage == 0:
                for e in (wide, narrow):
                    e.adopt_corrected()
                    e.build(k)
                    e.remove_short_linear_paths(k)
        # a position beyond 32 bits on the way out is refused, not truncated
        big = Engine(0)
        try:
            big.set_reads(toks, offs, vocab.two_v)
            big.set_positions(gs + (1 << 33), ge + (1 << 33), rl + (1 << 34))
            big.build(k)
            big.filter(3, 1)
            n = big.correct_reads()
            with pytest.raises(Exception, match="32 bits"):
                big.corrected32(*n)
            assert big.corrected(*n, True)["gene_start"].max() > (1 << 33)
        finally:
            big.close()
    finally:
        wide.close()
        narrow.close()


def test_sweep_with_a_hub_of_fifteen_hundred_neighbours(eng):
    """one gene-mer followed by 1 500 different ones: the hub's rows of the adjacency lists (all edges, amg_finalize;
    live edges, tip clipping and the correction) are longer than HUGE_ROW and are put in order through a bitmap over
    the edge ids instead of by rank (huge_row_in_order, amg_device.h); a second, smaller hub takes the rank route"""
    from amira_amd import synth
    reads = {}
    for i in range(1500):
        for c in range(3):
            reads[f"h{i:04d}_{c}"] = ["+a", "+b", "+c", f"+x{i}", f"+y{i}", f"+z{i}"]
    for i in range(200):
        for c in range(3):
            reads[f"s{i:04d}_{c}"] = [f"-u{i}", f"+v{i}", "+d", "-e", "+f"]
    pos = synth.positions_for(reads)
    fq = P.FakeFastq(synth.fake_fastq_lengths(reads))
    run_sweep(eng, reads, pos, fq, 3)
