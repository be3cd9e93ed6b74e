"""The array-backed way through the reference-shaped API (amira_amd.io.TokenizedReads / TokenizedPositions /
ReadLengths in place of the dicts): graph_utils.cleaning_sweep must hand back the same reads, positions and graph as
with plain dicts — with no per-read Python work in between."""
import numpy as np
import pytest

import procedures as P

pytestmark = pytest.mark.gpu


def _tokenized(reads, pos, fq):
    from amira_amd import tokenize
    from amira_amd.io import ReadLengths, TokenizedPositions, TokenizedReads
    vocab, toks, offs, ids = tokenize(reads)
    gs = np.concatenate([[p[0] for p in pos[r]] for r in ids]).astype(np.int64)
    ge = np.concatenate([[p[1] for p in pos[r]] for r in ids]).astype(np.int64)
    lengths = np.asarray([len(fq[r]["sequence"]) for r in ids], np.int64)
    return TokenizedReads(vocab, toks, offs, ids), TokenizedPositions(ids, offs, gs, ge), ReadLengths(ids, lengths)


@pytest.mark.parametrize("case", [(17, 800, 40, 150, 0.05, 5), (5, 500, 30, 60, 0.04, 3)])
def test_cleaning_sweep_on_arrays_equals_dicts(case):
    from amira_amd import graph_utils as gu
    seed, N, L, V, err, k = case
    reads, pos, fq = P.synth_inputs(seed, N, L, V, err)
    pos_d = {r: list(v) for r, v in pos.items()}
    g_d, reads_d, out_pos_d = gu.cleaning_sweep(dict(reads), pos_d, k, fq, 3)
    treads, tpos, tlen = _tokenized(reads, pos, fq)
    g_t, reads_t, out_pos_t = gu.cleaning_sweep(treads, tpos, k, tlen, 3)
    assert list(reads_t) == list(reads_d)
    for r in reads_d:
        assert reads_t[r] == reads_d[r], r
        assert [tuple(x) for x in out_pos_t[r]] == [tuple(x) for x in out_pos_d[r]], r
    # the drivers work on their own copies (the reference's .copy() / comprehensions): the caller's mappings are as before
    for r in pos_d:
        assert [tuple(x) for x in tpos[r]] == [tuple(x) for x in pos[r]] == [tuple(x) for x in pos_d[r]], r
    assert list(g_t.get_nodes()) == list(g_d.get_nodes())
    assert [(h, e.get_edge_coverage()) for h, e in g_t.get_edges().items()] == \
           [(h, e.get_edge_coverage()) for h, e in g_d.get_edges().items()]
    assert {r: list(v) for r, v in g_t.get_readNodes().items()} == {r: list(v) for r, v in g_d.get_readNodes().items()}
    g_d.close()
    g_t.close()


def test_removed_node_hashes_without_the_object_view():
    from amira_amd import GeneMerGraph
    reads, pos, fq = P.synth_inputs(17, 800, 40, 150, 0.05)
    g = GeneMerGraph(dict(reads), 5)
    g.filter_graph(3, 1)
    corrected, _ = g.correct_reads(fq)
    g.close()
    a, b = GeneMerGraph(dict(corrected), 5), GeneMerGraph(dict(corrected), 5)
    b.get_nodes()                       # b answers from its object view, a from the node tokens
    assert a._view is None
    ha, hb = a.remove_short_linear_paths(5), b.remove_short_linear_paths(5)
    assert ha == hb and len(ha) > 0
    a.close()
    b.close()


def test_correct_reads_updates_the_positions_it_was_given():
    """GeneMerGraph.correct_reads replaces the positions of every read it changed in the mapping the graph was built
    with (construct_graph.py:1282-1284, :1328) — also when that mapping is array-backed"""
    from amira_amd import GeneMerGraph
    reads, pos, fq = P.synth_inputs(17, 800, 40, 150, 0.05)
    pos_d = {r: list(v) for r, v in pos.items()}
    treads, tpos, tlen = _tokenized(reads, pos, fq)
    gd, gt = GeneMerGraph(dict(reads), 5, pos_d), GeneMerGraph(treads, 5, tpos)
    gd.filter_graph(3, 1)
    gt.filter_graph(3, 1)
    rd, pd = gd.correct_reads(fq)
    rt, pt = gt.correct_reads(tlen)
    assert list(rt) == list(rd)
    changed = 0
    for r in pos_d:
        assert [tuple(x) for x in tpos[r]] == [tuple(x) for x in pos_d[r]], r
        changed += [tuple(x) for x in pos_d[r]] != [tuple(x) for x in pos[r]]
    assert changed > 0
    for r in rd:
        assert rt[r] == rd[r] and [tuple(x) for x in pt[r]] == [tuple(x) for x in pd[r]]
    gd.close()
    gt.close()


@pytest.mark.parametrize("rate", [0.8, 0.5, 0.25])
def test_remove_junk_reads_on_arrays_equals_dicts(rate):
    """remove_junk_reads (:1398-1420) from the device's per-window ids — dict inputs and array-backed inputs — against
    the per-read loop over the object view (which a host edit switches back on)"""
    from amira_amd import GeneMerGraph
    reads, pos, fq = P.synth_inputs(17, 800, 40, 150, 0.05)
    pos_d = {r: list(v) for r, v in pos.items()}
    treads, tpos, _ = _tokenized(reads, pos, fq)
    gd, gt, gl = GeneMerGraph(dict(reads), 5, pos_d), GeneMerGraph(treads, 5, tpos), GeneMerGraph(dict(reads), 5, pos_d)
    for g in (gd, gt, gl):
        g.filter_graph(3, 1)
    gl.get_readNodes()
    gl._host_edits = True            # the reference-shaped loop
    want = gl.remove_junk_reads(rate)
    assert len(want[0]) > 0 and len(want[2]) > 0
    for got in (gd.remove_junk_reads(rate), gt.remove_junk_reads(rate)):
        for a, b in zip(got, want):
            assert list(a) == list(b)
            for r in b:
                assert [tuple(x) if isinstance(x, (list, tuple)) else x for x in a[r]] == \
                       [tuple(x) if isinstance(x, (list, tuple)) else x for x in b[r]]
    for g in (gd, gt, gl):
        g.close()


def test_corrected_reads_go_to_the_next_graph_on_the_device():
    """the output of correct_reads on array-backed inputs stays on the device (amira_amd.io.DeviceCorrected): the next
    GeneMerGraph takes it over device to device, the host arrays appear only when somebody reads them, and the engine
    that holds them goes back to the pool only then; nothing of this hangs in a reference cycle"""
    from amira_amd import GeneMerGraph
    reads, pos, fq = P.synth_inputs(17, 800, 40, 150, 0.05)
    pos_d = {r: list(v) for r, v in pos.items()}
    treads, tpos, tlen = _tokenized(reads, pos, fq)
    gd, gt = GeneMerGraph(dict(reads), 5, pos_d), GeneMerGraph(treads, 5, tpos)
    gd.filter_graph(3, 1)
    gt.filter_graph(3, 1)
    rd, pd = gd.correct_reads(fq)
    rt, pt = gt.correct_reads(tlen)
    assert rt.device_source() is not None and pt.device_source() is rt.device_source()
    from amira_amd.engine import _ENGINE_POOL
    held = rt.device_source().engine()
    gt.close()
    # the graph has let go of its engine, the corrected set keeps it alive and out of the pool
    assert gt._engine is None and held._h and all(held is not e for e in _ENGINE_POOL.get(held.device, []))
    g2d, g2t = GeneMerGraph(rd, 5, pd), GeneMerGraph(rt, 5, pt)
    assert rt.device_source() is not None             # nothing crossed PCIe for the rebuild
    assert list(g2t.get_nodes()) == list(g2d.get_nodes())
    assert [(h, e.get_edge_coverage()) for h, e in g2t.get_edges().items()] == \
           [(h, e.get_edge_coverage()) for h, e in g2d.get_edges().items()]
    assert {r: list(v) for r, v in g2t.get_readNodePositions().items()} == \
           {r: list(v) for r, v in g2d.get_readNodePositions().items()}   # reads the host positions: fetched now
    # (the drivers ask for hashes that are only made when looked at; a plain call returns the reference's list)
    clipped_t, clipped_d = g2t.remove_short_linear_paths(5, _lazy_hashes=True), g2d.remove_short_linear_paths(5)
    assert isinstance(clipped_d, list) and len(clipped_t) == len(clipped_d) > 0
    assert list(clipped_t) == clipped_d and clipped_t == clipped_d
    r2d, p2d = g2d.correct_reads(fq)
    r2t, p2t = g2t.correct_reads(tlen)
    assert list(r2t) == list(r2d)
    for r in r2d:
        assert r2t[r] == r2d[r] and [tuple(x) for x in p2t[r]] == [tuple(x) for x in p2d[r]]
    # fetched -> the first graph's engine went back (the second graph has taken it from the pool or it waits there)
    assert rt.device_source() is None and len(held._leases) == 0 and not held._pool_when_free
    for g in (gd, g2d, g2t):
        g.close()
    import weakref
    probe = weakref.ref(g2t)
    del g2t, g2d, gd, gt, r2t, p2t, rt, pt, g
    assert probe() is None   # freed by reference counting: a graph in a cycle would take its engine to the collector


def test_mappings_of_a_correction_survive_later_passes():
    """correct_reads on array inputs hands back mappings whose arrays still live on the device; a pass that changes
    the graph afterwards (it invalidates the engine's corrected set) must not take them away: the reference's dicts
    keep answering whatever happens to the graph later"""
    from amira_amd import GeneMerGraph
    reads, pos, fq = P.synth_inputs(17, 800, 40, 150, 0.05)
    pos_d = {r: list(v) for r, v in pos.items()}
    treads, tpos, tlen = _tokenized(reads, pos, fq)
    gd, gt = GeneMerGraph(dict(reads), 5, pos_d), GeneMerGraph(treads, 5, tpos)
    gd.filter_graph(3, 1)
    gt.filter_graph(3, 1)
    rd, pd = gd.correct_reads(fq)
    rt, pt = gt.correct_reads(tlen)
    assert rt.device_source() is not None
    gt.remove_short_linear_paths(5)               # settles the lease first
    gd.remove_short_linear_paths(5)
    assert rt.device_source() is None
    some = list(rd)[:50]
    for r in some:
        assert rt[r] == rd[r] and [tuple(x) for x in pt[r]] == [tuple(x) for x in pd[r]]
    gd.close()
    gt.close()
    # ... and the same through filter_graph / remove_node / remove_low_coverage_components (a fresh pair of graphs each
    # time: the reference's correct_reads is not meant to run twice on one graph — it would align the ORIGINAL genes
    # with positions its first run has already replaced)
    for mutate in (lambda g: g.filter_graph(4, 2), lambda g: g.remove_node(next(iter(g.get_nodes().values()))),
                   lambda g: g.remove_low_coverage_components(5)):
        treads, tpos, tlen = _tokenized(reads, pos, fq)
        gd, gt = GeneMerGraph(dict(reads), 5, {r: list(v) for r, v in pos.items()}), GeneMerGraph(treads, 5, tpos)
        gd.filter_graph(3, 1)
        gt.filter_graph(3, 1)
        r2t, p2t = gt.correct_reads(tlen)
        r2d, p2d = gd.correct_reads(fq)
        assert r2t.device_source() is not None
        mutate(gt)
        mutate(gd)
        assert r2t.device_source() is None
        assert list(r2t) == list(r2d)
        for r in list(r2d)[:50]:
            assert r2t[r] == r2d[r] and [tuple(x) for x in p2t[r]] == [tuple(x) for x in p2d[r]]
        g3 = GeneMerGraph(r2t, 5, p2t)           # (host arrays by now)
        g3d = GeneMerGraph(r2d, 5, p2d)
        assert list(g3.get_nodes()) == list(g3d.get_nodes())
        for g in (g3, g3d, gd, gt):
            g.close()


def test_int32_positions_through_the_api():
    """array-backed positions handed over as int32 (read coordinates fit) stay int32 on the way to the device
    (amg_set_positions32): the sweep returns the same reads, positions and graph as with int64 arrays"""
    from amira_amd import graph_utils as gu
    from amira_amd.io import TokenizedPositions
    reads, pos, fq = P.synth_inputs(17, 800, 40, 150, 0.05)
    treads, tpos, tlen = _tokenized(reads, pos, fq)
    tpos32 = TokenizedPositions(tpos.read_ids, tpos.read_offsets, tpos.gene_start.astype(np.int32),
                                tpos.gene_end.astype(np.int32))
    g64, r64, p64 = gu.cleaning_sweep(treads, tpos, 5, tlen, 3)
    g32, r32, p32 = gu.cleaning_sweep(treads, tpos32, 5, tlen, 3)
    assert g32._gs_val is None or g32._gs_val.dtype in (np.int32, np.int64)
    assert list(r32) == list(r64)
    for r in r64:
        assert r32[r] == r64[r] and [tuple(x) for x in p32[r]] == [tuple(x) for x in p64[r]], r
    assert list(g32.get_nodes()) == list(g64.get_nodes())
    g64.close()
    g32.close()
