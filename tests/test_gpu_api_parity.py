"""Host-side API of the HIP-backed GeneMerGraph against the CPU oracle on real fixtures:
topology queries, linear paths, unitig gene strings, gene lookups, single-node removal."""
import pytest

import dump as D
import procedures as P

pytestmark = pytest.mark.gpu


def _pair(name, k, with_pos=True):
    from amira_amd import GeneMerGraph as Product
    from amira_oracle import GeneMerGraph as Oracle
    calls, pos = P.fixture(name)
    mk = lambda cls: cls(dict(calls), k, {r: list(v) for r, v in pos.items()} if with_pos else None)
    return mk(Product), mk(Oracle)


@pytest.mark.parametrize("name,k", [("eight", 3), ("nine", 5), ("seven", 3)])
def test_topology_and_path_queries(name, k):
    p, o = _pair(name, k)
    assert list(p.get_nodes()) == list(o.get_nodes())
    assert list(p.get_edges()) == list(o.get_edges())
    assert p.components() == o.components() and p.get_number_of_component() == o.get_number_of_component()
    assert p.get_mean_node_coverage() == o.get_mean_node_coverage()
    assert p.get_all_node_coverages() == o.get_all_node_coverages()
    hashes = list(o.get_nodes())
    for h in hashes[:150] + hashes[-50:]:
        pn, on = p.get_node_by_hash(h), o.get_node_by_hash(h)
        assert p.get_degree(pn) == o.get_degree(on)
        assert [n.__hash__() for n in p.get_forward_neighbors(pn)] == [n.__hash__() for n in o.get_forward_neighbors(on)]
        assert [n.__hash__() for n in p.get_backward_neighbors(pn)] == [n.__hash__() for n in o.get_backward_neighbors(on)]
        assert p.get_all_neighbor_hashes(pn) == o.get_all_neighbor_hashes(on)
        assert p.get_gene_mer_label(pn) == o.get_gene_mer_label(on)
        assert p.get_reverse_gene_mer_genes(pn) == o.get_reverse_gene_mer_genes(on)
        lp, lo = p.get_linear_path_for_node(pn), o.get_linear_path_for_node(on)
        assert lp == lo
        assert p.get_linear_path_for_node(pn, True) == o.get_linear_path_for_node(on, True)
        if 1 < len(lo) <= 30:
            assert p.get_genes_in_unitig(lp) == o.get_genes_in_unitig(lo)
        for nb in o.get_all_neighbors(on)[:2]:
            pnb = p.get_node_by_hash(nb.__hash__())
            assert p.check_if_nodes_are_adjacent(pn, pnb) == o.check_if_nodes_are_adjacent(on, nb)
            if nb.__hash__() != h:
                assert p.get_edge_hashes_between_nodes(pn, pnb) == o.get_edge_hashes_between_nodes(on, nb)
    for deg in (1, 2, 3):
        assert [n.__hash__() for n in p.get_nodes_with_degree(deg)] == [n.__hash__() for n in o.get_nodes_with_degree(deg)]
    for c in o.components()[:3]:
        assert [n.__hash__() for n in p.get_nodes_in_component(c)] == [n.__hash__() for n in o.get_nodes_in_component(c)]
    some_reads = list(o.get_readNodes())[:40]
    for r in some_reads:
        assert [n.__hash__() for n in p.get_nodes_containing_read(r)] == [n.__hash__() for n in o.get_nodes_containing_read(r)]
    genes = sorted({g[1:] for r in some_reads for g in o.get_reads()[r]})[:8]
    for g in genes:
        assert [n.__hash__() for n in p.get_nodes_containing(g)] == [n.__hash__() for n in o.get_nodes_containing(g)]
    assert set(p.get_AMR_nodes(genes)) == set(o.get_AMR_nodes(genes))
    assert p.collect_reads_in_path(hashes[:20]) == o.collect_reads_in_path(hashes[:20])
    assert p.get_reads_for_nodes(hashes[:20]) == o.get_reads_for_nodes(hashes[:20])


def test_remove_node_and_friends():
    p, o = _pair("eight", 3)
    victims = list(o.get_nodes())[5:60:7]
    for h in victims:
        p.remove_node(p.get_node_by_hash(h))
        o.remove_node(o.get_node_by_hash(h))
    assert D.dump_graph(p) == D.dump_graph(o)
    assert p.get_reads_to_correct() == o.get_reads_to_correct()
    assert p.get_valid_reads_only() == o.get_valid_reads_only()
    assert p.remove_junk_reads(0.8) == o.remove_junk_reads(0.8)
    # and the passes after manual removals behave identically
    assert sorted(p.remove_short_linear_paths(3)) == sorted(o.remove_short_linear_paths(3))
    assert D.dump_graph(p) == D.dump_graph(o)
    fq = P.FakeFastq({r: 10 ** 6 for r in o.get_reads()})
    pg, pp = p.correct_reads(fq)
    og, op = o.correct_reads(fq)
    assert D.dump_corrected(pg, pp) == D.dump_corrected(og, op)


def test_empty_and_degenerate_graphs():
    from amira_amd import GeneMerGraph
    g = GeneMerGraph({}, 0)
    assert g.get_reads() == {} and g.get_kmerSize() == 0 and g.get_minNodeCoverage() == 1
    assert g.get_nodes() == {} and g.get_edges() == {} and g.get_minEdgeCoverage() == 1
    g = GeneMerGraph({"a": [], "b": ["+x"]}, 3)
    assert g.get_nodes() == {} and g.get_short_read_annotations() == {"a": [], "b": ["+x"]}
    assert g.components() == [] and g.correct_reads({}) == ({}, {})
    assert g.filter_graph(3, 1) is g and g.get_total_number_of_reads() == 2


def test_merge_graphs_is_the_single_graph_build():
    """graph_utils.merge_nodes / merge_edges / merge_reads / merge_graphs (reference :17-102): the merged graph of two
    read shards equals the graph built from their reads in sub-graph order, and build_multiprocessed_graph gives the
    cores = 1 result whatever `cores` says (the reference's own merge doubles shared edge coverages: SURVEY section 5)"""
    import procedures as P
    from amira_amd import graph_utils as gu
    reads, pos = P.fixture("nine")
    ids = list(reads)
    shards = [{r: reads[r] for r in ids[i::2]} for i in range(2)]
    pshards = [{r: pos[r] for r in ids[i::2]} for i in range(2)]
    subs = [gu.build_graph(shards[i], 3, pshards[i]) for i in range(2)]
    merged = gu.merge_graphs(subs)
    order = list(shards[0]) + list(shards[1])
    want = gu.build_graph({r: reads[r] for r in order}, 3, {r: pos[r] for r in order})

    def state(g):
        return ([(h, n.get_node_coverage(), n.get_component(), list(n.get_reads())) for h, n in g.get_nodes().items()],
                [(h, e.get_edge_coverage()) for h, e in g.get_edges().items()],
                {r: list(v) for r, v in g.get_readNodes().items()})

    assert state(merged) == state(want)
    again = gu.merge_nodes(subs)
    assert gu.merge_edges(subs, again) is None and gu.merge_reads(subs, again) is None
    assert state(again) == state(want)
    one = gu.build_multiprocessed_graph(reads, 3, 1, pos)
    two = gu.build_multiprocessed_graph(reads, 3, 2, pos)
    assert state(one) == state(two)
    for g in subs + [merged, want, again, one, two]:
        g.close()


def test_pickle_round_trip():
    """graph_utils.py:108-122: loky workers return whole GeneMerGraph objects; the state that travels is what the graph
    was made from plus the device passes applied since, replayed on the receiving side's own engine"""
    import pickle
    p, o = _pair("nine", 5)
    q = pickle.loads(pickle.dumps(p))
    assert D.dump_graph(q) == D.dump_graph(o) == D.dump_graph(p)
    for g in (p, o):
        g.filter_graph(3, 1)
        g.remove_node(g.get_node_by_hash(list(g.get_nodes())[7]))
        g.remove_short_linear_paths(5)
    blob = pickle.dumps(p)
    p.close()                         # the copy owes nothing to the original's engine
    q2 = pickle.loads(blob)
    assert D.dump_graph(q2) == D.dump_graph(o)
    assert q2.get_reads_to_correct() == o.get_reads_to_correct()
    assert (q2.get_minNodeCoverage(), q2.get_minEdgeCoverage()) == (o.get_minNodeCoverage(), o.get_minEdgeCoverage())
    fq = P.FakeFastq({r: 10 ** 6 for r in o.get_reads()})
    qg, qp = q2.correct_reads(fq)
    og, op = o.correct_reads(fq)
    assert D.dump_corrected(qg, qp) == D.dump_corrected(og, op)
    # array-backed mappings whose arrays still live on the device travel as arrays
    import numpy as np
    from amira_amd import GeneMerGraph, tokenize
    from amira_amd.io import ReadLengths, TokenizedPositions, TokenizedReads
    reads, pos, fq = P.synth_inputs(17, 800, 40, 150, 0.05)
    vocab, toks, offs, ids = tokenize(reads)
    gs = np.concatenate([[x[0] for x in pos[r]] for r in ids]).astype(np.int64)
    ge = np.concatenate([[x[1] for x in pos[r]] for r in ids]).astype(np.int64)
    g = GeneMerGraph(TokenizedReads(vocab, toks, offs, ids), 5, TokenizedPositions(ids, offs, gs, ge), _filter=(3, 1))
    rt, pt = g.correct_reads(ReadLengths(ids, np.asarray([len(fq[r]["sequence"]) for r in ids], np.int64)))
    g2 = GeneMerGraph(rt, 5, pt)
    assert rt.device_source() is not None
    g3 = pickle.loads(pickle.dumps(g2))
    assert list(g3.get_nodes()) == list(g2.get_nodes())
    assert {r: list(v) for r, v in g3.get_readNodes().items()} == {r: list(v) for r, v in g2.get_readNodes().items()}
    back = pickle.loads(pickle.dumps(g))      # a filtered build keeps its filter
    assert list(back.get_nodes()) == list(g.get_nodes()) and back.get_reads_to_correct() == g.get_reads_to_correct()
    for x in (q, q2, g, g2, g3, back):
        x.close()


def test_remove_edge_then_device_passes():
    """remove_edge followed by filter_graph works in the reference (construct_graph.py:409-428, 523-540): on a graph
    nobody has edited by hand the edge is removed on the device and device passes carry on"""
    p, o = _pair("eight", 3)
    doomed = list(o.get_edges())[3:40:5]
    for h in doomed:
        p.remove_edge(h)
        o.remove_edge(h)
    p.remove_edge(12345)              # unknown hashes are ignored
    assert not p._host_edits
    assert D.dump_graph(p) == D.dump_graph(o)
    p.filter_graph(3, 2)
    o.filter_graph(3, 2)
    assert D.dump_graph(p) == D.dump_graph(o)
    fq = P.FakeFastq({r: 10 ** 6 for r in o.get_reads()})
    pg, pp = p.correct_reads(fq)
    og, op = o.correct_reads(fq)
    assert D.dump_corrected(pg, pp) == D.dump_corrected(og, op)
    # tip clipping walks adjacencies from either end: with one direction of an adjacency gone it says so ...
    with pytest.raises(RuntimeError, match="twin"):
        p.remove_short_linear_paths(3)
    p.close()
    # ... and carries on once the twins are gone too (what filter_graph / remove_node always do)
    p, o = _pair("eight", 3)
    for h in doomed:
        for g in (p, o):
            if h not in g.get_edges():
                continue      # (removed as the twin of an earlier one)
            e = g.get_edge_by_hash(h)
            back = [x for x in g.get_edges().values()
                    if x.get_sourceNode() == e.get_targetNode() and x.get_targetNode() == e.get_sourceNode()
                    and x.get_sourceNodeDirection() == -e.get_targetNodeDirection()
                    and x.get_targetNodeDirection() == -e.get_sourceNodeDirection()]
            g.remove_edge(h)
            for x in back:
                g.remove_edge(x.__hash__())
    assert D.dump_graph(p) == D.dump_graph(o)
    assert sorted(p.remove_short_linear_paths(3)) == sorted(o.remove_short_linear_paths(3))
    assert D.dump_graph(p) == D.dump_graph(o)
    p.close()
