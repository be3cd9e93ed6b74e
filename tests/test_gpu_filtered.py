"""amg_build_filtered == amg_build + amg_filter (GeneMerGraph.__init__ + filter_graph, the opening of every
cleaning iteration): live nodes / edges / coverages / list orders / masked windows / reads to correct, and the
whole sweep that follows, on small cases; the full-size check is tests/test_gpu_fullsize.py."""
import json
import lzma
import os

import numpy as np
import pytest

import procedures as P
from helpers import live_arrays
from test_gpu_sweep import flat_positions

pytestmark = pytest.mark.gpu

CASES = [("synth", 17, 800, 40, 150, 5, 0.05), ("synth", 13, 300, 40, 250, 7, 0.02), ("synth", 5, 500, 30, 60, 3, 0.04),
         ("fixture", "nine", 3), ("fixture", "five", 5), ("fixture", "four", 5)]


def _inputs(case):
    from amira_amd import tokenize
    if case[0] == "synth":
        _, seed, N, L, V, k, err = case
        reads, pos, fq = P.synth_inputs(seed, N, L, V, err)
    else:
        reads, pos = P.fixture(case[1])
        k = case[2]
        fq = None
    vocab, toks, offs, ids = tokenize(reads)
    return reads, pos, fq, k, vocab, toks, offs, ids


def _same_live(a, b, what):
    ga, gb = live_arrays(a), live_arrays(b)
    for key in gb:
        assert np.array_equal(ga[key], gb[key]), (what, key)


@pytest.mark.parametrize("thr", [(3, 1), (2, 2), (1, 1), (5, 3)])
@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("key_mode", ["exact", "fp"])
def test_build_filtered_equals_build_then_filter(case, thr, key_mode, monkeypatch):
    from amira_amd import Engine
    if key_mode == "fp":   # fingerprint keys: the entry point falls back to the two calls
        monkeypatch.setenv("AMG_KEY_MODE", "fp")
    reads, pos, fq, k, vocab, toks, offs, ids = _inputs(case)
    a, b = Engine(0), Engine(0)
    try:
        for e in (a, b):
            e.set_reads(toks, offs, vocab.two_v)
        a.build_filtered(k, *thr)
        b.build(k)
        b.filter(*thr)
        _same_live(a, b, "after the filter")
        ca, cb = a.counts(), b.counts()
        assert ca["n_live_nodes"] == cb["n_live_nodes"] and ca["n_live_edges"] == cb["n_live_edges"]
        if key_mode == "exact":
            assert ca["n_nodes"] == cb["n_live_nodes"]          # only the survivors exist
        # component ids are the UNFILTERED graph's (the reference labels once, in __init__, construct_graph.py:101-102)
        na_, nb_ = a.nodes(), b.nodes()
        assert np.array_equal(na_["component"][na_["alive"] != 0], nb_["component"][nb_["alive"] != 0])
        assert a.counts()["n_components"] == b.counts()["n_components"]
        # what follows sees no difference: correction, rebuild
        na, nb = a.correct_reads(), b.correct_reads()
        assert na == nb
        xa, xb = a.corrected(*na, False), b.corrected(*nb, False)
        for key in ("tokens", "read_offsets", "orig_read", "changed"):
            assert np.array_equal(xa[key], xb[key]), key
    finally:
        a.close()
        b.close()


@pytest.mark.parametrize("case", CASES[:4])
def test_sweep_with_filtered_first_build(case):
    """the cleaning sweep with its first two steps fused against the sweep as the reference spells it"""
    from amira_amd import Engine
    reads, pos, fq, k, vocab, toks, offs, ids = _inputs(case)
    out = []
    for fused in (True, False):
        e = Engine(0)
        e.set_reads(toks, offs, vocab.two_v)
        if pos is not None:
            gs, ge = flat_positions(ids, reads, pos)
            rl = np.asarray([max([x[1] for x in pos[r]] + [0]) + 50 for r in ids], np.int64)
            e.set_positions(gs, ge, rl)
        if fused:
            e.build_filtered(k, 3, 1)
        else:
            e.build(k)
            e.filter(3, 1)
        n1 = e.correct_reads()
        c1 = e.corrected(*n1, pos is not None)
        e.adopt_corrected()
        e.build(k)
        rem = np.sort(e.remove_short_linear_paths(k))
        n2 = e.correct_reads()
        c2 = e.corrected(*n2, pos is not None)
        e.adopt_corrected()
        e.build(k)
        out.append((c1, rem, c2, e.nodes(), e.edges(), e.read_nodes()))
        e.close()
    (c1a, ra, c2a, na, ea, ta), (c1b, rb, c2b, nb, eb, tb) = out
    for x, y in ((c1a, c1b), (c2a, c2b), (na, nb), (ea, eb)):
        for key in y:
            if y[key] is not None:
                assert np.array_equal(x[key], y[key]), key
    assert np.array_equal(ra, rb) and np.array_equal(ta[0], tb[0]) and np.array_equal(ta[1], tb[1])


@pytest.mark.parametrize("thr", [(10 ** 6, 1), (1, 10 ** 6), (0, 0)])
def test_extreme_thresholds_and_empty_input(thr):
    """everything dropped, every edge dropped, nothing dropped; and no reads at all"""
    from amira_amd import Engine, tokenize
    reads, _, _ = P.synth_inputs(3, 200, 20, 50, 0.05)
    vocab, toks, offs, _ = tokenize(reads)
    a, b = Engine(0), Engine(0)
    try:
        for e in (a, b):
            e.set_reads(toks, offs, vocab.two_v)
        a.build_filtered(5, *thr)
        b.build(5)
        b.filter(*thr)
        _same_live(a, b, thr)
        assert a.correct_reads() == b.correct_reads()
        a.finalize()
        a.set_reads(np.zeros(0, np.int32), np.zeros(1, np.int64), 2)
        a.build_filtered(3, 3, 1)
        assert a.counts()["n_nodes"] == 0
    finally:
        a.close()
        b.close()


@pytest.mark.parametrize("thr", [(3, 1), (2, 2)])
@pytest.mark.parametrize("case", CASES)
def test_component_passes_after_build_filtered(case, thr):
    """what reads the component labels — remove_short_linear_paths' whole-component guard (:702-713) and
    remove_low_coverage_components (:950-958) — gives the same graph after the one-pass build + filter"""
    from amira_amd import Engine
    reads, pos, fq, k, vocab, toks, offs, ids = _inputs(case)
    a, b = Engine(0), Engine(0)
    try:
        for e in (a, b):
            e.set_reads(toks, offs, vocab.two_v)
        a.build_filtered(k, *thr)
        b.build(k)
        b.filter(*thr)
        live_b = np.cumsum(b.nodes()["alive"] != 0) - 1
        ra, rb = a.remove_short_linear_paths(k), b.remove_short_linear_paths(k)
        assert sorted(ra.tolist()) == sorted(live_b[rb].tolist())
        _same_live(a, b, "after tip clipping")
        a.remove_low_coverage_components(5)
        b.remove_low_coverage_components(5)
        _same_live(a, b, "after the component filter")
    finally:
        a.close()
        b.close()
