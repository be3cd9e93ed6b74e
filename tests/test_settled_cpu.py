"""Array-backed read / position mappings with reads replaced by hand (bubble popping) or redirected (correct_reads)
settle into arrays that say what the mapping says read by read; amg_nw_align (host code of libamg) gives the alignment
of the Python needleman_wunsch, ties included."""
import numpy as np
import pytest

from amira_amd.io import TokenizedPositions, TokenizedReads
from amira_amd.tokens import tokenize


def _reads(rng, n=40, v=12):
    names = [f"g{i}" for i in range(v)]
    return {f"r{i}": [("+" if rng.random() < 0.5 else "-") + names[int(rng.integers(0, v))]
                      for _ in range(int(rng.integers(0, 9)))] for i in range(n)}


def _positions(reads):
    return {r: [(10 * i, 10 * i + 7) for i in range(len(g))] for r, g in reads.items()}


def _flat(pos, ids):
    gs = np.asarray([p[0] for r in ids for p in pos[r]], np.int64)
    ge = np.asarray([p[1] for r in ids for p in pos[r]], np.int64)
    return gs, ge


@pytest.mark.parametrize("seed", range(6))
def test_reads_replaced_by_hand_settle_into_the_arrays(seed):
    rng = np.random.default_rng(seed)
    reads = _reads(rng)
    t = TokenizedReads(*tokenize(reads))
    assert t.settled() is t
    want = {r: list(g) for r, g in reads.items()}
    for r in list(reads)[:: 3 + seed % 3]:
        new = list(reversed(want[r])) + ["+g0"] * int(rng.integers(0, 3))
        if rng.random() < 0.3:
            new = []
        t[r] = new
        want[r] = new
    s = t.settled()
    assert s is not t and list(s) == list(reads)
    assert {r: s[r] for r in s} == want and {r: t[r] for r in t} == want
    assert s.read_offsets[-1] == len(s.tokens) == sum(len(g) for g in want.values())
    again = TokenizedReads(*tokenize(want))
    assert np.array_equal(again.read_offsets, s.read_offsets)
    assert s.vocab.decode(s.tokens) == again.vocab.decode(again.tokens)
    with pytest.raises(KeyError):
        t["a read nobody has seen"] = ["+g1"]


def test_a_gene_the_vocabulary_has_not_seen():
    rng = np.random.default_rng(11)
    reads = _reads(rng)
    t = TokenizedReads(*tokenize(reads))
    t["r3"] = ["+brand_new", "-g1"]
    s = t.settled()
    assert s["r3"] == ["+brand_new", "-g1"] and s["r4"] == reads["r4"] and list(s) == list(reads)


@pytest.mark.parametrize("seed", range(6))
def test_positions_redirected_and_replaced_settle_into_the_arrays(seed):
    rng = np.random.default_rng(100 + seed)
    reads = _reads(rng)
    ids = list(reads)
    pos = _positions(reads)
    t = TokenizedReads(*tokenize(reads))
    p = TokenizedPositions(ids, t.read_offsets, *_flat(pos, ids))
    assert p.settled() is p and p.as_made()
    want = {r: list(v) for r, v in pos.items()}
    # a correction redirects some reads to rows of another mapping
    other_ids = [f"o{i}" for i in range(7)]
    other = {o: [(1000 + 3 * i, 1001 + 3 * i) for i in range(int(rng.integers(0, 6)))] for o in other_ids}
    o_off = np.zeros(len(other_ids) + 1, np.int64)
    np.cumsum([len(other[o]) for o in other_ids], out=o_off[1:])
    op = TokenizedPositions(other_ids, o_off, *_flat(other, other_ids))
    rows = np.sort(rng.choice(len(ids), 7, replace=False))
    p.replace_rows(rows, op, np.arange(7))
    for i, r in enumerate(rows.tolist()):
        want[ids[r]] = other[other_ids[i]]
    # bubble popping sets some by hand (one of them a redirected read)
    for r in [ids[int(rows[0])], ids[1], ids[-1]]:
        new = [(5, 6)] * int(rng.integers(0, 4))
        p[r] = new
        want[r] = new
    c = p.copy()
    for q in (p, c):
        s = q.settled()
        assert s is not q and s.as_made()
        assert {r: s[r] for r in s} == {r: list(v) for r, v in want.items()}
        assert s.read_offsets[-1] == len(s.gene_start) == len(s.gene_end) == sum(len(v) for v in want.values())
    assert {r: p[r] for r in p} == {r: list(v) for r, v in want.items()}


@pytest.mark.parametrize("seed", range(30))
def test_native_alignment_equals_the_python_table(seed, monkeypatch):
    from amira_amd import GeneMerGraph
    rng = np.random.default_rng(seed)
    alphabet = ["+a", "-a", "+b", "+c", "-d"][: 2 + seed % 4]
    x = [alphabet[int(i)] for i in rng.integers(0, len(alphabet), int(rng.integers(0, 14)))]
    y = [alphabet[int(i)] for i in rng.integers(0, len(alphabet), int(rng.integers(0, 14)))]
    monkeypatch.setenv("AMG_NW_PYTHON", "1")
    want = GeneMerGraph.needleman_wunsch(None, x, y)
    monkeypatch.delenv("AMG_NW_PYTHON")
    assert GeneMerGraph.needleman_wunsch(None, x, y) == want
    # the helper itself, whatever the size
    from amira_amd import _ffi
    import ctypes as C
    code = {}
    xs = np.asarray([code.setdefault(g, len(code)) for g in x], np.int32)
    ys = np.asarray([code.setdefault(g, len(code)) for g in y], np.int32)
    ops, n = np.empty(len(x) + len(y) + 1, np.int8), C.c_int32(0)
    _ffi.check(_ffi.lib.amg_nw_align(_ffi.ptr(xs), len(x), _ffi.ptr(ys), len(y), _ffi.ptr(ops), C.byref(n)))
    got, i, j = [], 0, 0
    for op in ops[: n.value].tolist():
        got.append((x[i], y[j]) if op == 0 else (x[i], "*") if op == 1 else ("*", y[j]))
        i, j = i + (op != 2), j + (op != 1)
    assert got == want


def test_plain_dicts_of_settled_mappings():
    rng = np.random.default_rng(5)
    reads = _reads(rng)
    ids = list(reads)
    pos = _positions(reads)
    t = TokenizedReads(*tokenize(reads))
    p = TokenizedPositions(ids, t.read_offsets, *_flat(pos, ids))
    t["r2"] = ["+g1", "-g2", "+g3"]
    p["r2"] = [(1, 2), (3, 4), (5, 6)]
    reads["r2"], pos["r2"] = ["+g1", "-g2", "+g3"], [(1, 2), (3, 4), (5, 6)]
    assert t.to_dict() == reads and list(t.to_dict()) == ids
    assert p.to_dict() == {r: list(v) for r, v in pos.items()}


def test_writers_see_the_reads_and_positions_replaced_by_hand(tmp_path):
    """write_pandora_gene_calls (result_utils.py:1260-1264) on array-backed mappings after bubble popping rewrote reads"""
    import json
    from amira_amd.result_utils import write_pandora_gene_calls
    rng = np.random.default_rng(9)
    reads = _reads(rng)
    ids = list(reads)
    pos = _positions(reads)
    t = TokenizedReads(*tokenize(reads))
    p = TokenizedPositions(ids, t.read_offsets, *_flat(pos, ids))
    t["r5"], p["r5"] = ["-g3", "+g1"], [(7, 9), (11, 15)]
    reads["r5"], pos["r5"] = ["-g3", "+g1"], [(7, 9), (11, 15)]
    p["r1"]   # (a lookup alone leaves a list in the cache: not an edit)
    a, b = str(tmp_path / "calls.json"), str(tmp_path / "positions.json")
    write_pandora_gene_calls(str(tmp_path), p, t, a, b)
    assert open(a).read() == json.dumps(reads)
    assert open(b).read() == json.dumps({r: [list(x) for x in v] for r, v in pos.items()})
