"""Host logic of the array-backed mappings (amira_amd.io) that stand in for the reference's {read: genes} /
{read: [(start, end)]} / FASTQ dicts: lookups, subsets, the redirection of corrected reads, read lengths."""
import numpy as np

from amira_amd.io import ReadLengths, TokenizedPositions, TokenizedReads, _gather_rows
from amira_amd.tokens import Vocabulary


def _mappings():
    names = [f"g{i}" for i in range(6)]
    vocab = Vocabulary(names)
    reads = {"a": ["+g0", "-g1"], "b": ["+g2"], "c": [], "d": ["-g3", "+g4", "+g5", "-g0"]}
    ids = list(reads)
    toks = np.asarray([vocab._tok[g] for r in ids for g in reads[r]], np.int32)
    offs = np.concatenate([[0], np.cumsum([len(reads[r]) for r in ids])]).astype(np.int64)
    gs = np.arange(len(toks), dtype=np.int64) * 100
    ge = gs + 50
    return reads, ids, TokenizedReads(vocab, toks, offs, ids), TokenizedPositions(ids, offs, gs, ge)


def test_lookups_and_subsets():
    reads, ids, tr, tp = _mappings()
    assert list(tr) == ids and len(tr) == 4 and "c" in tr and "x" not in tr
    for r in ids:
        assert tr[r] == reads[r]
    assert tr.gene_at("d", -1) == "-g0" and tr.gene_at("a", 0) == "+g0"
    assert tp["d"] == [(300, 350), (400, 450), (500, 550), (600, 650)] and tp["c"] == []
    assert tp.pos_at("d", 1) == (400, 450)
    idx, off = _gather_rows(tr.read_offsets, [3, 0])
    assert idx.tolist() == [3, 4, 5, 6, 0, 1] and off.tolist() == [0, 4, 6]
    sub = tr.subset([3, 1])
    assert list(sub) == ["d", "b"] and sub["d"] == reads["d"] and sub["b"] == reads["b"]
    assert sub.source_rows.tolist() == [3, 1] and sub.subset([1]).source_rows.tolist() == [1]
    psub = tp.subset([3, 1])
    assert list(psub) == ["d", "b"] and psub["d"] == tp["d"] and psub["b"] == [(200, 250)]


def test_redirected_positions_and_copies():
    _, ids, tr, tp = _mappings()
    other = TokenizedPositions(["d", "a"], np.array([0, 2, 3]), np.array([7, 8, 9]), np.array([17, 18, 19]))
    mine = tp.copy()                       # what a driver works on: the caller's mapping stays as it was
    mine.replace_rows(np.array([3, 0]), other, np.array([0, 1]))
    assert mine["d"] == [(7, 17), (8, 18)] and mine["a"] == [(9, 19)] and mine["b"] == [(200, 250)]
    assert mine.pos_at("d", 1) == (8, 18)
    assert tp["d"] == [(300, 350), (400, 450), (500, 550), (600, 650)] and tp["a"] == [(0, 50), (100, 150)]
    assert mine.subset([3, 1])["d"] == [(7, 17), (8, 18)]     # subsets follow the redirection
    again = mine.copy()
    assert again["d"] == [(7, 17), (8, 18)]
    mine["b"] = [(1, 2)]                   # hand-set positions (bubble popping replaces a read's list)
    assert mine["b"] == [(1, 2)] and again["b"] == [(200, 250)]


def test_read_lengths():
    ids = ["a", "b", "c"]
    rl = ReadLengths(ids, [5, 6, 7])
    assert len(rl["b"]["sequence"]) == 6 and list(rl) == ids
    assert rl.lengths_array(ids) is rl.lengths
    assert rl.lengths_array(["c", "x", "a"]).tolist() == [7, 0, 5]
    assert rl.lengths_array(["c", "a"], rows_hint=np.array([2, 0])).tolist() == [7, 5]
    assert rl.lengths_array(["c", "a"], rows_hint=np.array([1, 0])).tolist() == [7, 5]   # a wrong hint is not trusted
    # a hint that is right at both ends and wrong in the middle (the same reads in another order) is not trusted either
    rl5 = ReadLengths(["a", "b", "c", "d", "e"], [1, 2, 3, 4, 5])
    assert rl5.lengths_array(["a", "c", "b", "e"], rows_hint=np.array([0, 1, 2, 4])).tolist() == [1, 3, 2, 5]
    # rows of ANOTHER list, out of range here at the front (a subset's rows need not ascend): no IndexError
    assert rl5.lengths_array(["e", "a"], rows_hint=np.array([9, 0])).tolist() == [5, 1]
    assert rl5.lengths_array(["e", "a"], rows_hint=np.array([-1, 0])).tolist() == [5, 1]
    # the identity token: rows into this very list are taken as they are
    assert rl5.lengths_array(["d", "b"], rows_hint=np.array([3, 1]), hint_ids=rl5.read_ids).tolist() == [4, 2]
    assert rl5.lengths_array(["d", "b"], rows_hint=np.array([3, 1]), hint_ids=["a", "b", "c", "d", "e"]).tolist() == [4, 2]


def test_source_ids_follow_subsets():
    _, ids, tr, _ = _mappings()
    assert tr.source_rows is None and tr.source_ids is None
    sub = tr.subset([3, 1])
    assert sub.source_ids is tr.read_ids
    assert sub.subset([1]).source_ids is tr.read_ids and sub.subset([1]).source_rows.tolist() == [1]


def test_names_ending_with_a_suffix():
    _, ids, tr, _ = _mappings()
    assert not tr.any_name_ends_with("_reverse")
    vocab = tr.vocab
    other = TokenizedReads(vocab, np.zeros(0, np.int32), np.zeros(4, np.int64), ["a", "b_reverse", "c"])
    assert other.any_name_ends_with("_reverse") and not other.any_name_ends_with("_reversed")
    inner = TokenizedReads(vocab, np.zeros(0, np.int32), np.zeros(3, np.int64), ["a_reverse_b", "c"])
    assert not inner.any_name_ends_with("_reverse")


def test_engine_leases_and_pool():
    """an engine whose buffers still hold the output of a correct_reads stays out of the pool until the last such output
    has been fetched or dropped (amira_amd.engine.release_engine / lease_done; no device needed for the bookkeeping)"""
    import weakref
    from amira_amd import engine as E
    from amira_amd.io import DeviceCorrected

    class FakeEngine:
        def __init__(self):
            self._h, self.device = True, 97
            self._leases, self._pool_when_free = weakref.WeakSet(), False
            self.fetched = 0

        def corrected(self, n_reads, n_tokens, have_pos, pos32=False):
            self.fetched += 1
            z = np.zeros(n_tokens, np.int64)
            return {"tokens": z.astype(np.int32), "gene_start": z, "gene_end": z}

    pool = E._ENGINE_POOL.setdefault(97, [])
    del pool[:]
    eng = FakeEngine()
    a, b = DeviceCorrected(eng, 2, 5, True), DeviceCorrected(eng, 2, 5, True)
    E.release_engine(eng)                       # the graph lets go: two outputs still live in the buffers
    assert pool == [] and eng._pool_when_free
    assert a.fetch()["tokens"].shape == (5,) and a.engine() is None and eng.fetched == 1
    a.fetch()
    assert eng.fetched == 1 and pool == []      # fetched once; the other output still holds the engine
    del b                                       # dropped without being read
    assert pool == [eng] and not eng._pool_when_free
    assert E.acquire_engine(97) is eng and pool == []
    E.release_engine(eng)                       # no leases: back at once
    assert pool == [eng]
    del pool[:]
