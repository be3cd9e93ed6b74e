"""Native gene-call I/O: JSON files <-> CSR tokens without a
per-gene Python loop.  `load_gene_calls` returns a TokenizedReads, which GeneMerGraph /
build_graph accept in place of the {read: [genes]} dict (it IS a read-only mapping of that
shape, decoded lazily), so the tokens go straight to the device.
"""
import ctypes as C
from collections.abc import Mapping

import numpy as np

from . import _ffi
from ._ffi import check, ptr
from .tokens import Vocabulary


class DeviceCorrected:
    """What one GeneMerGraph.correct_reads produced, still on the device: the corrected genes and their positions
    (240 MB + 960 MB for a million 60-gene reads).  The array-backed mappings correct_reads hands back point here;
    the arrays cross PCIe only if somebody looks at them on the host — a GeneMerGraph built from those mappings takes
    them over device to device (amg_set_reads_from_corrected).  Holds the ENGINE whose buffers these are — not the
    graph, whose own gene positions point back here (a cycle would leave graph and engine to the cyclic collector) —
    and keeps it out of the engine pool until the set has been fetched or dropped."""

    def __init__(self, engine, n_reads, n_tokens, have_pos):
        self._engine, self._shape, self._arrays = engine, (n_reads, n_tokens, have_pos), None
        engine._leases.add(self)

    def engine(self):
        """the engine that still holds the set (None once it has been fetched)"""
        return self._engine

    def fetch(self):
        if self._arrays is None:
            # (positions as int32 when they fit: gathered on the device, half the bytes over PCIe)
            out = self._engine.corrected(*self._shape, pos32=True)
            self._arrays = {"tokens": out["tokens"], "gene_start": out["gene_start"], "gene_end": out["gene_end"]}
            self._done()
        return self._arrays

    def __reduce__(self):
        """pickled as what it holds: the arrays, fetched (a device buffer does not travel)"""
        return (_fetched, (self.fetch(), self._shape))

    def _done(self):
        engine, self._engine = self._engine, None
        if engine is not None:
            from .engine import lease_done
            lease_done(engine, self)

    def __del__(self):
        try:
            self._done()
        except Exception:  # noqa: BLE001 - interpreter shutdown
            pass


class _Fetched(DeviceCorrected):
    """a DeviceCorrected after its trip through pickle: the arrays only, no engine"""

    def __init__(self, arrays, shape):   # noqa: super().__init__ not called: there is no engine to lease
        self._engine, self._shape, self._arrays = None, shape, arrays


def _fetched(arrays, shape):
    return _Fetched(arrays, shape)


def _gather_rows(offsets, rows):
    """(flat element indices, new offsets) of the CSR rows `rows`, in that order"""
    rows = np.asarray(rows, dtype=np.int64)
    n = offsets[rows + 1] - offsets[rows]
    new_off = np.zeros(len(rows) + 1, np.int64)
    np.cumsum(n, out=new_off[1:])
    idx = np.repeat(offsets[rows] - new_off[:-1], n) + np.arange(int(new_off[-1]), dtype=np.int64)
    return idx, new_off


class TokenizedReads(Mapping):
    """{read id: ["+geneA", ...]} backed by CSR token arrays (lists are decoded on access)."""

    def __init__(self, vocab, tokens, read_offsets, read_ids, source_rows=None, source_ids=None):
        # tokens: an int32 array, or a DeviceCorrected (the genes are still on the device: see the property)
        self.vocab, self._tokens, self.read_offsets, self.read_ids = vocab, tokens, read_offsets, read_ids
        # where these reads sat in the read list they descend from (correct_reads drops reads and keeps the order):
        # lets array-backed side tables (ReadLengths) follow without a lookup per read; None = they are that list.
        # source_ids: that list itself (the object), so that a side table can tell whether the rows are ITS rows
        self.source_rows = source_rows
        self.source_ids = source_ids if source_rows is not None else None
        self._index = None   # built on the first lookup by name (a million reads: ~0.2 s)
        self._cache = {}

    @property
    def tokens(self):
        if isinstance(self._tokens, DeviceCorrected):
            self._tokens = self._tokens.fetch()["tokens"]
        return self._tokens

    def device_source(self):
        """the DeviceCorrected these genes still live in, if nobody has asked for them on the host yet"""
        t = self._tokens
        return t if isinstance(t, DeviceCorrected) and t.engine() is not None else None

    def _idx(self):
        if self._index is None:
            self._index = dict(zip(self.read_ids, range(len(self.read_ids))))
        return self._index

    def any_name_ends_with(self, suffix):
        """is there a read whose name ends with `suffix`?  (asked once per mapping: read-path clustering names the
        reverse complement of read r "r_reverse", and a read really called that would count twice)"""
        memo = self.__dict__.setdefault("_suffix_memo", {})
        if suffix not in memo:
            # one C-level join and search instead of a million method calls; a name that itself holds a line break can
            # only make this answer True where it is not (the caller then takes the slower, name-by-name route)
            memo[suffix] = (suffix + "\n") in ("\n".join(self.read_ids) + "\n")
        return memo[suffix]

    def __getitem__(self, read_id):
        got = self._cache.get(read_id)
        if got is None:
            i = self._idx()[read_id]
            got = self._cache[read_id] = self.vocab.decode(
                self.tokens[self.read_offsets[i]:self.read_offsets[i + 1]])
        return got

    def subset(self, rows):
        """the reads at `rows` (indices into read_ids), in that order, as a TokenizedReads of their own"""
        if self.edited():
            return self.settled().subset(rows)
        rows = np.asarray(rows, dtype=np.int64)
        idx, new_off = _gather_rows(self.read_offsets, rows)
        ids = np.asarray(self.read_ids, dtype=object)[rows].tolist()
        src = rows if self.source_rows is None else self.source_rows[rows]
        return TokenizedReads(self.vocab, self.tokens[idx], new_off, ids, source_rows=src,
                              source_ids=self.read_ids if self.source_rows is None else self.source_ids)

    def gene_at(self, read_id, i):
        """self[read_id][i] without decoding the rest of the read"""
        got = self._cache.get(read_id)
        if got is not None:
            return got[i]
        r = self._idx()[read_id]
        a, b = int(self.read_offsets[r]), int(self.read_offsets[r + 1])
        if i < 0:
            i += b - a
        if not 0 <= i < b - a:
            raise IndexError(i)
        return self.vocab.gene(int(self.tokens[a + i]))

    def __setitem__(self, read_id, genes):   # bubble popping rewrites a read's genes (construct_graph.py:1534-1545)
        if read_id not in self._idx():
            raise KeyError(f"{read_id}: only the reads this mapping was made with can be replaced")
        self._cache[read_id] = genes
        self.__dict__.setdefault("_edited", set()).add(read_id)

    def edited(self):
        return bool(self.__dict__.get("_edited"))

    def settled(self):
        """this mapping as arrays: itself while no read has been replaced, otherwise a new TokenizedReads with the
        replaced reads spelled into the arrays (the untouched ones are copied in one piece)"""
        edited = self.__dict__.get("_edited")
        if not edited:
            return self
        idx = self._idx()
        rows = np.fromiter((idx[r] for r in edited), np.int64, len(edited))
        order = np.argsort(rows)
        rows = rows[order]
        token_of = self.vocab.token
        try:
            new_rows = [np.fromiter((token_of(g) for g in self._cache[self.read_ids[i]]), np.int32) for i in rows.tolist()]
        except (KeyError, ValueError, AssertionError):   # a gene the vocabulary has never seen: start over from the strings
            from .tokens import tokenize
            return TokenizedReads(*tokenize({r: self[r] for r in self.read_ids}))
        tokens, offs = _replace_rows(self.read_offsets, rows, new_rows, (self.tokens,))
        return TokenizedReads(self.vocab, tokens[0], offs, self.read_ids, source_rows=self.source_rows,
                              source_ids=self.source_ids)

    def to_dict(self):
        """{read id: [genes]} as a plain dict: every gene spelled in ONE pass over the tokens, the reads cut out of that
        list (a lookup per read decodes its own slice: a million small numpy slices)"""
        s = self.settled()
        genes, offs = s.vocab.decode(s.tokens), s.read_offsets.tolist()
        return {r: genes[offs[i]:offs[i + 1]] for i, r in enumerate(s.read_ids)}

    def __iter__(self):
        return iter(self.read_ids)

    def __len__(self):
        return len(self.read_ids)

    def __contains__(self, read_id):
        return read_id in self._idx()


def _replace_rows(offsets, rows, new_rows, columns, new_columns=None):
    """CSR arrays `columns` (all over `offsets`) with the rows `rows` (ascending) replaced: new_rows[i] is row rows[i]
    of the first column (new_columns[c][i] of column c when there are several).  Returns (new columns, new offsets);
    the untouched rows move in one gather."""
    n = len(offsets) - 1
    lens = np.diff(offsets)
    lens[rows] = [len(x) for x in new_rows]
    new_off = np.zeros(n + 1, np.int64)
    np.cumsum(lens, out=new_off[1:])
    keep = np.ones(n, bool)
    keep[rows] = False
    kept = np.flatnonzero(keep)
    src, _ = _gather_rows(offsets, kept)
    dst, _ = _gather_rows(new_off, kept)
    out = []
    for c, col in enumerate(columns):
        fresh = np.empty(int(new_off[-1]), col.dtype)
        fresh[dst] = col[src]
        values = new_rows if c == 0 else new_columns[c]
        for i, r in enumerate(rows.tolist()):
            fresh[new_off[r]:new_off[r + 1]] = values[i]
        out.append(fresh)
    return out, new_off


class TokenizedPositions(Mapping):
    """{read id: [(start, end), ...]} backed by two flat int64 arrays aligned with the tokens of a
    TokenizedReads; GeneMerGraph hands the arrays to the device as they are (no per-read loop)."""

    def __init__(self, read_ids, read_offsets, gene_start, gene_end):
        # gene_start / gene_end: int64 arrays, or one DeviceCorrected for both (still on the device)
        self.read_ids, self.read_offsets = read_ids, read_offsets
        self._gs, self._ge = gene_start, gene_end
        self._index = None
        self._cache = {}
        self._moved = None   # (row -> row of another TokenizedPositions or -1, that other mapping): replace_rows

    def _from_device(self):
        if isinstance(self._gs, DeviceCorrected):
            got = self._gs.fetch()
            self._gs, self._ge = got["gene_start"], got["gene_end"]

    @property
    def gene_start(self):
        self._from_device()
        return self._gs

    @property
    def gene_end(self):
        self._from_device()
        return self._ge

    def device_source(self):
        g = self._gs
        return g if isinstance(g, DeviceCorrected) and g.engine() is not None and self._moved is None and not self._cache else None

    def _idx(self):
        if self._index is None:
            self._index = dict(zip(self.read_ids, range(len(self.read_ids))))
        return self._index

    def __getitem__(self, read_id):
        got = self._cache.get(read_id)
        if got is None:
            i = self._idx()[read_id]
            src, j = self, i
            if self._moved is not None and self._moved[0][i] >= 0:   # replaced wholesale by a correction
                src, j = self._moved[1], int(self._moved[0][i])
            a, b = int(src.read_offsets[j]), int(src.read_offsets[j + 1])
            got = self._cache[read_id] = list(zip(src.gene_start[a:b].tolist(), src.gene_end[a:b].tolist()))
        return got

    def subset(self, rows):
        """the positions of the reads at `rows`, in that order, as a TokenizedPositions of their own"""
        rows = np.asarray(rows, dtype=np.int64)
        if self._moved is not None or self._cache:   # redirected / hand-set reads: row by row
            ids = [self.read_ids[i] for i in rows.tolist()]
            per = [self[r] for r in ids]
            new_off = np.zeros(len(ids) + 1, np.int64)
            np.cumsum([len(p) for p in per], out=new_off[1:])
            gs = np.fromiter((x[0] for p in per for x in p), dtype=np.int64, count=int(new_off[-1]))
            ge = np.fromiter((x[1] for p in per for x in p), dtype=np.int64, count=int(new_off[-1]))
            return TokenizedPositions(ids, new_off, gs, ge)
        idx, new_off = _gather_rows(self.read_offsets, rows)
        ids = np.asarray(self.read_ids, dtype=object)[rows].tolist()
        return TokenizedPositions(ids, new_off, self.gene_start[idx], self.gene_end[idx])

    def copy(self):
        """the caller's own copy (the arrays are shared — they are never written — the redirections are not)"""
        c = TokenizedPositions(self.read_ids, self.read_offsets, self._gs, self._ge)
        c._index = self._index
        c._cache = dict(self._cache)
        if self.__dict__.get("_edited"):
            c._edited = set(self._edited)
        if self._moved is not None:
            c._moved = (self._moved[0].copy(), self._moved[1])
        return c

    def to_dict(self):
        """{read id: [(start, end), ...]} as a plain dict, the pairs made in one pass"""
        s = self.settled()
        pairs, offs = list(zip(s.gene_start.tolist(), s.gene_end.tolist())), s.read_offsets.tolist()
        return {r: pairs[offs[i]:offs[i + 1]] for i, r in enumerate(s.read_ids)}

    def replace_rows(self, rows, other, other_rows):
        """the positions of reads `rows` (indices into read_ids) are from now on rows `other_rows` of the
        TokenizedPositions `other` — what GeneMerGraph.correct_reads does to the caller's gene positions for every
        read it changed (construct_graph.py:1282-1284, :1328), for a million reads at once"""
        if self._moved is not None and self._moved[1] is not other:   # a second correction on the same mapping
            pinned = self.__dict__.setdefault("_edited", set())
            for i in np.flatnonzero(self._moved[0] >= 0).tolist():
                self[self.read_ids[i]]       # pin what the first one left (rare: the drivers rebuild in between)
                pinned.add(self.read_ids[i])
            self._moved = None
        if self._moved is None:
            self._moved = (np.full(len(self.read_ids), -1, np.int64), other)
        self._moved[0][rows] = other_rows
        edited = self.__dict__.get("_edited")
        for i in np.asarray(rows).tolist() if len(self._cache) else ():
            self._cache.pop(self.read_ids[i], None)
            if edited:
                edited.discard(self.read_ids[i])

    def pos_at(self, read_id, i):
        """self[read_id][i] without building the read's list of pairs"""
        got = self._cache.get(read_id)
        if got is not None:
            return got[i]
        if self._moved is not None:
            return self[read_id][i]
        r = self._idx()[read_id]
        a, b = int(self.read_offsets[r]), int(self.read_offsets[r + 1])
        if i < 0:
            i += b - a
        if not 0 <= i < b - a:
            raise IndexError(i)
        return int(self.gene_start[a + i]), int(self.gene_end[a + i])

    def __setitem__(self, read_id, value):   # correct_reads / bubble popping replace a read's positions
        if read_id not in self._idx():
            raise KeyError(f"{read_id}: only the reads this mapping was made with can be replaced")
        self._cache[read_id] = value
        self.__dict__.setdefault("_edited", set()).add(read_id)

    def edited(self):
        return bool(self.__dict__.get("_edited"))

    def as_made(self):
        """nothing has been redirected (replace_rows) or replaced by hand since the arrays were made"""
        return self._moved is None and not self.__dict__.get("_edited")

    def settled(self):
        """this mapping as arrays: itself while nothing has been redirected or replaced, otherwise a new
        TokenizedPositions — the redirected reads gathered from the mapping they point at, the hand-set ones spelled in
        from their lists, everything else copied in one piece"""
        if self.as_made():
            return self
        gs, ge, offs = self.gene_start, self.gene_end, self.read_offsets
        if self._moved is not None:
            to, other = self._moved
            moved = np.flatnonzero(to >= 0)
            if len(moved):
                o_idx, o_off = _gather_rows(other.read_offsets, to[moved])
                starts, ends = other.gene_start[o_idx], other.gene_end[o_idx]
                cuts = o_off[1:-1]
                (gs, ge), offs = _replace_rows(offs, moved, np.split(starts, cuts), (gs, ge),
                                               {1: np.split(ends, cuts)})
        edited = self.__dict__.get("_edited")
        if edited:
            idx = self._idx()
            rows = np.sort(np.fromiter((idx[r] for r in edited), np.int64, len(edited)))
            lists = [self._cache[self.read_ids[i]] for i in rows.tolist()]
            (gs, ge), offs = _replace_rows(offs, rows, [[p[0] for p in l] for l in lists], (gs, ge),
                                           {1: [[p[1] for p in l] for l in lists]})
        return TokenizedPositions(self.read_ids, offs, gs, ge)

    def __iter__(self):
        return iter(self.read_ids)

    def __len__(self):
        return len(self.read_ids)

    def __contains__(self, read_id):
        return read_id in self._idx()


class _Sized:
    """stands in for a read's nucleotide string where only its length is asked for"""
    __slots__ = ("n",)

    def __init__(self, n):
        self.n = n

    def __len__(self):
        return self.n


class ReadLengths(Mapping):
    """{read id: {"sequence": <something with the read's length>}} backed by one int64 array — the only thing
    correct_reads wants from the FASTQ dict (len(fastq[read]["sequence"]), construct_graph.py:1685).
    GeneMerGraph.correct_reads takes the whole array at once (lengths_array) instead of a million lookups."""

    def __init__(self, read_ids, lengths):
        self.read_ids, self.lengths = read_ids, np.ascontiguousarray(lengths, np.int64)
        self._index = None

    def _idx(self):
        if self._index is None:
            self._index = dict(zip(self.read_ids, range(len(self.read_ids))))
        return self._index

    def __getitem__(self, read_id):
        return {"sequence": _Sized(int(self.lengths[self._idx()[read_id]]))}

    def __iter__(self):
        return iter(self.read_ids)

    def __len__(self):
        return len(self.read_ids)

    def lengths_array(self, read_ids, rows_hint=None, hint_ids=None):
        """lengths of `read_ids` in that order (0 for a read this mapping does not know).  rows_hint: where the
        caller believes these reads sit in the read list `hint_ids` (TokenizedReads.source_rows / source_ids).  The
        hint is taken as it is when that list IS this mapping's list (same object) and the rows are in range;
        otherwise only after every row has been checked against the names."""
        if read_ids is self.read_ids:
            return self.lengths
        if rows_hint is not None and len(rows_hint) == len(read_ids) and len(read_ids) > 0:
            rows = np.asarray(rows_hint, dtype=np.int64)
            if int(rows.min()) >= 0 and int(rows.max()) < len(self.read_ids):
                if hint_ids is not None and hint_ids is self.read_ids:
                    return self.lengths[rows]
                mine = self.__dict__.get("_ids_arr")
                if mine is None:
                    mine = self._ids_arr = np.asarray(self.read_ids, dtype=object)
                if np.array_equal(mine[rows], np.asarray(read_ids, dtype=object)):
                    return self.lengths[rows]
        if len(read_ids) == len(self.read_ids) and len(read_ids) > 0 and read_ids[0] == self.read_ids[0] \
                and read_ids[-1] == self.read_ids[-1] and read_ids == self.read_ids:
            return self.lengths
        idx = self._idx()
        rows = np.fromiter((idx.get(r, -1) for r in read_ids), dtype=np.int64, count=len(read_ids))
        return np.where(rows >= 0, self.lengths[np.maximum(rows, 0)], 0)


def _split(buf):
    return [] if len(buf) == 0 else buf.tobytes()[:-1].decode("utf-8").split("\0")


class _ArrayPool:
    """The loader's big output arrays (tokens, gene starts / ends: 0.24 + 2 x 0.48 GB for a million 60-gene reads) are
    fresh heap on every call, and a gigabyte of fresh heap is a quarter of a million page faults.  An array the pool
    handed out is given out again once nobody else refers to it any more (its reference count is back to the pool's
    own: every view, slice or mapping built on it holds a reference to it, so an array somebody can still see is never
    reused); otherwise a new one is made.  The test is CPython's reference count: a consumer that keeps only a raw
    POINTER into an array (ptr(a), a.ctypes.data) beyond the call it made it for must keep the array itself referenced
    too — everything in this package does.  `clear()` drops the pool's arrays (pre_processing.trim_buffers calls it)."""

    def __init__(self, keep=6):
        self._held, self._keep = [], keep

    def empty(self, n, dtype):
        import sys
        dtype = np.dtype(dtype)
        for a in self._held:
            # 3 = the list's reference + the loop variable + getrefcount's own argument
            if a.dtype == dtype and a.size >= n and a.size <= 2 * n + 1024 and sys.getrefcount(a) == 3:
                return a[:n] if a.size != n else a
        a = np.empty(n, dtype)
        if n >= (1 << 20):
            self._held.append(a)
            if len(self._held) > self._keep:
                self._held.pop(0)
        return a

    def clear(self):
        self._held.clear()


_pool = _ArrayPool()


def load_gene_calls(calls_json, positions_json=None, want_blanks=False):
    """-> TokenizedReads (and, with positions_json, {read: [(start, end), ...]} as two flat
    int64 arrays aligned with the tokens: (reads, gene_start, gene_end)); want_blanks adds whether some gene name of
    the file held a blank (names are stored with '_' in its place)."""
    h = C.c_void_p()
    check(_ffi.lib.amg_calls_load_json(str(calls_json).encode(), C.byref(h)))
    try:
        n = [C.c_int64(0) for _ in range(5)]
        check(_ffi.lib.amg_calls_counts(h, *[C.byref(x) for x in n]))
        n_reads, n_tokens, n_genes, names_bytes, ids_bytes = [x.value for x in n]
        tokens = _pool.empty(n_tokens, np.int32)
        offs = np.empty(n_reads + 1, np.int64)
        names = np.empty(names_bytes, np.uint8)
        ids = np.empty(ids_bytes, np.uint8)
        hashes = np.empty(n_genes * 32, np.uint8)
        check(_ffi.lib.amg_calls_get(h, ptr(tokens), ptr(offs), ptr(names), ptr(ids), ptr(hashes)))
        vocab = Vocabulary.from_ranked(_split(names), hashes.reshape(n_genes, 32))
        reads = TokenizedReads(vocab, tokens, offs, _split(ids))
        blanks = C.c_int32(0)
        if want_blanks:
            check(_ffi.lib.amg_calls_has_blanks(h, C.byref(blanks)))
        if positions_json is None:
            return (reads, bool(blanks.value)) if want_blanks else reads
        gs, ge = _pool.empty(n_tokens, np.int64), _pool.empty(n_tokens, np.int64)
        check(_ffi.lib.amg_calls_load_positions_json(h, str(positions_json).encode(), ptr(gs), ptr(ge)))
        return (reads, gs, ge, bool(blanks.value)) if want_blanks else (reads, gs, ge)
    finally:
        _ffi.lib.amg_calls_free(h)


def write_gene_calls(path, vocab, tokens, read_offsets, read_ids):
    """corrected CSR -> {"read": ["+gene", ...]} JSON (what result_utils.py:1260-1264 dumps)."""
    tokens = np.ascontiguousarray(tokens, np.int32)
    read_offsets = np.ascontiguousarray(read_offsets, np.int64)
    names = ("\0".join(vocab.names) + "\0").encode("utf-8") if vocab.names else b""
    ids = ("\0".join(read_ids) + "\0").encode("utf-8") if len(read_ids) else b""
    check(_ffi.lib.amg_calls_write_json(str(path).encode(), ptr(tokens), ptr(read_offsets), len(read_ids),
                                        names, len(vocab.names), ids))


def write_gene_positions(path, gene_start, gene_end, read_offsets, read_ids):
    """flat positions -> {"read": [[start, end], ...]} JSON (the second file of result_utils.py:1260-1264)"""
    narrow = getattr(gene_start, "dtype", None) == np.int32 and getattr(gene_end, "dtype", None) == np.int32
    gs = np.ascontiguousarray(gene_start, np.int32 if narrow else np.int64)
    ge = np.ascontiguousarray(gene_end, np.int32 if narrow else np.int64)
    read_offsets = np.ascontiguousarray(read_offsets, np.int64)
    ids = ("\0".join(read_ids) + "\0").encode("utf-8") if len(read_ids) else b""
    fn = _ffi.lib.amg_calls_write_positions_json32 if narrow else _ffi.lib.amg_calls_write_positions_json
    check(fn(str(path).encode(), ptr(gs), ptr(ge), ptr(read_offsets), len(read_ids), ids))
