"""Thin array-level wrapper over the C ABI (include/amg.h): numpy in, numpy out.
This is the layer bench.py times; amira_amd.construct_graph builds the reference's
object API on top of it."""
import ctypes as C
import weakref

import numpy as np

from . import _ffi
from ._ffi import check, ptr


class Engine:
    def __init__(self, device=0):
        self._h = C.c_void_p()
        check(_ffi.lib.amg_create(int(device), C.byref(self._h)))
        self.device = device
        # outputs of a correct_reads that still live in this engine's buffers (amira_amd.io.DeviceCorrected): while
        # there are any, the engine stays out of the pool
        self._leases = weakref.WeakSet()
        self._pool_when_free = False

    def close(self):
        if self._h:
            _ffi.lib.amg_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- inputs
    def set_reads(self, tokens, read_offsets, two_v):
        tokens = np.ascontiguousarray(tokens, dtype=np.int32)
        read_offsets = np.ascontiguousarray(read_offsets, dtype=np.int64)
        check(_ffi.lib.amg_set_reads(self._h, ptr(tokens), ptr(read_offsets),
                                     len(read_offsets) - 1, int(two_v), 0))

    def set_reads_device(self, tokens_ptr, read_offsets_ptr, n_reads, two_v, borrow=False):
        """device pointers; borrow=True: no copy, the caller keeps the arrays alive and unchanged
        until the next set_reads* / adopt_corrected"""
        check(_ffi.lib.amg_set_reads(self._h, C.c_void_p(tokens_ptr), C.c_void_p(read_offsets_ptr),
                                     int(n_reads), int(two_v), 2 if borrow else 1))

    def set_positions(self, gene_start, gene_end, read_len=None):
        """int32 arrays travel as they are (amg_set_positions32: half the bytes over PCIe), anything else as int64"""
        rl = None if read_len is None else np.ascontiguousarray(read_len, dtype=np.int64)
        if getattr(gene_start, "dtype", None) == np.int32 and getattr(gene_end, "dtype", None) == np.int32:
            gs, ge = np.ascontiguousarray(gene_start), np.ascontiguousarray(gene_end)
            check(_ffi.lib.amg_set_positions32(self._h, ptr(gs), ptr(ge), ptr(rl), 0))
            return
        gs = np.ascontiguousarray(gene_start, dtype=np.int64)
        ge = np.ascontiguousarray(gene_end, dtype=np.int64)
        check(_ffi.lib.amg_set_positions(self._h, ptr(gs), ptr(ge), ptr(rl), 0))

    def set_positions32_device(self, gs_ptr, ge_ptr, rl_ptr):
        """int32 position arrays already on the device (copied and widened: the caller's arrays are free afterwards)"""
        check(_ffi.lib.amg_set_positions32(self._h, C.c_void_p(gs_ptr), C.c_void_p(ge_ptr),
                                           C.c_void_p(rl_ptr) if rl_ptr else None, 1))

    def set_read_lengths(self, read_len):
        rl = np.ascontiguousarray(read_len, dtype=np.int64)
        check(_ffi.lib.amg_set_read_lengths(self._h, ptr(rl), 0))

    def set_positions_device(self, gs_ptr, ge_ptr, rl_ptr, borrow=False):
        check(_ffi.lib.amg_set_positions(self._h, C.c_void_p(gs_ptr), C.c_void_p(ge_ptr),
                                         C.c_void_p(rl_ptr) if rl_ptr else None, 2 if borrow else 1))

    # ---- build + counts
    def build(self, k):
        check(_ffi.lib.amg_build(self._h, int(k)))

    def build_filtered(self, k, min_node_cov, min_edge_cov):
        """build + filter_graph(min_node_cov, min_edge_cov) with the filter applied on the way (amg.h)"""
        check(_ffi.lib.amg_build_filtered(self._h, int(k), max(int(min_node_cov), 0), max(int(min_edge_cov), 0)))

    @staticmethod
    def build_multi(engines, ks):
        """the graphs of engines[0]'s reads for every k of ks, graph i on engines[i] (amg_build_multi: the reads are on
        the device once); the other engines borrow engines[0]'s read arrays: keep it alive and unchanged"""
        n = len(engines)
        assert n == len(ks) and n >= 1
        handles = (C.c_void_p * n)(*[e._h for e in engines])
        arr = (C.c_int32 * n)(*[int(k) for k in ks])
        check(_ffi.lib.amg_build_multi(handles, arr, n))

    def finalize(self):
        """component ids + per-node edge lists of the built graph now (they are otherwise made on first use)"""
        check(_ffi.lib.amg_finalize(self._h))

    def sizes(self):
        """(reads, genes) of the current read set; no device work"""
        nr, nt = C.c_int64(0), C.c_int64(0)
        check(_ffi.lib.amg_sizes(self._h, C.byref(nr), C.byref(nt)))
        return nr.value, nt.value

    def graph_sizes(self):
        """(nodes, directed edges, k) of the built graph; no device work"""
        nn, ne, k = C.c_int64(0), C.c_int64(0), C.c_int32(0)
        check(_ffi.lib.amg_graph_sizes(self._h, C.byref(nn), C.byref(ne), C.byref(k)))
        return nn.value, ne.value, k.value

    def counts(self):
        c = _ffi.Counts()
        check(_ffi.lib.amg_counts(self._h, C.byref(c)))
        return c.as_dict()

    def sync(self):
        check(_ffi.lib.amg_sync(self._h))

    def stream(self):
        return _ffi.lib.amg_stream(self._h)

    def set_timing(self, on):
        """per-stage HIP-event timing (two events per stage, ~5 us of stream idle each): on by default"""
        check(_ffi.lib.amg_set_timing(self._h, 1 if on else 0))

    def timings(self):
        names = (C.c_char_p * 64)()
        ms = (C.c_float * 64)()
        n = _ffi.lib.amg_last_timings(self._h, names, ms, 64)
        return [(names[i].decode(), float(ms[i])) for i in range(max(n, 0))]

    # ---- read-back
    @staticmethod
    def _take(buf, key, shape, dtype):
        """an output array: a view of the caller's (e.g. pinned) buffer buf[key] when given, else a new one"""
        n = int(np.prod(shape))
        if buf is not None and key in buf:
            a = buf[key]
            assert a.dtype == dtype and a.size >= n and a.flags["C_CONTIGUOUS"], key
            return a.reshape(-1)[:n].reshape(shape)
        return np.empty(shape, dtype)

    def nodes(self, buf=None):
        """buf: optional dict of preallocated arrays (capacity >= needed) the columns are read back into"""
        D, _, k = self.graph_sizes()
        out = {
            "tokens": self._take(buf, "tokens", (D, k), np.int32), "coverage": self._take(buf, "coverage", (D,), np.uint32),
            "first_token": self._take(buf, "first_token", (D,), np.int64), "first_dir": self._take(buf, "first_dir", (D,), np.int8),
            "component": self._take(buf, "component", (D,), np.int32), "alive": self._take(buf, "alive", (D,), np.uint8),
        }
        check(_ffi.lib.amg_get_nodes(self._h, ptr(out["tokens"]), ptr(out["coverage"]),
                                     ptr(out["first_token"]), ptr(out["first_dir"]),
                                     ptr(out["component"]), ptr(out["alive"])))
        return out

    def edges(self, buf=None):
        E = self.graph_sizes()[1]
        out = {"src": self._take(buf, "src", (E,), np.int32), "tgt": self._take(buf, "tgt", (E,), np.int32),
               "sdir": self._take(buf, "sdir", (E,), np.int8), "tdir": self._take(buf, "tdir", (E,), np.int8),
               "coverage": self._take(buf, "ecoverage", (E,), np.uint32), "alive": self._take(buf, "ealive", (E,), np.uint8)}
        check(_ffi.lib.amg_get_edges(self._h, ptr(out["src"]), ptr(out["tgt"]), ptr(out["sdir"]),
                                     ptr(out["tdir"]), ptr(out["coverage"]), ptr(out["alive"])))
        return out

    def read_nodes(self, buf=None):
        T = self.sizes()[1]
        node, d = self._take(buf, "tok_node", (T,), np.int32), self._take(buf, "tok_dir", (T,), np.int8)
        check(_ffi.lib.amg_get_read_nodes(self._h, ptr(node), ptr(d)))
        return node, d

    def read_node_ids(self, buf=None):
        """the node id per window alone (the directions stay on the device)"""
        T = self.sizes()[1]
        node = self._take(buf, "tok_node", (T,), np.int32)
        check(_ffi.lib.amg_get_read_nodes(self._h, ptr(node), None))
        return node

    def read_node_ids_of(self, first_token, n_windows):
        """the node ids of the windows of a few reads laid end to end: row r = windows first_token[r] ..
        first_token[r] + n_windows[r] - 1; returns (node ids, start of every row in them)"""
        first_token = np.ascontiguousarray(first_token, np.int64)
        starts = np.zeros(len(first_token) + 1, np.int64)
        np.cumsum(n_windows, out=starts[1:])
        out = np.empty(int(starts[-1]), np.int32)
        check(_ffi.lib.amg_get_read_nodes_rows(self._h, ptr(first_token), ptr(starts), len(first_token), ptr(out)))
        return out, starts

    def read_dirs(self, buf=None):
        """the gene-mer direction per window alone"""
        T = self.sizes()[1]
        d = self._take(buf, "tok_dir", (T,), np.int8)
        check(_ffi.lib.amg_get_read_nodes(self._h, None, ptr(d)))
        return d

    def node_adj(self):
        D, E, _ = self.graph_sizes()
        off, ids = np.empty(2 * D + 1, np.int64), np.empty(E, np.int32)
        check(_ffi.lib.amg_get_node_adj(self._h, ptr(off), ptr(ids)))
        return off, ids

    def node_reads(self):
        D = self.graph_sizes()[0]
        off = np.empty(D + 1, np.int64)
        # one call: the list cannot be longer than the windows of the read set (the pages of the buffer that are never
        # written are never touched); the two-call spelling of the C ABI builds the lists twice
        idx = np.empty(max(self.sizes()[1], 1), np.int32)
        check(_ffi.lib.amg_get_node_reads(self._h, ptr(off), ptr(idx)))
        return off, idx[:int(off[-1]) if D else 0]

    def reads_to_correct(self):
        f = np.empty(self.sizes()[0], np.uint8)
        check(_ffi.lib.amg_get_reads_to_correct(self._h, ptr(f)))
        return f

    # ---- passes
    def filter(self, min_node_cov, min_edge_cov):
        check(_ffi.lib.amg_filter(self._h, int(min_node_cov), int(min_edge_cov)))

    def remove_nodes(self, ids):
        ids = np.ascontiguousarray(ids, dtype=np.int32)
        check(_ffi.lib.amg_remove_nodes(self._h, ptr(ids), len(ids)))

    def remove_edges(self, ids):
        ids = np.ascontiguousarray(ids, dtype=np.int32)
        check(_ffi.lib.amg_remove_edges(self._h, ptr(ids), len(ids)))

    def remove_short_linear_paths(self, min_length, protect=None, want_ids=True):
        """ids of the removed nodes (ascending); want_ids=False: only their number — nothing but one count leaves the
        device (callers that go on with the graph on the device and ignore the list, as the reference's drivers do)"""
        n = C.c_int64(0)
        pr = None if protect is None else np.ascontiguousarray(protect, dtype=np.uint8)
        if not want_ids:
            check(_ffi.lib.amg_remove_short_linear_paths(self._h, int(min_length), ptr(pr), C.byref(n), None))
            return n.value
        ids = np.empty(self.graph_sizes()[0], np.int32)
        check(_ffi.lib.amg_remove_short_linear_paths(self._h, int(min_length), ptr(pr),
                                                     C.byref(n), ptr(ids)))
        return ids[: n.value].copy()

    def remove_low_coverage_components(self, min_cov):
        check(_ffi.lib.amg_remove_low_coverage_components(self._h, int(min_cov)))

    def correct_reads(self):
        nr, nt = C.c_int64(0), C.c_int64(0)
        check(_ffi.lib.amg_correct_reads(self._h, C.byref(nr), C.byref(nt)))
        return nr.value, nt.value

    def corrected(self, n_reads, n_tokens, with_positions, buf=None, pos32=False):
        """the corrected set on the host.  pos32: the positions as int32 arrays (gathered on the device, half the bytes
        over PCIe) when every one of them fits — int64 otherwise, as without the flag"""
        out = {"tokens": self._take(buf, "c_tokens", (n_tokens,), np.int32),
               "read_offsets": self._take(buf, "c_read_offsets", (n_reads + 1,), np.int64),
               "orig_read": self._take(buf, "c_orig_read", (n_reads,), np.int32),
               "changed": self._take(buf, "c_changed", (n_reads,), np.uint8)}
        gs = ge = None
        if with_positions and pos32:
            gs = self._take(buf, "c_gene_start32", (n_tokens,), np.int32)
            ge = self._take(buf, "c_gene_end32", (n_tokens,), np.int32)
            check(_ffi.lib.amg_get_corrected(self._h, ptr(out["tokens"]), ptr(out["read_offsets"]),
                                             ptr(out["orig_read"]), ptr(out["changed"]), None, None))
            rc = _ffi.lib.amg_get_corrected_positions32(self._h, ptr(gs), ptr(ge))
            if rc == 0:
                out["gene_start"], out["gene_end"] = gs, ge
                return out
            if rc != -2:   # (AMG_E_ARG: a position beyond 32 bits — the 64-bit arrays below)
                check(rc)
        if with_positions:
            gs = self._take(buf, "c_gene_start", (n_tokens,), np.int64)
            ge = self._take(buf, "c_gene_end", (n_tokens,), np.int64)
        check(_ffi.lib.amg_get_corrected(self._h, ptr(out["tokens"]), ptr(out["read_offsets"]),
                                         ptr(out["orig_read"]), ptr(out["changed"]), ptr(gs), ptr(ge)))
        out["gene_start"], out["gene_end"] = gs, ge
        return out

    def corrected32(self, n_reads, n_tokens, buf=None):
        """the corrected set with 32-bit positions, only the NEW ones moved (amg_get_corrected32): pos_src[i] >= 0 — read
        i's positions are the caller's own arrays from that index on; < 0 — new_start / new_end from -1 - pos_src[i] on"""
        out = {"tokens": self._take(buf, "c_tokens", (n_tokens,), np.int32),
               "read_offsets": self._take(buf, "c_read_offsets", (n_reads + 1,), np.int64),
               "orig_read": self._take(buf, "c_orig_read", (n_reads,), np.int32),
               "changed": self._take(buf, "c_changed", (n_reads,), np.uint8),
               "pos_src": self._take(buf, "c_pos_src", (n_reads,), np.int64)}
        ns = self._take(buf, "c_new_start", (n_tokens,), np.int32)
        ne = self._take(buf, "c_new_end", (n_tokens,), np.int32)
        n = C.c_int64(0)
        check(_ffi.lib.amg_get_corrected32(self._h, ptr(out["tokens"]), ptr(out["read_offsets"]), ptr(out["orig_read"]),
                                           ptr(out["changed"]), ptr(out["pos_src"]), ptr(ns), ptr(ne), C.byref(n)))
        out["new_start"], out["new_end"] = ns[: n.value], ne[: n.value]
        return out

    def corrected_index(self, n_reads):
        """the small arrays of a corrected set only: read offsets, the read each one was, whether it changed"""
        out = {"read_offsets": np.empty(n_reads + 1, np.int64), "orig_read": np.empty(n_reads, np.int32),
               "changed": np.empty(n_reads, np.uint8)}
        check(_ffi.lib.amg_get_corrected(self._h, None, ptr(out["read_offsets"]), ptr(out["orig_read"]),
                                         ptr(out["changed"]), None, None))
        return out

    def adopt_corrected(self):
        check(_ffi.lib.amg_adopt_corrected(self._h))

    def set_reads_from_corrected(self, src):
        """this engine's read set := the corrected set of engine `src` (genes, offsets, positions, read lengths),
        device to device"""
        check(_ffi.lib.amg_set_reads_from_corrected(self._h, src._h))

    # ---- K6: batched exact sub-list search
    def match_patterns(self, which, patterns):
        """patterns: list of int lists (tokens for which=0, node ids for which=1).
        Returns (offsets[n_pat + 1], hit_read, hit_pos), hits ordered by pattern, read, position."""
        n = len(patterns)
        offs = np.zeros(n + 1, np.int64)
        np.cumsum([len(p) for p in patterns], out=offs[1:])
        flat = np.fromiter((x for p in patterns for x in p), dtype=np.int32, count=int(offs[-1]))
        hit_off = np.zeros(n + 1, np.int64)
        check(_ffi.lib.amg_match_patterns(self._h, int(which), ptr(flat) if len(flat) else None,
                                          ptr(offs), n, ptr(hit_off), None, None))
        total = int(hit_off[-1])
        hr, hp = np.empty(total, np.int32), np.empty(total, np.int32)
        if total:
            check(_ffi.lib.amg_match_patterns(self._h, int(which), ptr(flat) if len(flat) else None,
                                              ptr(offs), n, ptr(hit_off), ptr(hr), ptr(hp)))
        return hit_off, hr, hp

    # ---- scaled MinHash of nucleotide segments (bubble popping)
    def minhash(self, segments, set_ids, ksize, scaled):
        """segments: list of str / bytes; set_ids: sketch id per segment.  Returns {sketch id: set of
        hashes} with sourmash's definition (MinHash(n=0, ksize, scaled).add_sequence(seg, force=True))."""
        out = {int(s): set() for s in set_ids}
        if not segments:
            return out
        blobs = [s.encode() if isinstance(s, str) else bytes(s) for s in segments]
        offs = np.zeros(len(blobs) + 1, np.int64)
        np.cumsum([len(b) for b in blobs], out=offs[1:])
        bases = np.frombuffer(b"".join(blobs), dtype=np.uint8)
        if len(bases) == 0:
            return out
        sets = np.ascontiguousarray(set_ids, dtype=np.int32)
        n = C.c_int64(0)
        # one pass: room for every k-mer start when all hashes are kept (scaled 1), for four times the expected share
        # otherwise; hashed a second time only if that was not enough
        cap = len(bases) if int(scaled) <= 1 else min(len(bases), 4 * len(bases) // int(scaled) + 1024)
        while True:
            o_set, o_hash = np.empty(cap, np.int32), np.empty(cap, np.uint64)
            check(_ffi.lib.amg_minhash(self._h, ptr(bases), ptr(offs), ptr(sets), len(blobs), int(ksize), int(scaled),
                                       ptr(o_set), ptr(o_hash), cap, C.byref(n)))
            if n.value <= cap:
                break
            cap = n.value
        if n.value == 0:
            return out
        o_set, o_hash = o_set[: n.value], o_hash[: n.value]
        order = np.lexsort((o_hash, o_set))
        o_set, o_hash = o_set[order], o_hash[order]
        cuts = np.flatnonzero(np.diff(o_set)) + 1
        for ids, hs in zip(np.split(o_set, cuts), np.split(o_hash, cuts)):
            out[int(ids[0])] = set(np.unique(hs).tolist())
        return out

    # ---- bubble popping on device ids (include/amg.h: amg_junction_paths, amg_path_sketch_overlaps)
    def junction_paths(self, max_distance):
        """the junctions of the live graph and every path between two of them that has a rival, in the order the
        reference's double loop over the junctions meets them (amg_junction_paths).  Returns a dict of arrays:
        junction_node, junction_dir, path_start (index of the start junction), path_off, path_node, path_dir, and
        `flags` (non-zero: do not use the paths — see include/amg.h)"""
        sizes = (C.c_int64 * 4)()
        check(_ffi.lib.amg_junction_paths(self._h, int(max_distance), sizes))
        J, P, N, flags = (int(x) for x in sizes)
        out = {"junction_node": np.empty(J, np.int32), "junction_dir": np.empty(J, np.int8),
               "path_start": np.empty(P, np.int32), "path_off": np.zeros(P + 1, np.int64),
               "path_node": np.empty(N, np.int32), "path_dir": np.empty(N, np.int8), "flags": flags}
        check(_ffi.lib.amg_get_junction_paths(self._h, ptr(out["junction_node"]), ptr(out["junction_dir"]),
                                              ptr(out["path_start"]), ptr(out["path_off"]), ptr(out["path_node"]),
                                              ptr(out["path_dir"])))
        return out

    def path_sketch_overlaps(self, seqs, row_to_seq, ksize, scaled, path_off, path_node, pair_a, pair_b):
        """sizes of the paths' sketches and the number of hashes each listed pair of paths shares
        (amg_path_sketch_overlaps); seqs: a Sequences on this engine's device"""
        path_off = np.ascontiguousarray(path_off, np.int64)
        path_node = np.ascontiguousarray(path_node, np.int32)
        pair_a, pair_b = np.ascontiguousarray(pair_a, np.int32), np.ascontiguousarray(pair_b, np.int32)
        rts = None if row_to_seq is None else np.ascontiguousarray(row_to_seq, np.int32)
        n_paths, n_pairs = len(path_off) - 1, len(pair_a)
        size, common = np.zeros(max(n_paths, 1), np.int64), np.zeros(max(n_pairs, 1), np.int64)
        check(_ffi.lib.amg_path_sketch_overlaps(self._h, seqs._h, ptr(rts), int(ksize), int(scaled), n_paths, ptr(path_off),
                                                ptr(path_node), n_pairs, ptr(pair_a), ptr(pair_b), ptr(size), ptr(common)))
        return size[:n_paths], common[:n_pairs]

    # ---- multi-GPU: read shards + key-owner table merge (include/amg.h; drivers in amira_amd/dist.py)
    @staticmethod
    def dist_unique_id():
        """128 bytes naming an RCCL communicator: make them on one rank, hand them to all, then dist_init"""
        buf = C.create_string_buffer(_ffi.UNIQUE_ID_BYTES)
        check(_ffi.lib.amg_dist_unique_id(buf, _ffi.UNIQUE_ID_BYTES))
        return buf.raw

    def dist_init(self, unique_id, rank, world):
        """this engine becomes rank `rank` of `world` over RCCL (one engine per process and GPU)"""
        assert len(unique_id) == _ffi.UNIQUE_ID_BYTES
        check(_ffi.lib.amg_dist_init(self._h, C.c_char_p(unique_id), int(rank), int(world)))

    def dist_init_external(self, rank, world):
        """rank `rank` of `world` with the exchanges performed by the caller (dist_merge_begin / dist_merge_next)"""
        check(_ffi.lib.amg_dist_init_external(self._h, int(rank), int(world)))

    def dist_merge(self, k, min_node_cov=1, min_edge_cov=1):
        """collective: the merged build of all ranks' reads (thresholds > 1 fuse filter_graph into it)"""
        check(_ffi.lib.amg_dist_merge(self._h, int(k), max(int(min_node_cov), 1), max(int(min_edge_cov), 1)))

    def dist_merge_begin(self, k, min_node_cov=1, min_edge_cov=1):
        check(_ffi.lib.amg_dist_merge_begin(self._h, int(k), max(int(min_node_cov), 1), max(int(min_edge_cov), 1)))

    def dist_merge_next(self):
        """None when the build is complete, else the exchange to perform before the next call (an _ffi.Xfer)"""
        x = _ffi.Xfer()
        rc = _ffi.lib.amg_dist_merge_next(self._h, C.byref(x))
        if rc < 0:
            check(rc)
        return x if rc == 1 else None

    @staticmethod
    def dist_merge_local(engines, k, min_node_cov=1, min_edge_cov=1):
        """len(engines) EMULATED ranks on one device (rank r = engines[r]): the device phases of the N-GPU run, the
        exchanges as device copies"""
        n = len(engines)
        handles = (C.c_void_p * n)(*[e._h for e in engines])
        check(_ffi.lib.amg_dist_merge_local(handles, n, int(k), max(int(min_node_cov), 1), max(int(min_edge_cov), 1)))

    def dist_finalize(self):
        check(_ffi.lib.amg_dist_finalize(self._h))

    def copy_d2h(self, dev_ptr, host_array):
        check(_ffi.lib.amg_copy_d2h(self._h, C.c_void_p(dev_ptr), ptr(host_array), host_array.nbytes))

    def copy_h2d(self, dev_ptr, host_array):
        check(_ffi.lib.amg_copy_h2d(self._h, C.c_void_p(dev_ptr), ptr(host_array), host_array.nbytes))

    def dist_stats(self, reset=True):
        out = (C.c_int64 * 8)()
        check(_ffi.lib.amg_dist_stats(self._h, out, 1 if reset else 0))
        return {"host_waits": out[0], "exchanges": out[1], "a2a_bytes_per_peer": out[2], "back_bytes_per_peer": out[3],
                "ag_bytes_contributed": out[4], "repeated_builds": out[5], "a2a_bytes_sent": out[6],
                "derived_builds": out[7]}

    def dist_phase_ms(self, on=True):
        """{phase: synchronised wall ms} of the merge driver since the last call; on: keep measuring"""
        names = (C.c_char_p * 32)()
        ms = (C.c_double * 32)()
        n = _ffi.lib.amg_dist_phase_ms(self._h, 1 if on else 0, names, ms, 32)
        return {names[i].decode(): float(ms[i]) for i in range(max(n, 0))}


# Engines (a HIP stream + grow-only device buffers each) are pooled per device: the reference's drivers build graph
# after graph (three per cleaning iteration, seven in choose_kmer_size), and an engine that has already sized its
# buffers for the read set makes the next build allocation-free.
_ENGINE_POOL = {}


def acquire_engine(device):
    free = _ENGINE_POOL.setdefault(device, [])
    while free:
        engine = free.pop()
        if engine._h:
            return engine
    return Engine(device)


def release_engine(engine):
    """back to the pool — at once, or, while the output of a correct_reads still lives in its buffers, when the last
    such output has been fetched or dropped (lease_done)"""
    if engine is None or not engine._h:
        return
    if len(engine._leases) > 0:
        engine._pool_when_free = True
        return
    engine._pool_when_free = False
    _ENGINE_POOL.setdefault(engine.device, []).append(engine)


def lease_done(engine, lease):
    engine._leases.discard(lease)
    if engine._pool_when_free and len(engine._leases) == 0:
        release_engine(engine)


class Sequences:
    """the reads' nucleotide sequences resident on a device (amg_seqs_create): one upload serves every graph of a
    cleaning run.  `sequences`: list of str (ASCII) or bytes, kept alive by the caller during the call only."""

    def __init__(self, sequences, device=0):
        n = len(sequences)
        ptrs = (C.c_void_p * max(n, 1))()
        lens = np.zeros(max(n, 1), np.int64)
        as_utf8 = C.pythonapi.PyUnicode_AsUTF8AndSize
        as_utf8.restype, as_utf8.argtypes = C.c_void_p, [C.py_object, C.POINTER(C.c_ssize_t)]
        size = C.c_ssize_t(0)
        keep = []
        for i, s in enumerate(sequences):
            if isinstance(s, str):
                ptrs[i] = as_utf8(s, C.byref(size))   # (the string's own bytes for ASCII text: nothing is copied)
                lens[i] = size.value
                if size.value != len(s):   # (positions index characters: a byte per character or nothing)
                    raise TypeError("a read's sequence holds characters outside ASCII")
            else:
                b = bytes(s)
                keep.append(b)
                ptrs[i] = C.cast(C.c_char_p(b), C.c_void_p).value
                lens[i] = len(b)
        self._h = C.c_void_p()
        self.device, self.n = int(device), n
        check(_ffi.lib.amg_seqs_create(int(device), ptrs, ptr(lens), n, C.byref(self._h)))
        del keep

    def close(self):
        if getattr(self, "_h", None):
            _ffi.lib.amg_seqs_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass
