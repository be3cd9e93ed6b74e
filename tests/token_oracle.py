"""ctypes wrapper of oracle/libtoken_oracle.so (test infrastructure only)."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "oracle", "libtoken_oracle.so")


class TokGraph(C.Structure):
    _fields_ = [("n_nodes", C.c_int64), ("n_edges", C.c_int64), ("n_windows", C.c_int64),
                ("n_short", C.c_int64), ("node_tokens", C.POINTER(C.c_int32)),
                ("node_cov", C.POINTER(C.c_uint32)), ("node_first_dir", C.POINTER(C.c_int8)),
                ("edge_src", C.POINTER(C.c_int32)), ("edge_tgt", C.POINTER(C.c_int32)),
                ("edge_sdir", C.POINTER(C.c_int8)), ("edge_tdir", C.POINTER(C.c_int8)),
                ("edge_cov", C.POINTER(C.c_uint32))]


def _lib():
    if not os.path.exists(SO):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
    lib = C.CDLL(SO)
    lib.token_oracle_build.restype = C.c_int
    lib.token_oracle_build.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32,
                                       C.c_void_p, C.c_void_p, C.POINTER(TokGraph)]
    lib.token_oracle_free.argtypes = [C.POINTER(TokGraph)]
    return lib


def build(tokens, read_off, k, two_v):
    """Sequential C build in token space -> dict of numpy arrays (engine conventions)."""
    lib = _lib()
    tokens = np.ascontiguousarray(tokens, np.int32)
    read_off = np.ascontiguousarray(read_off, np.int64)
    T = int(read_off[-1])
    tok_node, tok_dir = np.empty(T, np.int32), np.empty(T, np.int8)
    g = TokGraph()
    rc = lib.token_oracle_build(tokens.ctypes.data, read_off.ctypes.data, len(read_off) - 1, k, two_v,
                                tok_node.ctypes.data, tok_dir.ctypes.data, C.byref(g))
    if rc != 0:
        raise AssertionError("Gene-mer and reverse complement gene-mer are identical")
    D, E = g.n_nodes, g.n_edges
    as_np = np.ctypeslib.as_array
    out = {
        "n_windows": g.n_windows, "n_short": g.n_short,
        "tokens": as_np(g.node_tokens, (D * k,)).reshape(D, k).copy() if D else np.empty((0, k), np.int32),
        "coverage": as_np(g.node_cov, (D,)).copy() if D else np.empty(0, np.uint32),
        "first_dir": as_np(g.node_first_dir, (D,)).copy() if D else np.empty(0, np.int8),
        "src": as_np(g.edge_src, (E,)).copy() if E else np.empty(0, np.int32),
        "tgt": as_np(g.edge_tgt, (E,)).copy() if E else np.empty(0, np.int32),
        "sdir": as_np(g.edge_sdir, (E,)).copy() if E else np.empty(0, np.int8),
        "tdir": as_np(g.edge_tdir, (E,)).copy() if E else np.empty(0, np.int8),
        "ecov": as_np(g.edge_cov, (E,)).copy() if E else np.empty(0, np.uint32),
        "tok_node": tok_node, "tok_dir": tok_dir,
    }
    lib.token_oracle_free(C.byref(g))
    return out


# ---------------------------------------------------------------------------------------------
# oracle/token_sweep.c: the whole cleaning sweep in token space (stateful)
def _sweep_lib():
    lib = _lib()
    if getattr(lib, "_tsw_ready", False):
        return lib
    P, I64, I32 = C.c_void_p, C.c_int64, C.c_int32
    lib.tsw_new.restype = P
    lib.tsw_new.argtypes = [P, P, I64, I32, P, P, P]
    lib.tsw_free.argtypes = [P]
    lib.tsw_build.restype = C.c_int
    lib.tsw_build.argtypes = [P, I32]
    lib.tsw_filter.argtypes = [P, I64, I64]
    lib.tsw_clip.restype = I64
    lib.tsw_clip.argtypes = [P, I32, P, P]
    lib.tsw_remove_low_coverage_components.argtypes = [P, I64]
    lib.tsw_correct.argtypes = [P, P, P]
    lib.tsw_corrected.restype = C.c_int
    lib.tsw_corrected.argtypes = [P] * 7
    lib.tsw_adopt.restype = C.c_int
    lib.tsw_adopt.argtypes = [P]
    lib.tsw_counts.argtypes = [P, P]
    lib.tsw_nodes.argtypes = [P] * 6
    lib.tsw_edges.argtypes = [P] * 7
    lib.tsw_read_nodes.argtypes = [P] * 4
    lib.tsw_adj.argtypes = [P] * 3
    lib.tsw_set_threads.argtypes = [C.c_int]
    lib._tsw_ready = True
    return lib


class Sweep:
    """Stateful C oracle with the engine's array conventions (amira_amd.Engine method names)."""
    COUNT_KEYS = ("n_reads", "n_tokens", "n_windows", "n_short_reads", "n_nodes", "n_edges", "n_components",
                  "n_live_nodes", "n_live_edges", "n_reads_to_correct")

    def __init__(self, tokens, read_off, two_v, gene_start=None, gene_end=None, read_len=None):
        self.lib = _sweep_lib()
        tokens = np.ascontiguousarray(tokens, np.int32)
        read_off = np.ascontiguousarray(read_off, np.int64)
        ptr = lambda a: None if a is None else np.ascontiguousarray(a, np.int64).ctypes.data
        keep = [None if a is None else np.ascontiguousarray(a, np.int64) for a in (gene_start, gene_end, read_len)]
        self.h = self.lib.tsw_new(tokens.ctypes.data, read_off.ctypes.data, len(read_off) - 1, two_v,
                                  *[None if a is None else a.ctypes.data for a in keep])
        self.have_pos = gene_start is not None
        self.k = None

    def close(self):
        if self.h:
            self.lib.tsw_free(self.h)
            self.h = None

    __del__ = close

    def build(self, k):
        if self.lib.tsw_build(self.h, k) != 0:
            raise AssertionError("Gene-mer and reverse complement gene-mer are identical")
        self.k = k

    def counts(self):
        out = np.zeros(10, np.int64)
        self.lib.tsw_counts(self.h, out.ctypes.data)
        return dict(zip(self.COUNT_KEYS, out.tolist()))

    def nodes(self):
        c = self.counts()
        D = c["n_nodes"]
        o = {"tokens": np.empty((D, self.k), np.int32), "coverage": np.empty(D, np.uint32),
             "first_dir": np.empty(D, np.int8), "component": np.empty(D, np.int32), "alive": np.empty(D, np.uint8)}
        self.lib.tsw_nodes(self.h, *[o[x].ctypes.data for x in ("tokens", "coverage", "first_dir", "component", "alive")])
        return o

    def edges(self):
        E = self.counts()["n_edges"]
        o = {"src": np.empty(E, np.int32), "tgt": np.empty(E, np.int32), "sdir": np.empty(E, np.int8),
             "tdir": np.empty(E, np.int8), "coverage": np.empty(E, np.uint32), "alive": np.empty(E, np.uint8)}
        self.lib.tsw_edges(self.h, *[o[x].ctypes.data for x in ("src", "tgt", "sdir", "tdir", "coverage", "alive")])
        return o

    def read_nodes(self):
        c = self.counts()
        tn, td = np.empty(c["n_tokens"], np.int32), np.empty(c["n_tokens"], np.int8)
        self.lib.tsw_read_nodes(self.h, tn.ctypes.data, td.ctypes.data, None)
        return tn, td

    def reads_to_correct(self):
        fix = np.empty(self.counts()["n_reads"], np.uint8)
        self.lib.tsw_read_nodes(self.h, None, None, fix.ctypes.data)
        return fix

    def node_adj(self):
        c = self.counts()
        off, ids = np.empty(2 * c["n_nodes"] + 1, np.int64), np.empty(c["n_edges"], np.int32)
        self.lib.tsw_adj(self.h, off.ctypes.data, ids.ctypes.data)
        return off, ids

    def filter(self, min_node_cov, min_edge_cov):
        self.lib.tsw_filter(self.h, min_node_cov, min_edge_cov)

    def remove_short_linear_paths(self, min_length, protect=None):
        out = np.empty(self.counts()["n_nodes"], np.int32)
        p = None if protect is None else np.ascontiguousarray(protect, np.uint8)
        n = self.lib.tsw_clip(self.h, min_length, None if p is None else p.ctypes.data, out.ctypes.data)
        return out[:n].copy()

    def remove_low_coverage_components(self, m):
        self.lib.tsw_remove_low_coverage_components(self.h, m)

    def correct_reads(self, threads=1):
        """threads > 1: the reads in that many chunks side by side (every read is corrected on its own; the outputs are
        laid end to end) — the full-size tests; 1: the sequential restatement (what bench.py times as cpu_baseline)"""
        a, b = C.c_int64(), C.c_int64()
        self.lib.tsw_set_threads(int(threads))
        try:
            self.lib.tsw_correct(self.h, C.byref(a), C.byref(b))
        finally:
            self.lib.tsw_set_threads(1)
        return a.value, b.value

    def corrected(self, n_reads, n_tokens, with_pos=False):
        o = {"tokens": np.empty(n_tokens, np.int32), "read_offsets": np.empty(n_reads + 1, np.int64),
             "orig_read": np.empty(n_reads, np.int32), "changed": np.empty(n_reads, np.uint8)}
        if with_pos:
            o["gene_start"], o["gene_end"] = np.empty(n_tokens, np.int64), np.empty(n_tokens, np.int64)
        rc = self.lib.tsw_corrected(self.h, o["tokens"].ctypes.data, o["read_offsets"].ctypes.data,
                                    o["orig_read"].ctypes.data, o["changed"].ctypes.data,
                                    o["gene_start"].ctypes.data if with_pos else None,
                                    o["gene_end"].ctypes.data if with_pos else None)
        assert rc == 0
        return o

    def adopt_corrected(self):
        assert self.lib.tsw_adopt(self.h) == 0


class SweepEngine:
    """The C sweep oracle behind the method names of amira_amd.Engine, so that the procedures
    written for the engine (tests/test_gpu_sweep.py run_sweep, tests/helpers.py) can drive either."""

    def __init__(self):
        self.s = None
        self._reads = self._pos = None

    def close(self):
        if self.s:
            self.s.close()
            self.s = None

    def set_reads(self, tokens, read_offsets, two_v):
        self.close()
        self._reads = (np.array(tokens, np.int32), np.array(read_offsets, np.int64), int(two_v))
        self._pos = None

    def set_positions(self, gene_start, gene_end, read_len=None):
        self._pos = (gene_start, gene_end, read_len)

    def build(self, k):
        if self.s is None:
            pos = self._pos or (None, None, None)
            self.s = Sweep(*self._reads, *pos)
        self.s.build(k)
        self._node_reads = None

    def counts(self):
        c = self.s.counts()
        c["k"] = self.s.k
        return c

    def nodes(self):
        return self.s.nodes()

    def edges(self):
        return self.s.edges()

    def read_nodes(self):
        return self.s.read_nodes()

    def node_adj(self):
        return self.s.node_adj()

    def reads_to_correct(self):
        return self.s.reads_to_correct()

    def node_reads(self):
        """Node.listOfReads (construct_node.py:64-67): distinct reads in first-appearance order,
        derived from the per-window node ids of the build (kept across removals, as the lists are)"""
        if self._node_reads is None:
            tn, _ = self.s.read_nodes()
            off = self._reads[1] if self.s.counts()["n_reads"] == len(self._reads[1]) - 1 else None
            c = self.s.counts()
            if off is None or int(off[-1]) != c["n_tokens"]:
                raise RuntimeError("node_reads: read offsets of the current read set unknown")
            read_of = np.repeat(np.arange(c["n_reads"], dtype=np.int64), np.diff(off))
            m = tn >= 0
            pairs = np.unique(np.stack([tn[m].astype(np.int64), read_of[m]], 1), axis=0)
            cnt = np.bincount(pairs[:, 0], minlength=c["n_nodes"])
            o = np.zeros(c["n_nodes"] + 1, np.int64)
            np.cumsum(cnt, out=o[1:])
            self._node_reads = (o, pairs[:, 1].astype(np.int32))
        return self._node_reads

    def filter(self, a, b):
        self.s.filter(a, b)

    def remove_short_linear_paths(self, min_length, protect=None):
        return self.s.remove_short_linear_paths(min_length, protect)

    def remove_low_coverage_components(self, m):
        self.s.remove_low_coverage_components(m)

    def correct_reads(self):
        return self.s.correct_reads()

    def corrected(self, n_reads, n_tokens, with_positions):
        o = self.s.corrected(n_reads, n_tokens, bool(with_positions))
        o.setdefault("gene_start", None)
        o.setdefault("gene_end", None)
        self._last_off = o["read_offsets"]
        return o

    def adopt_corrected(self):
        self.s.adopt_corrected()
        c = self.s.counts()
        self._reads = (None, self._last_off, self._reads[2])
