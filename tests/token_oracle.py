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
