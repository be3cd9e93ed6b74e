"""ctypes binding of libamg.so (include/amg.h).  There is no CPU path: importing this
module without the built library raises, and amg_create raises without a HIP device."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libamg.so")


class AmgError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libamg error {code}: {msg}")
        self.code = code


E_PALINDROME = -4
E_NOMEM = -6
MAX_K = 16


class Counts(C.Structure):
    _fields_ = [(n, C.c_int64) for n in (
        "n_reads", "n_tokens", "n_windows", "n_short_reads", "n_nodes", "n_edges", "n_pairs",
        "n_components", "n_live_nodes", "n_live_edges", "n_reads_to_correct",
        "node_table_slots", "edge_table_slots", "build_retries")] + [("k", C.c_int32), ("two_v", C.c_int32),
                                                                 ("exact_keys", C.c_int32), ("derived", C.c_int32)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


class Xfer(C.Structure):
    """include/amg.h amg_xfer: one exchange of a merged build (device pointers, host count arrays)"""
    _fields_ = [("kind", C.c_int32), ("elem_bytes", C.c_int32), ("send", C.c_void_p), ("recv", C.c_void_p),
                ("send_counts", C.POINTER(C.c_int64)), ("recv_counts", C.POINTER(C.c_int64)), ("count", C.c_int64)]


XFER_ALL_TO_ALL, XFER_ALL_GATHER = 1, 2
UNIQUE_ID_BYTES = 128


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(make -C amira_amd/csrc).  amira_amd has no CPU fallback."
        )
    # PyTorch-ROCm bundles its own libamdhip64; two HIP runtimes in one process do not share
    # the device.  Importing torch first makes the loader reuse the already-loaded runtime
    # (same soname) for libamg.so, whichever order the application imports things in.
    try:
        import torch  # noqa: F401
    except Exception:  # noqa: BLE001 - torch is optional for the single-GPU path
        pass
    lib = C.CDLL(LIB_PATH)
    P, I32, I64, U32 = C.c_void_p, C.c_int32, C.c_int64, C.c_uint32
    sig = {
        "amg_create": (C.c_int, [C.c_int, C.POINTER(P)]),
        "amg_destroy": (C.c_int, [P]),
        "amg_last_error": (C.c_char_p, []),
        "amg_sync": (C.c_int, [P]),
        "amg_stream": (P, [P]),
        "amg_set_reads": (C.c_int, [P, P, P, I64, I32, C.c_int]),
        "amg_set_positions": (C.c_int, [P, P, P, P, C.c_int]),
        "amg_set_positions32": (C.c_int, [P, P, P, P, C.c_int]),
        "amg_set_read_lengths": (C.c_int, [P, P, C.c_int]),
        "amg_build": (C.c_int, [P, I32]),
        "amg_build_filtered": (C.c_int, [P, C.c_int32, C.c_uint32, C.c_uint32]),
        "amg_counts": (C.c_int, [P, C.POINTER(Counts)]),
        "amg_build_multi": (C.c_int, [C.POINTER(P), C.POINTER(I32), I32]),
        "amg_finalize": (C.c_int, [P]),
        "amg_sizes": (C.c_int, [P, C.POINTER(I64), C.POINTER(I64)]),
        "amg_graph_sizes": (C.c_int, [P, C.POINTER(I64), C.POINTER(I64), C.POINTER(I32)]),
        "amg_get_nodes": (C.c_int, [P, P, P, P, P, P, P]),
        "amg_get_edges": (C.c_int, [P, P, P, P, P, P, P]),
        "amg_get_read_nodes": (C.c_int, [P, P, P]),
        "amg_get_read_nodes_rows": (C.c_int, [P, P, P, I64, P]),
        "amg_get_node_adj": (C.c_int, [P, P, P]),
        "amg_get_node_reads": (C.c_int, [P, P, P]),
        "amg_filter": (C.c_int, [P, U32, U32]),
        "amg_remove_nodes": (C.c_int, [P, P, I64]),
        "amg_remove_edges": (C.c_int, [P, P, I64]),
        "amg_remove_short_linear_paths": (C.c_int, [P, I32, P, C.POINTER(I64), P]),
        "amg_remove_low_coverage_components": (C.c_int, [P, U32]),
        "amg_get_reads_to_correct": (C.c_int, [P, P]),
        "amg_correct_reads": (C.c_int, [P, C.POINTER(I64), C.POINTER(I64)]),
        "amg_get_corrected": (C.c_int, [P, P, P, P, P, P, P]),
        "amg_get_corrected32": (C.c_int, [P, P, P, P, P, P, P, P, C.POINTER(I64)]),
        "amg_get_corrected_positions32": (C.c_int, [P, P, P]),
        "amg_adopt_corrected": (C.c_int, [P]),
        "amg_set_reads_from_corrected": (C.c_int, [P, P]),
        "amg_match_patterns": (C.c_int, [P, C.c_int, P, P, I64, P, P, P]),
        "amg_minhash": (C.c_int, [P, P, P, P, I64, I32, C.c_uint64, P, P, I64, C.POINTER(I64)]),
        "amg_junction_paths": (C.c_int, [P, I32, C.POINTER(I64)]),
        "amg_get_junction_paths": (C.c_int, [P, P, P, P, P, P, P]),
        "amg_seqs_create": (C.c_int, [I32, P, P, I64, C.POINTER(P)]),
        "amg_seqs_destroy": (C.c_int, [P]),
        "amg_path_sketch_overlaps": (C.c_int, [P, P, P, I32, C.c_uint64, I64, P, P, I64, P, P, P, P]),
        "amg_nw_align": (C.c_int, [P, I32, P, I32, P, C.POINTER(I32)]),
        "amg_dist_unique_id": (C.c_int, [P, I32]),
        "amg_dist_init": (C.c_int, [P, P, I32, I32]),
        "amg_dist_merge": (C.c_int, [P, I32, U32, U32]),
        "amg_dist_finalize": (C.c_int, [P]),
        "amg_dist_init_external": (C.c_int, [P, I32, I32]),
        "amg_dist_merge_begin": (C.c_int, [P, I32, U32, U32]),
        "amg_dist_merge_next": (C.c_int, [P, C.POINTER(Xfer)]),
        "amg_dist_merge_local": (C.c_int, [C.POINTER(P), I32, I32, U32, U32]),
        "amg_copy_d2h": (C.c_int, [P, P, P, I64]),
        "amg_copy_h2d": (C.c_int, [P, P, P, I64]),
        "amg_dist_stats": (C.c_int, [P, C.POINTER(I64), I32]),
        "amg_dist_phase_ms": (C.c_int, [P, I32, C.POINTER(C.c_char_p), C.POINTER(C.c_double), I32]),
        "amg_calls_load_json": (C.c_int, [C.c_char_p, C.POINTER(P)]),
        "amg_calls_counts": (C.c_int, [P, C.POINTER(I64), C.POINTER(I64), C.POINTER(I64), C.POINTER(I64),
                                       C.POINTER(I64)]),
        "amg_calls_get": (C.c_int, [P, P, P, P, P, P]),
        "amg_calls_load_positions_json": (C.c_int, [P, C.c_char_p, P, P]),
        "amg_calls_write_json": (C.c_int, [C.c_char_p, P, P, I64, P, I64, P]),
        "amg_calls_write_positions_json": (C.c_int, [C.c_char_p, P, P, P, I64, P]),
        "amg_calls_write_positions_json32": (C.c_int, [C.c_char_p, P, P, P, I64, P]),
        "amg_calls_first_use": (C.c_int, [P, I64, I32, P, I64, P]),
        "amg_calls_has_blanks": (C.c_int, [P, C.POINTER(I32)]),
        "amg_calls_free": (C.c_int, [P]),
        "amg_calls_trim": (C.c_int, [C.POINTER(I64)]),
        "amg_cluster_full_blocks": (C.c_int, [P, P, I64, P, P, I32, P, I64, I64, C.POINTER(P)]),
        "amg_cluster_anchor_stats": (C.c_int, [P, P, I64, P, P, I32, I64, P]),
        "amg_cluster_blocks_sizes": (C.c_int, [P, C.POINTER(I64), C.POINTER(I64)]),
        "amg_cluster_blocks_get": (C.c_int, [P, P, P]),
        "amg_cluster_blocks_free": (C.c_int, [P]),
        "amg_py_tuple_hash": (I64, [P, I64]),
        "amg_pyset_script": (C.c_int, [P, I64, P, I32, P, P]),
        "amg_last_timings": (C.c_int, [P, C.POINTER(C.c_char_p), C.POINTER(C.c_float), C.c_int]),
        "amg_set_timing": (C.c_int, [P, C.c_int]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)  # AttributeError here == header / library mismatch
        fn.restype, fn.argtypes = res, args
    return lib, sorted(sig)


lib, SYMBOLS = _load()


def check(rc):
    if rc != 0:
        raise AmgError(rc, lib.amg_last_error().decode(errors="replace"))


def ptr(a):
    """pointer to a C-contiguous numpy array (or None)."""
    if a is None:
        return None
    assert isinstance(a, np.ndarray) and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.c_void_p)
