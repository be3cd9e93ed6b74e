"""Multi-GPU build: read shards + key-owner table merge, the reference's
build_multiprocessed_graph + merge_graphs (graph_utils.py:94-124) as ONE graph on every rank.

The merge lives behind the C ABI (include/amg.h "multi-GPU"; amira_amd/csrc/amg_dist.hip): libamg runs the device
phases AND the exchanges between them — RCCL all-to-alls to the key owners and back, an all-gather of the records each
rank holds — on the engine's own stream, with two host waits per kind of record.  What is left here:

  dist_build           one process per GPU under torch.distributed.  Backend "nccl" (= RCCL): the engine gets its own
                       communicator (the unique id travels through the process group once) and `amg_dist_merge` does the
                       rest.  Backend "gloo" (functional tests: ranks that share one GPU, no RCCL): the exchanges libamg
                       asks for (`amg_dist_merge_next`) are performed here, staged through the host.
  dist_build_loopback  W emulated ranks in ONE process on one GPU (`amg_dist_merge_local`): tests, tools/scaling_model.py.
"""
import numpy as np

from . import _ffi
from .engine import Engine


def _group_info(group):
    import torch.distributed as dist
    return dist.get_world_size(group), dist.get_rank(group), dist.get_backend(group)


def ensure_transport(engine, group=None):
    """the engine as a rank of `group`: "rccl" — its own RCCL communicator, made once per (engine, group) — or "host"
    (gloo: the caller of libamg performs the exchanges)"""
    import torch.distributed as dist
    world, rank, backend = _group_info(group)
    mode = "host" if backend == "gloo" else "rccl"
    have = getattr(engine, "_dist", None)
    if have == (world, rank, mode, id(group)):
        return mode
    if mode == "rccl":
        # the 128 bytes that name the communicator: made on the group's first rank, handed round by the group itself
        box = [Engine.dist_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        engine.dist_init(box[0], rank, world)
    else:
        engine.dist_init_external(rank, world)
    engine._dist = (world, rank, mode, id(group))
    return mode


def perform_host(engine, x, group=None):
    """one exchange of a merged build (an _ffi.Xfer: device pointers, host counts) over a transport that moves HOST
    memory (gloo): device -> host, the collective, host -> device"""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    eb = int(x.elem_bytes)
    if x.kind == _ffi.XFER_ALL_TO_ALL:
        sc = [int(x.send_counts[p]) * eb for p in range(world)]
        rc = [int(x.recv_counts[p]) * eb for p in range(world)]
        send = np.empty(sum(sc), np.uint8)
        engine.copy_d2h(x.send, send)
        recv = torch.empty(sum(rc), dtype=torch.uint8)
        dist.all_to_all_single(recv, torch.from_numpy(send), rc, sc, group=group)
        engine.copy_h2d(x.recv, recv.numpy())
    elif x.kind == _ffi.XFER_ALL_GATHER:
        n = int(x.count) * eb
        send = np.empty(n, np.uint8)
        engine.copy_d2h(x.send, send)
        recv = torch.empty(world * n, dtype=torch.uint8)
        dist.all_gather_into_tensor(recv, torch.from_numpy(send), group=group)
        engine.copy_h2d(x.recv, recv.numpy())
    else:
        raise ValueError(f"unknown exchange kind {x.kind}")


def dist_build(engine, k, group=None, min_node_cov=1, min_edge_cov=1):
    """Collective: call on every rank with its own engine (reads already set, rank r holding the reads after rank
    r - 1's).  min_node_cov / min_edge_cov > 1 fuse filter_graph into the merge.  Failures of any rank reach every rank
    (AmgError); a merge-key collision makes all ranks repeat the build with the next seed, inside libamg."""
    if ensure_transport(engine, group) == "rccl":
        engine.dist_merge(k, min_node_cov, min_edge_cov)
        return
    engine.dist_merge_begin(k, min_node_cov, min_edge_cov)
    while True:
        x = engine.dist_merge_next()
        if x is None:
            return
        perform_host(engine, x, group)


def dist_build_loopback(engines, k, min_node_cov=1, min_edge_cov=1):
    """Emulate len(engines) ranks in ONE process (tests on a single GPU, the scaling model): the device phases are
    exactly those of dist_build, the exchanges device copies."""
    Engine.dist_merge_local(engines, k, min_node_cov, min_edge_cov)
