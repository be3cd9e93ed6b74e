"""world_size-2 gloo tests (CPU) of the N > 1 exchange plumbing of amira_amd.dist: the exchanges libamg's merge
driver asks for (include/amg.h amg_xfer: a variable all-to-all of whole records, an equal-size all-gather) performed by
`perform_host` between real processes must deliver exactly what the descriptor says — rank r's block for rank d lands
behind the blocks of the ranks below r, whatever the counts (empty blocks included).  The device side of the merge (and
its failure hand-shake, which lives in libamg) is tested on the GPU: tests/test_gpu_dist.py."""
import ctypes as C
import os

import numpy as np
import torch.multiprocessing as mp

REC = 24


class _HostEngine:
    """stands in for an Engine whose "device" memory is host memory: perform_host only ever asks an engine to copy"""

    @staticmethod
    def copy_d2h(dev_ptr, host_array):
        if host_array.nbytes:
            C.memmove(host_array.ctypes.data, dev_ptr, host_array.nbytes)

    @staticmethod
    def copy_h2d(dev_ptr, host_array):
        if host_array.nbytes:
            C.memmove(dev_ptr, host_array.ctypes.data, host_array.nbytes)


def _payload(rank, world):
    rng = np.random.default_rng(100 + rank)
    counts = [int(x) for x in rng.integers(0, 7, world)]
    if rank == 1:
        counts[0] = 0  # an empty block
    data = rng.integers(0, 256, sum(counts) * REC, dtype=np.uint8)
    return counts, data


def _xfer(kind, elem_bytes, send, recv, send_counts=None, recv_counts=None, count=0):
    from amira_amd import _ffi
    x = _ffi.Xfer()
    x.kind, x.elem_bytes, x.count = kind, elem_bytes, count
    x.send, x.recv = send.ctypes.data, recv.ctypes.data
    keep = [send, recv]
    if send_counts is not None:
        sc, rc = np.asarray(send_counts, np.int64), np.asarray(recv_counts, np.int64)
        x.send_counts = sc.ctypes.data_as(C.POINTER(C.c_int64))
        x.recv_counts = rc.ctypes.data_as(C.POINTER(C.c_int64))
        keep += [sc, rc]
    return x, keep


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1")
    import torch.distributed as dist
    # (a file as the rendezvous: a TCP port picked by the parent can be taken by the time the store listens on it)
    dist.init_process_group("gloo", init_method=f"file://{port}", rank=rank, world_size=world)
    from amira_amd import _ffi
    from amira_amd.dist import perform_host
    eng = _HostEngine()
    counts, data = _payload(rank, world)
    # the count messages first, as the merge driver does: four words per peer
    msg = np.zeros(world * 4, np.int64)
    msg[0::4] = counts
    msg[1::4] = 1000 + rank
    got_msg = np.zeros(world * 4, np.int64)
    x, keep = _xfer(_ffi.XFER_ALL_TO_ALL, 32, msg, got_msg, [1] * world, [1] * world)
    perform_host(eng, x)
    recv_counts = [int(v) for v in got_msg[0::4]]
    # the records
    recv = np.zeros(max(sum(recv_counts), 1) * REC, np.uint8)
    x, keep = _xfer(_ffi.XFER_ALL_TO_ALL, REC, data if len(data) else np.zeros(1, np.uint8), recv, counts, recv_counts)
    perform_host(eng, x)
    # one 16-byte reply per received record travels back: {1000 * owner + index at the owner, 7}
    n = sum(recv_counts)
    replies = np.zeros(max(n, 1) * 2, np.int64)
    replies[0:2 * n:2] = np.arange(n) + 1000 * rank
    replies[1:2 * n:2] = 7
    mine = np.zeros(max(sum(counts), 1) * 2, np.int64)
    x, keep = _xfer(_ffi.XFER_ALL_TO_ALL, 16, replies, mine, recv_counts, counts)
    perform_host(eng, x)
    # an equal-size all-gather of m records
    m = 3
    held = np.full(m * REC, rank + 1, np.uint8)
    everything = np.zeros(world * m * REC, np.uint8)
    x, keep = _xfer(_ffi.XFER_ALL_GATHER, REC, held, everything, count=m)
    perform_host(eng, x)
    q.put((rank, got_msg.copy(), recv[: n * REC].copy(), recv_counts, mine[: 2 * sum(counts)].copy(), everything.copy()))
    dist.barrier()
    dist.destroy_process_group()


def test_exchanges_world2_gloo(tmp_path):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = str(tmp_path / "rendezvous")
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(world):
        r, msg, recv, recv_counts, mine, everything = q.get(timeout=120)
        got[r] = (msg, recv, recv_counts, mine, everything)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    payload = [_payload(r, world) for r in range(world)]
    for dst in range(world):
        # count messages: block src of rank dst = what src addressed to dst
        assert got[dst][2] == [payload[src][0][dst] for src in range(world)]
        assert got[dst][0][1::4].tolist() == [1000 + src for src in range(world)]
        parts = []
        for src in range(world):
            counts, data = payload[src]
            off = sum(counts[:dst]) * REC
            parts.append(data[off: off + counts[dst] * REC])
        assert np.array_equal(got[dst][1], np.concatenate(parts))
    # replies: rank r sent counts[d] records to owner d, which received them after those of the ranks below r
    for r in range(world):
        counts = payload[r][0]
        want = []
        for d in range(world):
            before = sum(got[d][2][:r])
            want += [1000 * d + before + i for i in range(counts[d])]
        assert got[r][3][0::2].tolist() == want
        assert (got[r][3][1::2] == 7).all()
        m = 3
        assert np.array_equal(got[r][4], np.repeat(np.arange(1, world + 1, dtype=np.uint8), m * REC))
