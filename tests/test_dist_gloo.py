"""world_size-2 gloo test (CPU) of the N > 1 exchange plumbing used by amira_amd.dist: the
variable-size record all-to-all and all-gather must deliver exactly what the single-process
loop-back driver delivers."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

REC = 48


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _payload(rank, world):
    rng = np.random.default_rng(100 + rank)
    counts = [int(x) for x in rng.integers(0, 7, world)]
    if rank == 1:
        counts[0] = 0  # an empty bucket
    data = rng.integers(0, 256, sum(counts) * REC, dtype=np.uint8)
    return counts, data


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from amira_amd.dist import exchange_a2a, exchange_ag
    from amira_amd.dist import exchange_back
    counts, data = _payload(rank, world)
    recv, recv_counts = exchange_a2a(torch.from_numpy(data.copy()), counts, REC)
    n = sum(recv_counts)
    owned = torch.from_numpy(data[: (rank + 2) * REC].copy()) if len(data) >= (rank + 2) * REC else torch.zeros(0, dtype=torch.uint8)
    n_owned = len(owned) // REC
    slots, n_slots, total = exchange_ag(owned if n_owned else torch.zeros(REC, dtype=torch.uint8), n_owned, REC)
    # the padded all-gather: world equal parts of n_slots / world record slots, zero tails
    m = n_slots // world
    parts = slots.numpy().reshape(world, m * REC)
    # one reply per received record travels back to its sender: reply = 1000 * owner + index at the owner
    replies = torch.arange(max(n, 1), dtype=torch.int64) + 1000 * rank
    mine = exchange_back(replies, recv_counts, counts)
    q.put((rank, recv[: n * REC].numpy().copy(), n, parts.copy(), total, n_owned, recv_counts,
           mine[: sum(counts)].numpy().copy()))
    dist.barrier()
    dist.destroy_process_group()


def test_exchange_world2_gloo():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(world):
        r, recv, n, parts, total, n_owned, recv_counts, mine = q.get(timeout=120)
        got[r] = (recv, n, parts, total, n_owned, recv_counts, mine)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    payload = [_payload(r, world) for r in range(world)]
    for dst in range(world):
        parts = []
        for src in range(world):
            counts, data = payload[src]
            off = sum(counts[:dst]) * REC
            parts.append(data[off: off + counts[dst] * REC])
        want = np.concatenate(parts)
        assert got[dst][1] == len(want) // REC
        assert np.array_equal(got[dst][0], want)
    owned = [payload[r][1][: got[r][4] * REC] for r in range(world)]
    for r in range(world):
        assert got[r][3] == sum(got[x][4] for x in range(world))
        for x in range(world):   # rank x's part: its records, then zeros
            part = got[r][2][x]
            assert np.array_equal(part[: len(owned[x])], owned[x])
            assert not part[len(owned[x]):].any()
    # replies: rank r sent counts[d] records to owner d, which received them after those of the ranks below r
    for r in range(world):
        counts = payload[r][0]
        want = []
        for d in range(world):
            before = sum(got[d][5][:r])
            want += [1000 * d + before + i for i in range(counts[d])]
        assert got[r][6].tolist() == want


def _failing_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from amira_amd.dist import PeerFailed, exchange_a2a, exchange_ag
    seen = []
    # rank 1's device phase "failed" before the all-to-all: it signals with negative counts
    try:
        counts = [-1] * world if rank == 1 else [2, 3]
        exchange_a2a(torch.zeros(max(sum(c for c in counts if c > 0), 1) * REC, dtype=torch.uint8), counts, REC)
        seen.append("a2a went through")
    except PeerFailed as e:
        seen.append(("a2a", e.ranks if rank != 1 else "self"))
    # ... and before the all-gather
    try:
        exchange_ag(torch.zeros(REC, dtype=torch.uint8), -1 if rank == 1 else 1, REC)
        seen.append("ag went through")
    except PeerFailed as e:
        seen.append(("ag", e.ranks))
    q.put((rank, seen))
    dist.barrier()
    dist.destroy_process_group()


def test_failed_rank_releases_its_peers_gloo():
    """a rank whose device phase failed takes part in the next count exchange with negative counts: nobody hangs,
    everybody raises (amira_amd.dist.dist_build)"""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_failing_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got[0] == [("a2a", [1]), ("ag", [1])]
    assert got[1] == [("a2a", "self"), ("ag", [1])]


class _FakeEngine:
    """host-side stand-in for an Engine (device None): the device phases are no-ops that may fail as told, so that the
    driver's failure hand-shake and retry logic run between real processes without a GPU"""
    device = None

    def __init__(self, fail):
        self.fail = fail          # {attempt: error code raised by dist_nodes_local}
        self.calls = []

    @staticmethod
    def dist_record_bytes(k):
        return 48, 24

    def sizes(self):
        return 10, 100

    def stream(self):
        return 0

    def dist_set_filter(self, a, b):
        pass

    def dist_nodes_local(self, k, base, total, world, attempt=0):
        self.calls.append(("nodes_local", attempt))
        code = self.fail.get(attempt)
        if code is not None:
            from amira_amd._ffi import AmgError
            raise AmgError(code, "told to fail")
        return [0] * world

    def dist_edges_local(self, world):
        return [0] * world

    def dist_pack(self, what, ptr):
        pass

    def dist_reduce(self, what, recv, n_recv, n_sources, owned, replies=None):
        return 0

    def dist_global(self, what, all_ptr, n_slots, n_total, replies=None):
        pass


def _handshake_worker(rank, world, port, q, plan):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from amira_amd._ffi import AmgError
    from amira_amd.dist import PeerFailed, dist_build
    eng = _FakeEngine(plan[rank])
    try:
        dist_build(eng, 5)
        got = "built"
    except PeerFailed as e:
        got = ("PeerFailed", sorted(e.codes.items()), e.retry)
    except AmgError as e:
        got = ("AmgError", e.code)
    q.put((rank, got, [a for _, a in eng.calls]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("plan,want", [
    # a collision on one rank, a fatal error on the other IN THE SAME PHASE: nobody retries (the colliding rank used to
    # re-enter the build alone and wait in its first all-gather for ever), everybody raises
    (({0: -8}, {0: -3}), {0: ("PeerFailed", [(0, -2), (1, -1)], False), 1: ("AmgError", -3)}),
    # a collision on one rank only: every rank repeats the build with the next seed, which goes through
    (({0: -8}, {}), {0: "built", 1: "built"}),
    # collisions on both ranks, twice: third attempt goes through
    (({0: -8, 1: -8}, {0: -8, 1: -8}), {0: "built", 1: "built"}),
])
def test_collision_hand_shake_between_processes(plan, want):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_handshake_worker, args=(r, world, port, q, plan)) for r in range(world)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(world):
        r, res, attempts = q.get(timeout=120)     # (a hang would end here)
        got[r] = (res, attempts)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in range(world):
        assert got[r][0] == want[r], (r, got[r])
    n_attempts = max(len(p) for p in plan) + 1 if all(v == "built" for v in want.values()) else 1
    for r in range(world):
        assert got[r][1] == list(range(n_attempts)), (r, got[r][1])
