"""Multi-GPU build: read shards + key-owner table merge over RCCL.

One process per GPU (torch.distributed, backend "nccl" == RCCL over xGMI).  Every rank holds
a contiguous shard of the reads in its Engine; `dist_build` produces on every rank the
single-graph result (the graph GeneMerGraph would build from ALL reads) plus the node ids of
the rank's own reads.  The device work is the ten `amg_dist_*` phases of libamg; the two
all-to-alls and two all-gathers in between are issued here.

`steps()` is written as a generator that yields each exchange, so the same phase sequence
is driven either by torch.distributed (`dist_build`) or, in one process, by the loop-back
driver `dist_build_loopback` that tests use to emulate W ranks on one GPU.
"""
import os

import torch


def steps(engine, k, world, rank, token_base, token_total, min_node_cov=1, min_edge_cov=1):
    """min_node_cov / min_edge_cov > 1 fuse filter_graph into the merge (amg_dist_set_filter).
       yield ("a2a", send_tensor, send_counts, rec_bytes) -> (recv_tensor, n_recv)
       yield ("ag", owned_tensor, n_owned, rec_bytes)      -> (all_tensor, n_total)"""
    node_bytes, edge_bytes = engine.dist_record_bytes(k)
    dev = torch.device("cuda", engine.device)
    engine.dist_set_filter(min_node_cov, min_edge_cov)
    for what, rec_bytes in (("nodes", node_bytes), ("edges", edge_bytes)):
        if what == "nodes":
            send_counts = engine.dist_nodes_local(k, token_base, token_total, world)
        else:
            send_counts = engine.dist_edges_local(world)
        send = torch.empty(max(sum(send_counts), 1) * rec_bytes, dtype=torch.uint8, device=dev)
        engine.dist_pack(what, send.data_ptr())
        recv, n_recv = yield ("a2a", send, send_counts, rec_bytes)
        n_owned = engine.dist_reduce(what, recv.data_ptr(), n_recv)
        owned = torch.empty(max(n_owned, 1) * rec_bytes, dtype=torch.uint8, device=dev)
        engine.dist_owned(what, owned.data_ptr())
        everything, n_total = yield ("ag", owned, n_owned, rec_bytes)
        engine.dist_global(what, everything.data_ptr(), n_total)


class PeerFailed(RuntimeError):
    """a rank's device phase failed (table overflow, fingerprint collision, bad input): the build is off on every rank"""

    def __init__(self, ranks):
        super().__init__(f"merged build abandoned: device phase failed on rank(s) {ranks}")
        self.ranks = ranks


def exchange_a2a(buf, send_counts, rec_bytes, group=None):
    """variable-size all-to-all of whole records (works on device tensors with RCCL and on CPU
    tensors with gloo): returns (recv tensor, number of records received)."""
    import torch.distributed as dist
    dev = buf.device
    sc = torch.tensor(send_counts, dtype=torch.int64, device=dev)
    rc = torch.empty_like(sc)
    dist.all_to_all_single(rc, sc, group=group)
    recv_counts = rc.tolist()
    if min(recv_counts, default=0) < 0 or min(send_counts, default=0) < 0:   # see dist_build
        raise PeerFailed([r for r, n in enumerate(recv_counts) if n < 0])
    n_send, n_recv = sum(send_counts), sum(recv_counts)
    recv = torch.empty(max(n_recv, 1) * rec_bytes, dtype=torch.uint8, device=dev)
    dist.all_to_all_single(recv[: n_recv * rec_bytes], buf[: n_send * rec_bytes],
                           [n * rec_bytes for n in recv_counts], [n * rec_bytes for n in send_counts],
                           group=group)
    return recv, n_recv


def exchange_ag(buf, n_owned, rec_bytes, group=None):
    """variable-size all-gather of whole records: returns (all records in rank order, total)."""
    import torch.distributed as dist
    dev = buf.device
    world = dist.get_world_size(group)
    no = torch.tensor([n_owned], dtype=torch.int64, device=dev)
    allno = torch.empty(world, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(allno, no, group=group)
    counts = allno.tolist()
    if min(counts) < 0:   # see dist_build
        raise PeerFailed([r for r, n in enumerate(counts) if n < 0])
    m = max(max(counts), 1)  # equal-size contributions: pad to the largest, compact afterwards
    padded = torch.zeros(m * rec_bytes, dtype=torch.uint8, device=dev)
    padded[: n_owned * rec_bytes] = buf[: n_owned * rec_bytes]
    out = torch.empty(world * m * rec_bytes, dtype=torch.uint8, device=dev)
    dist.all_gather_into_tensor(out, padded, group=group)
    parts = [out[r * m * rec_bytes: r * m * rec_bytes + counts[r] * rec_bytes] for r in range(world)]
    total = sum(counts)
    everything = torch.cat(parts) if total else torch.empty(rec_bytes, dtype=torch.uint8, device=dev)
    return everything.contiguous(), total


def dist_build(engine, k, group=None, min_node_cov=1, min_edge_cov=1, always_exchange=False):
    """Collective: call on every rank with its own engine (reads already set).  At world size 1 the records
    do not travel (always_exchange=True sends them through the collectives anyway: tests of the plumbing)."""
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    dev = torch.device("cuda", engine.device)
    n_local = torch.tensor([engine.sizes()[1]], dtype=torch.int64, device=dev)
    gathered = torch.empty(world, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(gathered, n_local, group=group)
    tokens = gathered.tolist()
    gen = steps(engine, k, world, rank, sum(tokens[:rank]), sum(tokens), min_node_cov, min_edge_cov)
    reply = None
    expected = iter(("a2a", "ag", "a2a", "ag"))
    while True:
        nxt = next(expected, None)
        try:
            op, buf, arg, rec_bytes = gen.send(reply)
        except StopIteration:
            return
        except Exception:
            # A failing device phase must not leave the other ranks waiting in the collective they enter next:
            # take part in its count exchange with negative counts — every rank (this one included) then sees
            # them and leaves before any data moves — and re-raise the local error.
            if world > 1 and nxt is not None:
                dummy = torch.zeros(1, dtype=torch.uint8, device=dev)
                try:
                    if nxt == "a2a":
                        exchange_a2a(dummy, [-1] * world, 1, group)
                    else:
                        exchange_ag(dummy, -1, 1, group)
                except PeerFailed:
                    pass
            raise
        if world == 1 and not (always_exchange or os.environ.get("AMG_DIST_ALWAYS_EXCHANGE")):  # nothing to exchange
            n = sum(arg) if op == "a2a" else arg
            reply = (buf, n)
            continue
        reply = (exchange_a2a if op == "a2a" else exchange_ag)(buf, arg, rec_bytes, group)
        torch.cuda.current_stream(dev).synchronize()


def dist_build_loopback(engines, k, min_node_cov=1, min_edge_cov=1):
    """Emulate len(engines) ranks in ONE process (tests on a single GPU): the exchanges are
    plain tensor copies, the device phases are exactly those of dist_build."""
    world = len(engines)
    tokens = [e.sizes()[1] for e in engines]
    gens = [steps(e, k, world, r, sum(tokens[:r]), sum(tokens), min_node_cov, min_edge_cov)
            for r, e in enumerate(engines)]
    replies = [None] * world
    while True:
        reqs = []
        for r, g in enumerate(gens):
            try:
                reqs.append(g.send(replies[r]))
            except StopIteration:
                reqs.append(None)
        if all(q is None for q in reqs):
            return
        assert all(q is not None for q in reqs), "ranks fell out of step"
        op, rec_bytes = reqs[0][0], reqs[0][3]
        if op == "a2a":
            for dst in range(world):
                parts, n = [], 0
                for src in range(world):
                    _, buf, counts, _ = reqs[src]
                    off = sum(counts[:dst]) * rec_bytes
                    parts.append(buf[off: off + counts[dst] * rec_bytes])
                    n += counts[dst]
                recv = torch.cat(parts) if n else torch.empty(rec_bytes, dtype=torch.uint8, device=parts[0].device)
                replies[dst] = (recv.contiguous(), n)
        else:
            parts = [q[1][: q[2] * rec_bytes] for q in reqs]
            n = sum(q[2] for q in reqs)
            for dst in range(world):
                everything = torch.cat(parts) if n else torch.empty(rec_bytes, dtype=torch.uint8, device=parts[0].device)
                replies[dst] = (everything.contiguous(), n)
        torch.cuda.synchronize()
