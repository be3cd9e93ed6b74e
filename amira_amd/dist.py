"""Multi-GPU build: read shards + key-owner table merge over RCCL.

One process per GPU (torch.distributed, backend "nccl" == RCCL over xGMI).  Every rank holds
a contiguous shard of the reads in its Engine; `dist_build` produces on every rank the
single-graph result (the graph GeneMerGraph would build from ALL reads) plus the node ids of
the rank's own reads.  The device work is the eight `amg_dist_*` phases of libamg (include/amg.h);
the collectives between them are issued here, ON THE ENGINE'S OWN STREAM, so device phases and
collectives are ordered by the stream and the host only waits where it needs a number (the record
counts that size the next buffer):

    nodes   local -> pack ==all-to-all==> reduce ==all-gather of the survivors==>
                                                 ==all-to-all back (one reply per record)==> global
    edges   local -> pack ==all-to-all==> reduce ==all-gather==> global

An owner reduces the records of its keys and answers every record with the key's global first-seen
value; global node id = rank of first-seen = a prefix count over a bitmap of the global token space,
which every rank computes from the gathered survivors — no rank sorts or hashes the global table.

`steps()` is written as a generator that yields each exchange, so the same phase sequence
is driven either by torch.distributed (`dist_build`) or, in one process, by the loop-back
driver `dist_build_loopback` that tests use to emulate W ranks on one GPU.
"""
import os

import torch

from ._ffi import AmgError

REPLY_BYTES = 8
E_COLLISION = -8          # include/amg.h AMG_E_COLLISION: every rank repeats the build with the next seed
MAX_ATTEMPTS = 4


def _dev(engine):
    """where an engine's buffers live (engine.device None: a host-side stand-in, tests of the driver without a GPU)"""
    return torch.device("cpu") if engine.device is None else torch.device("cuda", engine.device)


def steps(engine, k, world, rank, token_base, token_total, min_node_cov=1, min_edge_cov=1, attempt=0):
    """min_node_cov / min_edge_cov > 1 fuse filter_graph into the merge (amg_dist_set_filter).
       yield ("a2a", send, send_counts, rec_bytes)            -> (recv, recv_counts)
       yield ("ag", owned, n_owned, rec_bytes)                -> (all_slots, n_slots, n_total)
       yield ("back", replies, recv_counts, send_counts)      -> my_replies (one int64 per record sent)"""
    node_bytes, edge_bytes = engine.dist_record_bytes(k)
    dev = _dev(engine)
    engine.dist_set_filter(min_node_cov, min_edge_cov)
    for what, rec_bytes in (("nodes", node_bytes), ("edges", edge_bytes)):
        if what == "nodes":
            send_counts = engine.dist_nodes_local(k, token_base, token_total, world, attempt)
        else:
            send_counts = engine.dist_edges_local(world)
        send = torch.empty(max(sum(send_counts), 1) * rec_bytes, dtype=torch.uint8, device=dev)
        engine.dist_pack(what, send.data_ptr())
        recv, recv_counts = yield ("a2a", send, send_counts, rec_bytes)
        n_recv = sum(recv_counts)
        n_sources = sum(1 for c in recv_counts if c > 0)
        owned = torch.empty(max(n_recv, 1) * rec_bytes, dtype=torch.uint8, device=dev)
        replies = None
        if what == "nodes":
            replies = torch.empty(max(n_recv, 1), dtype=torch.int64, device=dev)
            n_owned = engine.dist_reduce(what, recv.data_ptr(), n_recv, n_sources, owned.data_ptr(), replies.data_ptr())
        else:
            n_owned = engine.dist_reduce(what, recv.data_ptr(), n_recv, n_sources, owned.data_ptr())
        everything, n_slots, n_total = yield ("ag", owned, n_owned, rec_bytes)
        if what == "nodes":
            mine = yield ("back", replies, recv_counts, send_counts)
            engine.dist_global(what, everything.data_ptr(), n_slots, n_total, mine.data_ptr())
        else:
            engine.dist_global(what, everything.data_ptr(), n_slots, n_total)


class PeerFailed(RuntimeError):
    """a rank's device phase failed (table overflow, fingerprint collision, bad input): the build is off on every
    rank.  codes[r] < 0 for the ranks that failed: -1 an error, -2 a merge-key collision (retry with the next seed)"""

    def __init__(self, ranks, codes=None):
        super().__init__(f"merged build abandoned: device phase failed on rank(s) {ranks}")
        self.ranks = ranks
        self.codes = codes or {r: -1 for r in ranks}

    @property
    def retry(self):
        return bool(self.codes) and all(c == -2 for c in self.codes.values())


def _host_staged(group):
    """gloo moves CPU tensors: device buffers are staged through the host around its collectives — the whole driver
    (count exchanges, padded all-gather, reply trip, failure hand-shake) then runs between processes without RCCL,
    e.g. two ranks sharing one GPU in a test; the production transport is backend "nccl" (= RCCL), device to device"""
    import torch.distributed as dist
    return dist.get_backend(group) == "gloo"


def _a2a_single(out, inp, out_splits, in_splits, group):
    import torch.distributed as dist
    if _host_staged(group) and out.is_cuda:
        h_out = torch.empty(out.shape, dtype=out.dtype)
        dist.all_to_all_single(h_out, inp.cpu(), out_splits, in_splits, group=group)
        out.copy_(h_out)
    else:
        dist.all_to_all_single(out, inp, out_splits, in_splits, group=group)


def _all_gather(out, inp, group):
    import torch.distributed as dist
    if _host_staged(group) and out.is_cuda:
        h_out = torch.empty(out.shape, dtype=out.dtype)
        dist.all_gather_into_tensor(h_out, inp.cpu(), group=group)
        out.copy_(h_out)
    else:
        dist.all_gather_into_tensor(out, inp, group=group)


def _read_counts(t, read):
    """a small int64 count tensor on the host: through the engine's pinned mailbox when the tensor lives on the
    engine's device and the collective that filled it ran on the engine's stream (`read` = Engine.fetch_words), else
    the framework's read-back"""
    if read is not None and t.is_cuda and t.numel() <= 32:
        return read(t.data_ptr(), t.numel())
    return t.tolist()


def exchange_a2a(buf, send_counts, rec_bytes, group=None, read=None):
    """variable-size all-to-all of whole records (works on device tensors with RCCL and on CPU
    tensors with gloo): returns (recv tensor, records received from every rank)."""
    import torch.distributed as dist
    dev = buf.device
    sc = torch.tensor(send_counts, dtype=torch.int64, device=dev)
    rc = torch.empty_like(sc)
    _a2a_single(rc, sc, None, None, group)
    recv_counts = _read_counts(rc, read)
    if min(recv_counts, default=0) < 0 or min(send_counts, default=0) < 0:   # see dist_build
        bad = {r: n for r, n in enumerate(recv_counts) if n < 0}
        raise PeerFailed(sorted(bad), bad)
    n_send, n_recv = sum(send_counts), sum(recv_counts)
    recv = torch.empty(max(n_recv, 1) * rec_bytes, dtype=torch.uint8, device=dev)
    _a2a_single(recv[: n_recv * rec_bytes], buf[: n_send * rec_bytes],
                [n * rec_bytes for n in recv_counts], [n * rec_bytes for n in send_counts], group)
    return recv, recv_counts


def exchange_ag(buf, n_owned, rec_bytes, group=None, read=None):
    """variable-size all-gather of whole records as ONE equal-size all-gather: every rank contributes m = the largest
    count, its unused tail zeroed (a record's first 8 bytes are never zero, so the consumer skips the padding; owners
    are chosen by hash, so the counts are nearly equal and the padding is small).
    Returns (world * m record slots, world * m, total records)."""
    import torch.distributed as dist
    dev = buf.device
    world = dist.get_world_size(group)
    no = torch.tensor([n_owned], dtype=torch.int64, device=dev)
    allno = torch.empty(world, dtype=torch.int64, device=dev)
    _all_gather(allno, no, group)
    counts = _read_counts(allno, read)
    if min(counts) < 0:   # see dist_build
        bad = {r: n for r, n in enumerate(counts) if n < 0}
        raise PeerFailed(sorted(bad), bad)
    m = max(max(counts), 1)
    if n_owned == m and buf.numel() >= m * rec_bytes:
        padded = buf[: m * rec_bytes]
    else:
        padded = torch.zeros(m * rec_bytes, dtype=torch.uint8, device=dev)
        padded[: n_owned * rec_bytes] = buf[: n_owned * rec_bytes]
    out = torch.empty(world * m * rec_bytes, dtype=torch.uint8, device=dev)
    _all_gather(out, padded, group)
    return out, world * m, sum(counts)


def exchange_back(replies, recv_counts, send_counts, group=None):
    """the first all-to-all in reverse: one int64 per record goes back to the rank that sent the record"""
    import torch.distributed as dist
    n_send, n_recv = sum(send_counts), sum(recv_counts)
    mine = torch.empty(max(n_send, 1), dtype=torch.int64, device=replies.device)
    _a2a_single(mine[:n_send], replies[:n_recv], list(send_counts), list(recv_counts), group)
    return mine


def engine_stream(engine):
    """the engine's own HIP stream as a torch stream: tensors made and collectives issued under
    `torch.cuda.stream(engine_stream(e))` are ordered with the engine's kernels by the stream itself"""
    return torch.cuda.ExternalStream(engine.stream(), device=torch.device("cuda", engine.device))


def dist_build(engine, k, group=None, min_node_cov=1, min_edge_cov=1, always_exchange=False):
    """Collective: call on every rank with its own engine (reads already set).  At world size 1 the records
    do not travel (always_exchange=True sends them through the collectives anyway: tests of the plumbing).
    A merge-key collision between two gene-mers (AMG_E_COLLISION on the rank that owns the key) makes every
    rank repeat the build with the next fingerprint seed."""
    import contextlib
    for attempt in range(MAX_ATTEMPTS):
        try:
            with (contextlib.nullcontext() if engine.device is None else torch.cuda.stream(engine_stream(engine))):
                return _dist_build_once(engine, k, group, min_node_cov, min_edge_cov, always_exchange, attempt)
        except PeerFailed as e:
            if not e.retry or attempt + 1 == MAX_ATTEMPTS:
                raise
        except AmgError as e:
            if e.code != E_COLLISION or attempt + 1 == MAX_ATTEMPTS:
                raise


def _dist_build_once(engine, k, group, min_node_cov, min_edge_cov, always_exchange, attempt):
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    dev = _dev(engine)
    # counts of the exchanges come to the host through the engine's pinned mailbox when the transport is RCCL (the
    # collectives run on the engine's stream); gloo stages through the host anyway
    read = engine.fetch_words if (engine.device is not None and hasattr(engine, "fetch_words")
                                  and not _host_staged(group) and not os.environ.get("AMG_DIST_PLAIN_READBACK")) else None
    exchange = world > 1 or always_exchange or bool(os.environ.get("AMG_DIST_ALWAYS_EXCHANGE"))
    if world > 1:
        n_local = torch.tensor([engine.sizes()[1]], dtype=torch.int64, device=dev)
        gathered = torch.empty(world, dtype=torch.int64, device=dev)
        _all_gather(gathered, n_local, group)
        tokens = _read_counts(gathered, read)
    else:
        tokens = [engine.sizes()[1]]
    gen = steps(engine, k, world, rank, sum(tokens[:rank]), sum(tokens), min_node_cov, min_edge_cov, attempt)
    reply = None
    # the collectives in the order every rank issues them; the ones that open with a count exchange can carry a failure
    expected = iter(("a2a", "ag", "back", "a2a", "ag"))
    while True:
        nxt = next(expected, None)
        try:
            req = gen.send(reply)
        except StopIteration:
            return
        except Exception as err:
            # A failing device phase must not leave the other ranks waiting in the collective they enter next:
            # take part in its count exchange with negative counts — every rank (this one included) then sees
            # them and leaves before any data moves — and re-raise the local error.  (-2: a merge-key collision,
            # after which every rank retries with the next seed.)  No device phase runs between "ag" and "back".
            if world > 1 and nxt in ("a2a", "ag"):
                code = -2 if isinstance(err, AmgError) and err.code == E_COLLISION else -1
                dummy = torch.zeros(1, dtype=torch.uint8, device=dev)
                try:
                    if nxt == "a2a":
                        exchange_a2a(dummy, [code] * world, 1, group, read)
                    else:
                        exchange_ag(dummy, code, 1, group, read)
                except PeerFailed as seen:
                    # What every rank saw in this hand-shake decides what every rank does next: retry only when ALL
                    # the failures were collisions.  A local collision next to another rank's fatal error must not
                    # send this rank into a retry that nobody else joins (it would wait in the next all-gather for
                    # ever): the peers' verdict replaces the local one.
                    if code == -2 and not seen.retry:
                        raise seen from err
            raise
        op = req[0]
        if not exchange:   # one rank: nothing travels
            if op == "a2a":
                reply = (req[1], list(req[2]))
            elif op == "ag":
                reply = (req[1], req[2], req[2])
            else:
                reply = req[1]
        elif op == "a2a":
            reply = exchange_a2a(req[1], req[2], req[3], group, read)
        elif op == "ag":
            reply = exchange_ag(req[1], req[2], req[3], group, read)
        else:
            reply = exchange_back(req[1], req[2], req[3], group)


def dist_build_loopback(engines, k, min_node_cov=1, min_edge_cov=1):
    """Emulate len(engines) ranks in ONE process (tests on a single GPU): the exchanges are
    plain tensor copies, the device phases are exactly those of dist_build."""
    world = len(engines)
    tokens = [e.sizes()[1] for e in engines]
    for attempt in range(MAX_ATTEMPTS):
        try:
            return _loopback_once(engines, k, world, tokens, min_node_cov, min_edge_cov, attempt)
        except AmgError as e:
            if e.code != E_COLLISION or attempt + 1 == MAX_ATTEMPTS:
                raise


def _loopback_once(engines, k, world, tokens, min_node_cov, min_edge_cov, attempt):
    gens = [steps(e, k, world, r, sum(tokens[:r]), sum(tokens), min_node_cov, min_edge_cov, attempt)
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
        torch.cuda.synchronize()   # the engines' phases run on their own streams, the copies below on torch's
        op = reqs[0][0]
        dev = reqs[0][1].device
        if op == "a2a":
            rec_bytes = reqs[0][3]
            for dst in range(world):
                parts, counts = [], []
                for src in range(world):
                    _, buf, sc, _ = reqs[src]
                    off = sum(sc[:dst]) * rec_bytes
                    parts.append(buf[off: off + sc[dst] * rec_bytes])
                    counts.append(sc[dst])
                recv = torch.cat(parts) if sum(counts) else torch.empty(rec_bytes, dtype=torch.uint8, device=dev)
                replies[dst] = (recv.contiguous(), counts)
        elif op == "ag":
            rec_bytes = reqs[0][3]
            m = max(max(q[2] for q in reqs), 1)
            out = torch.zeros(world * m * rec_bytes, dtype=torch.uint8, device=dev)
            for r, q in enumerate(reqs):
                out[r * m * rec_bytes: (r * m + q[2]) * rec_bytes] = q[1][: q[2] * rec_bytes]
            n = sum(q[2] for q in reqs)
            for dst in range(world):
                replies[dst] = (out, world * m, n)
        else:   # "back": rank dst gets, from every owner src, the replies to the records it sent there
            for dst in range(world):
                parts = []
                for src in range(world):
                    _, rep, recv_counts, _ = reqs[src]
                    off = sum(recv_counts[:dst])
                    parts.append(rep[off: off + recv_counts[dst]])
                mine = torch.cat(parts) if parts else torch.empty(1, dtype=torch.int64, device=dev)
                if mine.numel() == 0:
                    mine = torch.empty(1, dtype=torch.int64, device=dev)
                replies[dst] = mine.contiguous()
        torch.cuda.synchronize()
