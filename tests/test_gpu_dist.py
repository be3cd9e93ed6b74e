"""Read-sharded build + table merge: W emulated ranks on ONE GPU (loop-back exchange) must
reproduce the unsharded graph exactly — node / edge tables on every rank, and each rank's
window -> node ids equal to its slice of the unsharded result."""
import numpy as np
import pytest

import procedures as P
from helpers import csr_lists
from test_gpu_sweep import flat_positions

pytestmark = pytest.mark.gpu


@pytest.fixture(params=["exact", "fp"], autouse=True)
def key_mode(request, monkeypatch):
    """every test runs twice: shards that keep their local tables in the exact-key layout (the
    default when the tuple fits) and shards on the fingerprint path (large k / vocabulary)"""
    if request.param == "fp":
        monkeypatch.setenv("AMG_KEY_MODE", "fp")
    return request.param


def shard_bounds(n_reads, world):
    return [(r * n_reads) // world for r in range(world + 1)]


def make_shards(toks, offs, world):
    b = shard_bounds(len(offs) - 1, world)
    out = []
    for r in range(world):
        lo, hi = b[r], b[r + 1]
        o = offs[lo:hi + 1] - offs[lo]
        out.append((toks[offs[lo]:offs[hi]], o, lo, hi))
    return out


def graph_state(eng):
    n, e = eng.nodes(), eng.edges()
    off, adj = eng.node_adj()
    return {"tokens": n["tokens"], "coverage": n["coverage"], "first_dir": n["first_dir"],
            "component": n["component"], "alive": n["alive"], "src": e["src"], "tgt": e["tgt"],
            "sdir": e["sdir"], "tdir": e["tdir"], "ecov": e["coverage"], "ealive": e["alive"],
            "adj_off": off, "adj": adj}


def assert_same_graph(a, b):
    for key in a:
        assert np.array_equal(a[key], b[key]), key


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("case", [("synth", 7, 400, 30, 300, 5, 0.03), ("synth", 13, 300, 40, 250, 7, 0.02),
                                  ("fixture", "nine", 3), ("fixture", "five", 5)])
def test_sharded_build_equals_unsharded(world, case):
    from amira_amd import Engine, tokenize
    from amira_amd.dist import dist_build_loopback
    if case[0] == "synth":
        _, seed, N, L, V, k, err = case
        reads, _, _ = P.synth_inputs(seed, N, L, V, err)
    else:
        reads, _ = P.fixture(case[1])
        k = case[2]
    vocab, toks, offs, read_ids = tokenize(reads)
    ref = Engine(0)
    ref.set_reads(toks, offs, vocab.two_v)
    ref.build(k)
    want = graph_state(ref)
    want_node, want_dir = ref.read_nodes()
    engines = []
    for t, o, lo, hi in make_shards(toks, offs, world):
        e = Engine(0)
        e.set_reads(t, o, vocab.two_v)
        engines.append(e)
    dist_build_loopback(engines, k)
    for r, (e, (t, o, lo, hi)) in enumerate(zip(engines, make_shards(toks, offs, world))):
        assert_same_graph(graph_state(e), want)
        node, d = e.read_nodes()
        assert np.array_equal(node, want_node[offs[lo]:offs[hi]])
        assert np.array_equal(d, want_dir[offs[lo]:offs[hi]])
        c = e.counts()
        assert c["n_nodes"] == ref.counts()["n_nodes"] and c["n_edges"] == ref.counts()["n_edges"]
        assert c["n_components"] == ref.counts()["n_components"]
    for e in engines + [ref]:
        e.close()


@pytest.mark.parametrize("thr", [None, (3, 1)])
def test_rank_without_reads(thr):
    """a rank that holds no reads at all (and one that holds a single short read) takes part in every exchange and
    ends with the whole graph"""
    from amira_amd import Engine, tokenize
    from amira_amd.dist import dist_build_loopback
    reads, _, _ = P.synth_inputs(7, 400, 30, 300, 0.03)
    k = 5
    vocab, toks, offs, _ = tokenize(reads)
    R = len(offs) - 1
    bounds = [0, 0, R // 2, R // 2, R]          # ranks 0 and 2 are empty
    ref = Engine(0)
    ref.set_reads(toks, offs, vocab.two_v)
    ref.build(k)
    if thr:
        ref.filter(*thr)
    engines = []
    for r in range(len(bounds) - 1):
        lo, hi = bounds[r], bounds[r + 1]
        e = Engine(0)
        e.set_reads(toks[offs[lo]:offs[hi]], offs[lo:hi + 1] - offs[lo], vocab.two_v)
        engines.append(e)
    dist_build_loopback(engines, k, *(thr or (1, 1)))
    if thr:
        want = live_state(ref)
        for r, e in enumerate(engines):
            got = live_state(e)
            lo, hi = bounds[r], bounds[r + 1]
            for key in ("tokens", "coverage", "first_dir", "src", "tgt", "sdir", "tdir", "ecov"):
                assert np.array_equal(got[key], want[key]), key
            assert np.array_equal(got["tok_node"], want["tok_node"][offs[lo]:offs[hi]])
            assert np.array_equal(got["to_correct"], want["to_correct"][lo:hi])
    else:
        want = graph_state(ref)
        want_node = ref.read_nodes()[0]
        for r, e in enumerate(engines):
            assert_same_graph(graph_state(e), want)
            assert np.array_equal(e.read_nodes()[0], want_node[offs[bounds[r]]:offs[bounds[r + 1]]])
    for e in engines + [ref]:
        e.close()


@pytest.mark.parametrize("derive", [True, False])
@pytest.mark.parametrize("world", [2, 3])
def test_sharded_sweep_equals_unsharded(world, derive, monkeypatch):
    """build -> filter -> correct -> build -> clip -> correct -> build with every build merged
    across the emulated ranks; the concatenation of the ranks' corrected reads must equal the
    single-GPU sweep.  derive: the third merged build is made from the second one's live part when no rank
    re-threaded a read after the clip (amg_derive.hip; the ranks ask each other) — or never (AMG_NO_DERIVE=1)."""
    if not derive:
        monkeypatch.setenv("AMG_NO_DERIVE", "1")
    from amira_amd import Engine, tokenize
    from amira_amd.dist import dist_build_loopback
    reads, pos, fq = P.synth_inputs(17, 800, 40, 150, 0.05)
    k = 5
    vocab, toks, offs, read_ids = tokenize(reads)
    gs, ge = flat_positions(read_ids, reads, pos)
    rl = np.asarray([len(fq[r]["sequence"]) for r in read_ids], dtype=np.int64)

    def sweep_single():
        e = Engine(0)
        e.set_reads(toks, offs, vocab.two_v)
        e.set_positions(gs, ge, rl)
        outs = []
        e.build(k); e.filter(3, 1)
        n = e.correct_reads(); outs.append(e.corrected(*n, True)); e.adopt_corrected()
        e.build(k); e.remove_short_linear_paths(k)
        n = e.correct_reads(); outs.append(e.corrected(*n, True)); e.adopt_corrected()
        e.build(k)
        g = graph_state(e)
        e.close()
        return outs, g

    want_outs, want_graph = sweep_single()
    engines, bounds = [], shard_bounds(len(read_ids), world)
    for r in range(world):
        lo, hi = bounds[r], bounds[r + 1]
        e = Engine(0)
        e.set_reads(toks[offs[lo]:offs[hi]], offs[lo:hi + 1] - offs[lo], vocab.two_v)
        e.set_positions(gs[offs[lo]:offs[hi]], ge[offs[lo]:offs[hi]], rl[lo:hi])
        engines.append(e)
    got_outs = []
    dist_build_loopback(engines, k)
    for e in engines:
        e.filter(3, 1)
    got_outs.append([e.corrected(*e.correct_reads(), True) for e in engines])
    for e in engines:
        e.adopt_corrected()
    dist_build_loopback(engines, k)
    removed = [sorted(e.remove_short_linear_paths(k).tolist()) for e in engines]
    assert all(x == removed[0] for x in removed)
    got_outs.append([e.corrected(*e.correct_reads(), True) for e in engines])
    for e in engines:
        e.adopt_corrected()
    for e in engines:
        e.dist_stats(reset=True)
    dist_build_loopback(engines, k)
    taken = [e.dist_stats()["derived_builds"] for e in engines]
    assert taken == [1 if derive else 0] * world, taken
    for e in engines:
        assert_same_graph(graph_state(e), want_graph)
    for stage in range(2):
        for key in ("tokens", "gene_start", "gene_end", "changed"):
            cat = np.concatenate([o[key] for o in got_outs[stage]])
            assert np.array_equal(cat, want_outs[stage][key]), (stage, key)
        lens = np.concatenate([np.diff(o["read_offsets"]) for o in got_outs[stage]])
        assert np.array_equal(lens, np.diff(want_outs[stage]["read_offsets"]))
    for e in engines:
        e.close()


def test_dist_build_over_rccl_world1(monkeypatch, tmp_path):
    """amg_dist_init / amg_dist_merge over RCCL itself (libamg's own communicator; its ncclSend / ncclRecv groups and
    all-gathers on the engine's stream), world = 1 with every exchange forced through the transport"""
    import os
    import socket
    import torch
    import torch.distributed as dist
    from amira_amd import Engine, tokenize
    from amira_amd.dist import dist_build
    monkeypatch.setenv("AMG_DIST_ALWAYS_EXCHANGE", "1")
    torch.cuda.set_device(0)
    # (a file as the rendezvous: a TCP port picked here can be taken by the time the store listens on it)
    dist.init_process_group("nccl", init_method=f"file://{tmp_path}/rendezvous", rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    try:
        reads, _, _ = P.synth_inputs(7, 400, 30, 300, 0.03)
        vocab, toks, offs, _ = tokenize(reads)
        ref = Engine(0); ref.set_reads(toks, offs, vocab.two_v); ref.build(5)
        eng = Engine(0); eng.set_reads(toks, offs, vocab.two_v); dist_build(eng, 5)
        assert eng.dist_stats()["exchanges"] == 10     # five per kind of record, all of them through RCCL
        assert_same_graph(graph_state(eng), graph_state(ref))
        assert np.array_equal(eng.read_nodes()[0], ref.read_nodes()[0])
        eng.close(); ref.close()
    finally:
        dist.destroy_process_group()


def test_dist_merge_without_torch_distributed():
    """the C entry points alone, as a C caller would use them: unique id, amg_dist_init, amg_dist_merge (world 1)"""
    from amira_amd import Engine, tokenize
    reads, _, _ = P.synth_inputs(7, 400, 30, 300, 0.03)
    vocab, toks, offs, _ = tokenize(reads)
    ref = Engine(0); ref.set_reads(toks, offs, vocab.two_v); ref.build(5)
    eng = Engine(0); eng.set_reads(toks, offs, vocab.two_v)
    eng.dist_init(Engine.dist_unique_id(), 0, 1)
    eng.dist_merge(5)
    assert_same_graph(graph_state(eng), graph_state(ref))
    eng.dist_finalize()
    eng.dist_merge(5)          # world 1 needs no communicator
    assert_same_graph(graph_state(eng), graph_state(ref))
    eng.close(); ref.close()


@pytest.mark.parametrize("world", [2, 4])
def test_merge_key_collisions_repeat_the_build_on_every_rank(world, monkeypatch):
    """merge keys cut to 10 bits on the first attempt (test hook): gene-mers share them, some rank notices (a tuple that
    is not its key's holder's; on fingerprint shards the edge pass's verify), everybody learns it in the next count
    exchange and repeats the build with the next seed"""
    from amira_amd import Engine, tokenize
    from amira_amd.dist import dist_build_loopback
    monkeypatch.setenv("AMG_TEST_DIST_WEAK_KEYS", "1")
    reads, _, _ = P.synth_inputs(17, 800, 40, 150, 0.05)
    k = 5
    vocab, toks, offs, _ = tokenize(reads)
    ref = Engine(0); ref.set_reads(toks, offs, vocab.two_v); ref.build(k)
    want = graph_state(ref)
    engines = []
    for t, o, lo, hi in make_shards(toks, offs, world):
        e = Engine(0); e.set_reads(t, o, vocab.two_v); engines.append(e)
    dist_build_loopback(engines, k)
    for e in engines:
        assert e.dist_stats()["repeated_builds"] == 1
        assert_same_graph(graph_state(e), want)
    for e in engines + [ref]:
        e.close()


def test_failing_rank_releases_its_peers(monkeypatch):
    """a rank whose device phase fails takes part in the next count exchange with a negative code: nobody waits for it,
    everybody raises — the failing rank its own error, the others "rank 1 failed" """
    from amira_amd import Engine, tokenize
    from amira_amd._ffi import AmgError
    monkeypatch.setenv("AMG_TEST_DIST_FAIL", "1")
    reads, _, _ = P.synth_inputs(7, 400, 30, 300, 0.03)
    vocab, toks, offs, _ = tokenize(reads)
    engines = []
    for t, o, lo, hi in make_shards(toks, offs, 3):
        e = Engine(0); e.set_reads(t, o, vocab.two_v); engines.append(e)
    for r, e in enumerate(engines):
        e.dist_init_external(r, 3)
        e.dist_merge_begin(5)
    # by hand what amg_dist_merge_local does, to see every rank's own verdict
    xs = [e.dist_merge_next() for e in engines]
    assert all(x is not None and x.kind == 1 and x.elem_bytes == 32 for x in xs)     # the count messages
    msgs = [np.empty(3 * 4, np.int64) for _ in engines]
    for e, x, m in zip(engines, xs, msgs):
        e.copy_d2h(x.send, m)
    for dst, (e, x) in enumerate(zip(engines, xs)):
        e.copy_h2d(x.recv, np.concatenate([m[4 * dst: 4 * dst + 4] for m in msgs]))
    codes = []
    for e in engines:
        with pytest.raises(AmgError) as err:
            e.dist_merge_next()
        codes.append(err.value.code)
    assert codes == [-7, -3, -7]          # AMG_E_DIST on the peers, the failing rank's own AMG_E_STATE
    monkeypatch.delenv("AMG_TEST_DIST_FAIL")
    from amira_amd.dist import dist_build_loopback
    dist_build_loopback(engines, 5)       # the engines are usable afterwards
    for e in engines:
        e.close()


def live_state(eng):
    """graph restricted to live nodes / edges with dense renumbering (component ids left out:
    a fused-filter build labels the components of the FILTERED graph)."""
    n, e = eng.nodes(), eng.edges()
    off, adj = eng.node_adj()
    nk, ek = n["alive"] != 0, e["alive"] != 0
    nmap, emap = np.cumsum(nk) - 1, np.cumsum(ek) - 1
    rows = csr_lists(off, adj, e["alive"])
    tok_node, tok_dir = eng.read_nodes()
    mapped = tok_node.copy()
    m = tok_node >= 0
    mapped[m] = nmap[tok_node[m]]
    return {"tokens": n["tokens"][nk], "coverage": n["coverage"][nk], "first_dir": n["first_dir"][nk],
            "src": nmap[e["src"][ek]], "tgt": nmap[e["tgt"][ek]], "sdir": e["sdir"][ek], "tdir": e["tdir"][ek],
            "ecov": e["coverage"][ek],
            "adj": [[int(emap[x]) for x in rows[r]] for i in np.nonzero(nk)[0] for r in (2 * i, 2 * i + 1)],
            "tok_node": mapped, "tok_dir": np.where(m, tok_dir, 0), "to_correct": eng.reads_to_correct()}


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("thr", [(3, 1), (2, 2)])
def test_fused_filter_merge_equals_build_then_filter(world, thr):
    from amira_amd import Engine, tokenize
    from amira_amd.dist import dist_build_loopback
    reads, _, _ = P.synth_inputs(17, 800, 40, 150, 0.05)
    k = 5
    vocab, toks, offs, _ = tokenize(reads)
    ref = Engine(0)
    ref.set_reads(toks, offs, vocab.two_v)
    ref.build(k)
    ref.filter(*thr)
    want = live_state(ref)
    engines = []
    for t, o, lo, hi in make_shards(toks, offs, world):
        e = Engine(0)
        e.set_reads(t, o, vocab.two_v)
        engines.append(e)
    dist_build_loopback(engines, k, *thr)
    bounds = shard_bounds(len(offs) - 1, world)
    for r, e in enumerate(engines):
        got = live_state(e)
        lo, hi = bounds[r], bounds[r + 1]
        for key in ("tokens", "coverage", "first_dir", "src", "tgt", "sdir", "tdir", "ecov"):
            assert np.array_equal(got[key], want[key]), key
        assert got["adj"] == want["adj"]
        assert np.array_equal(got["tok_node"], want["tok_node"][offs[lo]:offs[hi]])
        assert np.array_equal(got["tok_dir"], want["tok_dir"][offs[lo]:offs[hi]])
        assert np.array_equal(got["to_correct"], want["to_correct"][lo:hi])
        assert e.counts()["n_nodes"] == len(want["coverage"])  # only survivors were replicated
    for e in engines + [ref]:
        e.close()


def test_sharded_sweep_with_fused_first_filter():
    """the bench's N > 1 sweep: first build merged WITH filter_graph(3,1) fused in"""
    from amira_amd import Engine, tokenize
    from amira_amd.dist import dist_build_loopback
    reads, pos, fq = P.synth_inputs(17, 800, 40, 150, 0.05)
    k, world = 5, 3
    vocab, toks, offs, read_ids = tokenize(reads)
    gs, ge = flat_positions(read_ids, reads, pos)
    rl = np.asarray([len(fq[r]["sequence"]) for r in read_ids], dtype=np.int64)
    single = Engine(0)
    single.set_reads(toks, offs, vocab.two_v)
    single.set_positions(gs, ge, rl)
    single.build(k); single.filter(3, 1)
    want = single.corrected(*single.correct_reads(), True)
    engines, bounds = [], shard_bounds(len(read_ids), world)
    for r in range(world):
        lo, hi = bounds[r], bounds[r + 1]
        e = Engine(0)
        e.set_reads(toks[offs[lo]:offs[hi]], offs[lo:hi + 1] - offs[lo], vocab.two_v)
        e.set_positions(gs[offs[lo]:offs[hi]], ge[offs[lo]:offs[hi]], rl[lo:hi])
        engines.append(e)
    dist_build_loopback(engines, k, 3, 1)
    got = [e.corrected(*e.correct_reads(), True) for e in engines]
    for key in ("tokens", "gene_start", "gene_end", "changed"):
        assert np.array_equal(np.concatenate([o[key] for o in got]), want[key]), key
    for e in engines + [single]:
        e.close()


def _rccl_worker(rank, world, port, out_dir, empty_rank=None):
    """one rank of a REAL multi-process merged build over RCCL (needs >= world GPUs on the node)"""
    import os
    os.environ.update(MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    import sys
    import numpy as np
    import torch
    import torch.distributed as dist
    try:   # an environment in which two ranks cannot even meet is not a failure of the merge (exit code 77 = skip)
        torch.cuda.set_device(rank)
        dist.init_process_group("nccl", init_method=f"file://{port}", rank=rank, world_size=world,
                                device_id=torch.device("cuda", rank))
        probe = torch.ones(1, device="cuda")
        dist.all_reduce(probe)
        assert float(probe.item()) == world
    except Exception as exc:  # noqa: BLE001
        print("RCCL set-up failed:", exc, flush=True)
        sys.exit(77)
    from amira_amd import Engine, tokenize
    from amira_amd.dist import dist_build
    reads, _, _ = P.synth_inputs(7, 400, 30, 300, 0.03)
    vocab, toks, offs, _ = tokenize(reads)
    t, o, lo, hi = _rccl_shards(toks, offs, world, empty_rank)[rank]
    eng = Engine(rank)
    eng.set_reads(t, o, vocab.two_v)
    dist_build(eng, 5)
    st = graph_state(eng)
    st["tok_node"] = eng.read_nodes()[0]
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), **st)
    eng.close()
    dist.barrier()
    dist.destroy_process_group()


def _rccl_shards(toks, offs, world, empty_rank):
    """shards of the two-process test; empty_rank: that rank holds no reads, the others share them"""
    if empty_rank is None:
        return make_shards(toks, offs, world)
    rest = make_shards(toks, offs, world - 1)
    R = len(offs) - 1
    edge = 0 if empty_rank == 0 else rest[empty_rank - 1][3]
    rest.insert(empty_rank, (toks[:0], offs[:1] - offs[0], edge, edge))
    return rest


@pytest.mark.parametrize("empty_rank", [None, 1])
def test_dist_build_over_rccl_two_processes(tmp_path, empty_rank):
    """two processes, two GPUs, RCCL all-to-all / all-gather between them: every rank must end with the graph the
    unsharded engine builds (skipped on a single-GPU box; the driver's multi-GPU tier runs it)"""
    import socket
    import torch
    import torch.multiprocessing as mp
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    from amira_amd import Engine, tokenize
    world = 2
    port = str(tmp_path / "rendezvous")    # (a file: see test_dist_build_over_rccl_world1)
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_rccl_worker, args=(r, world, port, str(tmp_path), empty_rank)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=300)
    for p in procs:   # a rank left waiting for one that gave up
        if p.is_alive():
            p.terminate()
            p.join(timeout=30)
    if any(p.exitcode == 77 for p in procs):
        pytest.skip("the two ranks could not set RCCL up on this machine")
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    reads, _, _ = P.synth_inputs(7, 400, 30, 300, 0.03)
    vocab, toks, offs, _ = tokenize(reads)
    ref = Engine(0)
    ref.set_reads(toks, offs, vocab.two_v)
    ref.build(5)
    want = graph_state(ref)
    ref_nodes = ref.read_nodes()[0]
    shards = _rccl_shards(toks, offs, world, empty_rank)
    for r in range(world):
        got = np.load(tmp_path / f"rank{r}.npz")
        for key in want:
            assert np.array_equal(got[key], want[key]), (r, key)
        assert np.array_equal(got["tok_node"], ref_nodes[offs[shards[r][2]]:offs[shards[r][3]]])
    ref.close()


def _gloo_worker(rank, world, port, out_dir, empty_rank, sweep, env=None):
    """one rank of a multi-PROCESS merged build whose ranks share ONE GPU: torch.distributed over gloo (the record
    buffers are staged through the host around the collectives, amira_amd/dist.py) — everything of the N > 1 driver
    except the RCCL transport: count exchanges, padded all-gather, reply trip, the order of the collectives"""
    import os
    os.environ.update(MASTER_ADDR="127.0.0.1")
    os.environ.update(env or {})
    import numpy as np
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", init_method=f"file://{port}", rank=rank, world_size=world)
    from amira_amd import Engine, tokenize
    from amira_amd.dist import dist_build
    reads, pos, fq = P.synth_inputs(17, 800, 40, 150, 0.05)
    vocab, toks, offs, read_ids = tokenize(reads)
    t, o, lo, hi = _rccl_shards(toks, offs, world, empty_rank)[rank]
    eng = Engine(0)
    eng.set_reads(t, o, vocab.two_v)
    out = {}
    if not sweep:
        dist_build(eng, 5)
        out = graph_state(eng)
        out["tok_node"] = eng.read_nodes()[0]
    else:
        gs, ge = flat_positions(read_ids, reads, pos)
        rl = np.asarray([len(fq[r]["sequence"]) for r in read_ids], dtype=np.int64)
        eng.set_positions(gs[offs[lo]:offs[hi]], ge[offs[lo]:offs[hi]], rl[lo:hi])
        dist_build(eng, 5, None, 3, 1)                       # first build with filter_graph(3, 1) fused into the merge
        c1 = eng.corrected(*eng.correct_reads(), True)
        eng.adopt_corrected()
        dist_build(eng, 5)
        removed = eng.remove_short_linear_paths(5)
        c2 = eng.corrected(*eng.correct_reads(), True)
        eng.adopt_corrected()
        dist_build(eng, 5)
        out = graph_state(eng)
        for i, c in enumerate((c1, c2)):
            for key in ("tokens", "gene_start", "gene_end", "changed"):
                out[f"c{i}_{key}"] = c[key]
            out[f"c{i}_len"] = np.diff(c["read_offsets"])
        out["removed"] = np.sort(removed)
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), **out)
    eng.close()
    dist.barrier()
    dist.destroy_process_group()


def _run_gloo(tmp_path, world, empty_rank, sweep, env=None):
    import torch.multiprocessing as mp
    port = str(tmp_path / "rendezvous")    # (a file: see test_dist_build_over_rccl_world1)
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_gloo_worker, args=(r, world, port, str(tmp_path), empty_rank, sweep, env)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=300)
    for p in procs:
        if p.is_alive():
            p.terminate()
            p.join(timeout=30)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    return [np.load(tmp_path / f"rank{r}.npz") for r in range(world)]


@pytest.mark.parametrize("world,empty_rank", [(2, None), (3, 1)])
@pytest.mark.parametrize("weak_keys", [False, True])
def test_dist_build_between_processes_sharing_the_gpu(tmp_path, world, empty_rank, key_mode, weak_keys):
    """the merge between real processes (gloo, one GPU shared: libamg asks for the exchanges, amira_amd.dist performs
    them through the host): every rank ends with the unsharded graph; weak_keys: after a repeated build (merge-key
    collisions on the first attempt, test hook)"""
    from amira_amd import Engine, tokenize
    if weak_keys and (world, empty_rank) != (2, None):
        pytest.skip("one configuration is enough")
    got = _run_gloo(tmp_path, world, empty_rank, sweep=False, env={"AMG_TEST_DIST_WEAK_KEYS": "1"} if weak_keys else None)
    reads, _, _ = P.synth_inputs(17, 800, 40, 150, 0.05)
    vocab, toks, offs, _ = tokenize(reads)
    ref = Engine(0)
    ref.set_reads(toks, offs, vocab.two_v)
    ref.build(5)
    want = graph_state(ref)
    ref_nodes = ref.read_nodes()[0]
    shards = _rccl_shards(toks, offs, world, empty_rank)
    for r in range(world):
        for key in want:
            assert np.array_equal(got[r][key], want[key]), (r, key)
        assert np.array_equal(got[r]["tok_node"], ref_nodes[offs[shards[r][2]]:offs[shards[r][3]]])
    ref.close()


def test_dist_sweep_between_processes_sharing_the_gpu(tmp_path, key_mode):
    """the bench's N > 1 sweep (fused first filter, every build merged) between two real processes over gloo: final
    graph on every rank and the concatenated corrected reads equal the single-GPU sweep"""
    from amira_amd import Engine, tokenize
    world = 2
    got = _run_gloo(tmp_path, world, None, sweep=True)
    reads, pos, fq = P.synth_inputs(17, 800, 40, 150, 0.05)
    vocab, toks, offs, read_ids = tokenize(reads)
    gs, ge = flat_positions(read_ids, reads, pos)
    rl = np.asarray([len(fq[r]["sequence"]) for r in read_ids], dtype=np.int64)
    e = Engine(0)
    e.set_reads(toks, offs, vocab.two_v)
    e.set_positions(gs, ge, rl)
    e.build(5); e.filter(3, 1)
    c1 = e.corrected(*e.correct_reads(), True); e.adopt_corrected()
    e.build(5); removed = np.sort(e.remove_short_linear_paths(5))
    c2 = e.corrected(*e.correct_reads(), True); e.adopt_corrected()
    e.build(5)
    want = graph_state(e)
    for r in range(world):
        for key in want:
            assert np.array_equal(got[r][key], want[key]), (r, key)
        assert np.array_equal(got[r]["removed"], removed)
    for i, c in enumerate((c1, c2)):
        for key in ("tokens", "gene_start", "gene_end", "changed"):
            assert np.array_equal(np.concatenate([got[r][f"c{i}_{key}"] for r in range(world)]), c[key]), (i, key)
        assert np.array_equal(np.concatenate([got[r][f"c{i}_len"] for r in range(world)]), np.diff(c["read_offsets"]))
    e.close()
