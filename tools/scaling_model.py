"""usage: python tools/scaling_model.py [reads_per_rank [worlds, e.g. 1,8]] > profiles/r6_scaling_model.json

A MODEL, not a measurement, of `bench.py --gpus N` (weak scaling, N x reads_per_rank reads, every build merged by key
owner): the whole cleaning sweep is run with W = 1, 2, 4, 8 EMULATED ranks on the one GPU there is (one Engine per
rank, amg_dist_merge_local: the device phases are exactly those of the N-GPU run, only the wire is missing) and every
device phase is timed per rank (synchronised wall time of the phase, summed over the ranks of a step and divided by W:
libamg's own per-phase clock for the merge, amg_dist_phase_ms, a wrapper around the Engine call for the rest).  Bytes
that would cross xGMI, exchanges and host waits are counted by the merge driver itself (amg_dist_stats).  From these:

    t_rank(N)  = device ms of one rank's sweep at world N            (measured, emulated)
    t_wire(N)  = sum over exchanges of (bytes one rank sends to ONE peer) / link GB/s + latency per collective
                 (all-to-all / all-gather on a fully connected xGMI node: every peer has its own link, 7 links busy
                 at once, so the time of an exchange is the per-peer share over one link)
    t_sync(N)  = host waits per sweep on exchanged counts x t_roundtrip
    efficiency = t_rank(1, unmerged single-GPU sweep with the same fused first filter) / (t_rank(N) + t_wire(N) + t_sync(N))

Assumptions are in the output ("assumptions").  The replicated share (phases every rank runs on the GLOBAL graph:
unpacking the gathered records, emission, filter, clipping, live adjacency, components) is reported separately: it is
what does not shrink with N."""
import collections, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from amira_amd import Engine
from amira_amd import dist as D

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
WORLDS = tuple(int(x) for x in sys.argv[2].split(",")) if len(sys.argv) > 2 else (1, 2, 4, 8)   # a subset: for a profiler run
LINK_GBS, COLL_LAT_US, SYNC_US = 153.0 * 0.8, 25.0, 60.0   # xGMI link (80 % of 153 GB/s), per-collective latency, host round trip
w = bench.WORKLOADS["cfg3-sweep"]
L, k = w["L"], w["k"]
acc = collections.defaultdict(float)


def timed(name, fn):
    def wrap(self, *a, **kw):
        torch.cuda.synchronize(); t = time.perf_counter(); r = fn(self, *a, **kw); torch.cuda.synchronize()
        acc[name] += (time.perf_counter() - t) * 1e3
        return r
    return wrap


for n in ("filter", "correct_reads", "adopt_corrected", "remove_short_linear_paths", "finalize", "build", "build_filtered"):
    setattr(Engine, n, timed(n, getattr(Engine, n)))

out = {"what": "model of bench.py --gpus N from emulated ranks on one GPU; NOT a multi-GPU measurement", "reads_per_rank": N,
       "assumptions": {"link_GBs": LINK_GBS, "collective_latency_us": COLL_LAT_US, "host_round_trip_us": SYNC_US,
                       "topology": "8 GPUs fully connected, one xGMI link per peer: an exchange takes (bytes to ONE peer) / link",
                       "device_ms": "synchronised wall time of each phase, summed over a step's ranks, / W"},
       "worlds": {}}

# the single-GPU yardstick: the same sweep unmerged, first filter fused (what the merged path is compared with)
vocab, toks, offs = bench.make_tokens(w, 0, N)
gs = np.tile(np.arange(L, dtype=np.int64) * 1000, N); ge = gs + 899
rl = np.full(N, L * 1000 + 100, np.int64)
e = Engine(0)
for rep in range(3):
    e.set_reads(toks, offs, vocab.two_v); e.set_positions(gs, ge, rl)
    acc.clear(); torch.cuda.synchronize(); t = time.perf_counter()
    e.build_filtered(k, 3, 1); e.correct_reads(); e.adopt_corrected(); e.build(k); e.remove_short_linear_paths(k, want_ids=False)
    e.correct_reads(); e.adopt_corrected(); e.build(k); e.finalize(); torch.cuda.synchronize()
    single = (time.perf_counter() - t) * 1e3
e.close()
out["single_gpu_fused_sweep_ms"] = round(single, 3)

for W in WORLDS:
    vocab, toks, offs = bench.make_tokens(w, 0, W * N)
    engines = []
    for r in range(W):
        en = Engine(0)
        en.set_timing(False)
        engines.append(en)
    best = None
    for rep in range(2):
        for r, en in enumerate(engines):
            lo, hi = r * N, (r + 1) * N
            en.set_reads(toks[offs[lo]:offs[hi]], offs[lo:hi + 1] - offs[lo], vocab.two_v)
            en.set_positions(gs, ge, rl)
            en.dist_stats(reset=True)
            en.dist_phase_ms(on=True)
        acc.clear()
        torch.cuda.synchronize()
        D.dist_build_loopback(engines, k, 3, 1)
        for en in engines:
            en.correct_reads(); en.adopt_corrected()
        D.dist_build_loopback(engines, k)
        for en in engines:
            en.remove_short_linear_paths(k, want_ids=False)
        for en in engines:
            en.correct_reads(); en.adopt_corrected()
        D.dist_build_loopback(engines, k)
        for en in engines:
            en.finalize()
        torch.cuda.synchronize()
        phases = collections.defaultdict(float)
        stats = collections.defaultdict(float)
        for en in engines:
            for name, ms in en.dist_phase_ms(on=False).items():
                phases["merge:" + name] += ms / W
            for name, v in en.dist_stats(reset=True).items():
                stats[name] += v / W
        for name, ms in acc.items():
            phases[name] += ms / W
        best = (dict(phases), dict(stats))
    phases, stats = best
    replicated = sum(v for n, v in phases.items() if n.endswith("_global") or n in ("filter", "remove_short_linear_paths", "finalize"))
    t_rank = sum(phases.values())
    t_wire = 0.0 if W == 1 else (sum(stats.get(n, 0.0) for n in ("a2a_bytes_per_peer", "back_bytes_per_peer", "ag_bytes_contributed"))
                                 / (LINK_GBS * 1e9) * 1e3 + stats["exchanges"] * COLL_LAT_US * 1e-3)
    t_sync = 0.0 if W == 1 else stats["host_waits"] * SYNC_US * 1e-3
    out["worlds"][str(W)] = {
        "device_ms_per_rank_per_phase": {n: round(v, 3) for n, v in sorted(phases.items())},
        "device_ms_per_rank": round(t_rank, 3), "replicated_ms_per_rank": round(replicated, 3),
        "replicated_share": round(replicated / t_rank, 3),
        "per_rank_per_sweep": {n: int(v) for n, v in stats.items()},
        "host_round_trips_per_sweep": stats["host_waits"], "wire_ms": round(t_wire, 3), "sync_ms": round(t_sync, 3),
        "predicted_sweep_ms": round(t_rank + t_wire + t_sync, 3),
        "predicted_weak_scaling_efficiency_vs_merged_n1": None,
        "predicted_weak_scaling_efficiency_vs_single_gpu_sweep": round(single / (t_rank + t_wire + t_sync), 3)}
    for en in engines:
        en.close()
base = out["worlds"].get("1", {}).get("predicted_sweep_ms", float("nan"))
for W in out["worlds"]:
    out["worlds"][W]["predicted_weak_scaling_efficiency_vs_merged_n1"] = round(base / out["worlds"][W]["predicted_sweep_ms"], 3)
print(json.dumps(out, indent=1))
