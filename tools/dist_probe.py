"""Emulate W ranks on ONE GPU (loop-back exchange) at bench scale to see what the merge phases
cost per rank (no communication time): python tools/dist_probe.py W reads_per_rank"""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from amira_amd import Engine
from amira_amd.dist import dist_build_loopback
W = int(sys.argv[1]); N = int(sys.argv[2])
w = bench.WORKLOADS["cfg3"]
engines = []
for r in range(W):
    vocab, toks, offs = bench.make_tokens(w, r * N, (r + 1) * N)
    e = Engine(0); e.set_reads(toks, offs, vocab.two_v); engines.append(e)
single = Engine(0); vocab, toks, offs = bench.make_tokens(w, 0, N); single.set_reads(toks, offs, vocab.two_v)
for it in range(2):
    t = time.perf_counter(); single.build(5); single.filter(3, 1); torch.cuda.synchronize(); t1 = time.perf_counter() - t
    t = time.perf_counter(); dist_build_loopback(engines, 5, 3, 1); torch.cuda.synchronize(); t2 = time.perf_counter() - t
    t = time.perf_counter(); dist_build_loopback(engines, 5); torch.cuda.synchronize(); t3 = time.perf_counter() - t
    print(json.dumps({"W": W, "reads_per_rank": N, "single_build_filter_ms": round(t1 * 1e3, 1),
                      "fused_merge_ms_per_rank": round(t2 * 1e3 / W, 1), "plain_merge_ms_per_rank": round(t3 * 1e3 / W, 1),
                      "nodes_fused": engines[0].counts()["n_nodes"]}))

# ---- per-phase device time (fused), accumulated over ranks
import collections
acc = collections.defaultdict(float)
def timed(name, fn):
    def wrap(self, *a, **k):
        torch.cuda.synchronize(); t = time.perf_counter(); r = fn(self, *a, **k); torch.cuda.synchronize()
        acc[name + ":" + (str(a[0]) if name in ("pack", "reduce", "owned", "global") else "")] += time.perf_counter() - t
        return r
    return wrap
Engine.dist_nodes_local = timed("nodes_local", Engine.dist_nodes_local)
Engine.dist_edges_local = timed("edges_local", Engine.dist_edges_local)
Engine.dist_pack = timed("pack", Engine.dist_pack)
Engine.dist_reduce = timed("reduce", Engine.dist_reduce)
Engine.dist_global = timed("global", Engine.dist_global)
t = time.perf_counter(); dist_build_loopback(engines, 5, 3, 1); torch.cuda.synchronize(); tot = time.perf_counter() - t
print("fused per-rank phase ms:", {k: round(v * 1e3 / W, 2) for k, v in acc.items()}, "total/rank", round(tot * 1e3 / W, 1))
acc.clear()
t = time.perf_counter(); dist_build_loopback(engines, 5); torch.cuda.synchronize(); tot = time.perf_counter() - t
print("plain per-rank phase ms:", {k: round(v * 1e3 / W, 2) for k, v in acc.items()}, "total/rank", round(tot * 1e3 / W, 1))
