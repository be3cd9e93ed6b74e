"""Per-phase wall times (ms, synchronised) of the merged build inside the cfg 3 sweep on ONE rank (loop-back exchange):
the fused-filter first build and the plain rebuild of the corrected reads, next to the single-GPU build.
python tools/dist_sweep_probe.py"""
import os, sys, time, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from amira_amd import Engine
from amira_amd.dist import dist_build_loopback
w = bench.WORKLOADS["cfg3"]; N, L, k = w["N"], w["L"], w["k"]
vocab, toks, offs = bench.make_tokens(w, 0, N)
gs = np.tile(np.arange(L, dtype=np.int64) * 1000, N); ge = gs + 899
rl = np.full(N, L * 1000 + 100, dtype=np.int64)
acc = collections.defaultdict(float)
def timed(name, fn):
    def wrap(self, *a, **kw):
        torch.cuda.synchronize(); t = time.perf_counter(); r = fn(self, *a, **kw); torch.cuda.synchronize()
        acc[name + (":" + str(a[0]) if name in ("pack", "reduce", "owned", "global") else "")] += time.perf_counter() - t
        return r
    return wrap
for n in ("dist_nodes_local", "dist_edges_local", "dist_pack", "dist_reduce", "dist_global"):
    setattr(Engine, n, timed(n[5:], getattr(Engine, n)))
eng = Engine(0)
for it in range(3):
    eng.set_reads(toks, offs, vocab.two_v); eng.set_positions(gs, ge, rl)
    acc.clear(); torch.cuda.synchronize(); t = time.perf_counter()
    dist_build_loopback([eng], k, 3, 1); torch.cuda.synchronize(); t1 = (time.perf_counter() - t) * 1e3
    p1 = {n: round(v * 1e3, 2) for n, v in acc.items()}
    eng.correct_reads(); eng.adopt_corrected()
    acc.clear(); torch.cuda.synchronize(); t = time.perf_counter()
    dist_build_loopback([eng], k); torch.cuda.synchronize(); t2 = (time.perf_counter() - t) * 1e3
    p2 = {n: round(v * 1e3, 2) for n, v in acc.items()}
    t = time.perf_counter(); eng.build(k); eng.sync(); t3 = (time.perf_counter() - t) * 1e3
print("fused first build", round(t1, 2), p1)
print("plain rebuild", round(t2, 2), p2, "| single-GPU rebuild", round(t3, 2))
