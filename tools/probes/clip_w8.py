"""probe: where the tip clipping of a merged rebuild at 8 emulated ranks spends its time (stage events + wall)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench
from amira_amd import Engine
from amira_amd import dist as D
W = int(sys.argv[1]) if len(sys.argv) > 1 else 8
N = 1_000_000
w = bench.WORKLOADS["cfg3-sweep"]; L, k = w["L"], w["k"]
vocab, toks, offs = bench.make_tokens(w, 0, W * N)
gs = np.tile(np.arange(L, dtype=np.int64) * 1000, N); ge = gs + 899; rl = np.full(N, L * 1000 + 100, np.int64)
engines = [Engine(0) for _ in range(W)]
for rep in range(2):
    for r, en in enumerate(engines):
        lo, hi = r * N, (r + 1) * N
        en.set_reads(toks[offs[lo]:offs[hi]], offs[lo:hi + 1] - offs[lo], vocab.two_v); en.set_positions(gs, ge, rl)
    D.dist_build_loopback(engines, k, 3, 1)
    for en in engines:
        en.correct_reads(); en.adopt_corrected()
    D.dist_build_loopback(engines, k)
    for i, en in enumerate(engines):
        torch.cuda.synchronize(); t = time.perf_counter()
        n = en.remove_short_linear_paths(k, want_ids=False)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t) * 1e3
        if rep == 1 and i < 2:
            print(f"rank {i}: clip wall {dt:.3f} ms, removed {n}, graph {en.graph_sizes()}, stages {en.timings()}", flush=True)
    for i, en in enumerate(engines):
        torch.cuda.synchronize(); t = time.perf_counter()
        en.correct_reads()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t) * 1e3
        if rep == 1 and i < 2:
            print(f"rank {i}: correct wall {dt:.3f} ms, stages {en.timings()}", flush=True)
        en.adopt_corrected()
    D.dist_build_loopback(engines, k)
    for i, en in enumerate(engines):
        torch.cuda.synchronize(); t = time.perf_counter()
        en.finalize()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t) * 1e3
        if rep == 1 and i < 2:
            print(f"rank {i}: finalize wall {dt:.3f} ms, stages {en.timings()}", flush=True)
