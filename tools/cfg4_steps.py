"""usage: python tools/cfg4_steps.py [steps] — wall time of every cfg 4 step (build + assign_reads_to_genes + close) by
itself, with what the cyclic collector did meanwhile: how noisy is the Python side?"""
import gc, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from amira_amd import GeneMerGraph, synth
from amira_amd.io import TokenizedPositions, TokenizedReads

w = bench.WORKLOADS["cfg4"]
N, L, k = w["N"], w["L"], w["k"]
vocab, toks, offs = bench.make_tokens(w, 0, N)
ids = synth.read_names(0, N)
reads = TokenizedReads(vocab, toks, offs, ids)
gs = np.tile(np.arange(L, dtype=np.int64) * 1000, N)
pos = TokenizedPositions(ids, offs, gs, gs + 899)
genes = [f"amr{j}" for j in range(w["n_amr"])]
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
    c0 = [s["collections"] for s in gc.get_stats()]
    t0 = time.perf_counter()
    g = GeneMerGraph(reads, k, pos)
    t1 = time.perf_counter()
    g.assign_reads_to_genes(genes, 1, {}, None)
    t2 = time.perf_counter()
    g.close()
    del g
    t3 = time.perf_counter()
    c1 = [s["collections"] for s in gc.get_stats()]
    print(f"step {i}: {1e3 * (t3 - t0):7.1f} ms  build {1e3 * (t1 - t0):6.1f}  assign {1e3 * (t2 - t1):7.1f}  close {1e3 * (t3 - t2):6.1f}"
          f"  collections by generation {[b - a for a, b in zip(c0, c1)]}", flush=True)
