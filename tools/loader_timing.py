"""Writers and loader of the native JSON front end at cfg 3's size, the loader with its phase timings (AMG_CALLS_TIMING=1).
usage: python tools/loader_timing.py [threads]   (threads -> AMG_CALLS_THREADS)"""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1:
    os.environ["AMG_CALLS_THREADS"] = sys.argv[1]
import numpy as np
import bench
from amira_amd import synth
from amira_amd.io import load_gene_calls, write_gene_calls, write_gene_positions

w = bench.WORKLOADS["cfg3-sweep"]
vocab, toks, offs = bench.make_tokens(w, 0, w["N"])
N, L = w["N"], w["L"]
ids = synth.read_names(0, N)
gs = np.tile(np.arange(L, dtype=np.int64) * 1000, N)
base = "/dev/shm" if os.path.isdir("/dev/shm") else None
with tempfile.TemporaryDirectory(dir=base) as d:
    cj, pj = os.path.join(d, "c.json"), os.path.join(d, "p.json")
    for rep in range(2):   # writers: a fresh file, then the same file replaced
        t = time.perf_counter(); write_gene_calls(cj, vocab, toks, offs, ids); t1 = time.perf_counter() - t
        t = time.perf_counter(); write_gene_positions(pj, gs, gs + 899, offs, ids); t2 = time.perf_counter() - t
        print("write calls %.3f s (%.2f GB/s), positions %.3f s (%.2f GB/s)" % (t1, os.path.getsize(cj) / t1 / 1e9, t2, os.path.getsize(pj) / t2 / 1e9))
    load_gene_calls(cj, pj)
    os.environ["AMG_CALLS_TIMING"] = "1"
    t = time.perf_counter()
    load_gene_calls(cj, pj)
    print("wall", round(time.perf_counter() - t, 3), "threads", os.environ.get("AMG_CALLS_THREADS", "auto"))
