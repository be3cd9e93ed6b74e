"""cProfile of assign_reads_to_genes at BASELINE config 4 (1 M reads, 10 planted AMR genes), with the time
the cyclic garbage collector takes counted separately (usage: cfg4_profile.py [N] [cumulative|tottime])"""
import cProfile, gc, io, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from amira_amd import GeneMerGraph, synth
from amira_amd.io import TokenizedPositions, TokenizedReads
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
order = sys.argv[2] if len(sys.argv) > 2 else "cumulative"
w = dict(bench.WORKLOADS["cfg4"], N=N)
vocab, toks, offs = bench.make_tokens(w, 0, N)
ids = synth.read_names(0, N)
gs = np.tile(np.arange(w["L"], dtype=np.int64) * 1000, N)
g = GeneMerGraph(TokenizedReads(vocab, toks, offs, ids), w["k"], TokenizedPositions(ids, offs, gs, gs + 899))
spent = {"gc": 0.0, "t0": 0.0, "n": 0}
def on_gc(phase, info):
    if phase == "start":
        spent["t0"] = time.perf_counter()
    else:
        spent["gc"] += time.perf_counter() - spent["t0"]; spent["n"] += 1
gc.callbacks.append(on_gc)
t = time.perf_counter()
pr = cProfile.Profile(); pr.enable()
g.assign_reads_to_genes([f"amr{j}" for j in range(10)], 1, {}, None)
pr.disable()
print(f"wall {time.perf_counter() - t:.2f} s (under cProfile), garbage collector {spent['gc']:.2f} s in {spent['n']} runs")
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats(order).print_stats(24); print(s.getvalue()[-3800:])
