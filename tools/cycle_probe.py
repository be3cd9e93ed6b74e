"""which objects of one API sweep end up in reference cycles (they would only be freed by the cyclic collector, and an
engine that dies there is destroyed instead of going back to the pool)"""
import gc, os, sys
from collections import Counter
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from amira_amd import graph_utils as gu, synth
from amira_amd.io import ReadLengths, TokenizedPositions, TokenizedReads

w = dict(bench.WORKLOADS["cfg3-sweep"]); w["N"] = 20000
vocab, toks, offs = bench.make_tokens(w, 0, w["N"])
N, L, k = w["N"], w["L"], w["k"]
ids = synth.read_names(0, N)
gs = np.tile(np.arange(L, dtype=np.int64) * 1000, N); ge = gs + 899
lengths = ReadLengths(ids, np.full(N, L * 1000 + 100, np.int64))

def sweep():
    reads, pos = TokenizedReads(vocab, toks, offs, ids), TokenizedPositions(ids, offs, gs, ge)
    g = gu.build_filtered_graph(reads, k, pos, 3, 1)
    r1, p1 = g.correct_reads(lengths)
    g2 = gu.build_multiprocessed_graph(r1, k, 1, p1)
    g2.remove_short_linear_paths(k)
    r2, p2 = g2.correct_reads(lengths)
    g3 = gu.build_multiprocessed_graph(r2, k, 1, p2)
    n = g3.get_total_number_of_nodes()
    for x in (g, g2, g3):
        x.close()
    return n

sweep()
gc.collect()
gc.disable()
gc.set_debug(gc.DEBUG_SAVEALL)
sweep()
n = gc.collect()
print("unreachable objects found by the collector:", n)
print(Counter(type(o).__name__ for o in gc.garbage).most_common(15))
for o in gc.garbage:
    if type(o).__name__ in ("GeneMerGraph", "Engine", "DeviceCorrected", "TokenizedReads", "TokenizedPositions"):
        refs = [type(r).__name__ for r in gc.get_referrers(o) if r is not gc.garbage][:8]
        print(type(o).__name__, "<-", refs)
gc.set_debug(0)
gc.garbage.clear()
from amira_amd.construct_graph import GeneMerGraph, _ENGINE_POOL
from amira_amd.io import DeviceCorrected
sweep()
alive = Counter(type(o).__name__ for o in gc.get_objects() if isinstance(o, (GeneMerGraph, DeviceCorrected, TokenizedPositions, TokenizedReads)))
print("alive after a sweep, before the collector runs:", dict(alive), "pooled engines:", {d: len(v) for d, v in _ENGINE_POOL.items()})
print("collected:", gc.collect())
alive = Counter(type(o).__name__ for o in gc.get_objects() if isinstance(o, (GeneMerGraph, DeviceCorrected, TokenizedPositions, TokenizedReads)))
print("alive after the collector:", dict(alive), "pooled engines:", {d: len(v) for d, v in _ENGINE_POOL.items()})
