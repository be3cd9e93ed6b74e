"""Latency of small builds (the reference's own data sizes): wall ms per build + stage times."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from amira_amd import Engine, synth
eng = Engine(0)
for N, L, V, k in ((5000, 12, 2000, 3), (20000, 20, 5000, 3), (100000, 40, 5000, 5)):
    ids, sts = synth.block_reads(7, 0, N, L, V, 0.02)
    toks = np.where(sts == 1, V + ids, V - 1 - ids).astype(np.int32).ravel()
    offs = np.arange(0, (N + 1) * L, L, dtype=np.int64)
    eng.set_reads(toks, offs, 2 * V)
    for _ in range(3): eng.build(k)
    t = time.perf_counter()
    for _ in range(20): eng.build(k)
    dt = (time.perf_counter() - t) / 20 * 1e3
    st = dict(eng.timings())
    print(f"N={N} L={L} k={k}: {dt:.3f} ms/build wall, device stages sum {sum(st.values()):.3f} ms, nodes {eng.counts()['n_nodes']}")
    print("   ", {a: round(b, 3) for a, b in st.items()})
