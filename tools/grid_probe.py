import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from amira_amd import Engine, synth
def tokens(err, N=1_000_000, L=60, V=20000, seed=20250908):
    ids, sts = synth.block_reads(seed, 0, N, L, V, err)
    toks = np.where(sts == 1, V + ids, V - 1 - ids).astype(np.int32).ravel()
    return toks, np.arange(0, (N + 1) * L, L, dtype=np.int64), 2 * V
eng = Engine(0)
for err in (0.0, 0.02):
    toks, offs, two_v = tokens(err); eng.set_reads(toks, offs, two_v)
    for grid in (0, 1024, 2048, 4096, 8192):
        os.environ["AMG_X_GRID"] = str(grid)
        res = []
        for rep in range(4):
            eng.build(5)
            t = dict(eng.timings()); res.append(round(t["node_upsert"], 3))
        print(f"err={err} grid={grid} node_upsert ms: {res}", flush=True)
