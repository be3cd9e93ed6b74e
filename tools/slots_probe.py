"""Table-size experiment: node/edge upsert time of a REBUILD (table sized node_hint x AMG_SLOT_MULT)."""
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
    toks, offs, two_v = tokens(err)
    eng.set_reads(toks, offs, two_v)
    for mult in (3, 8, 32, 128, 512):
        os.environ["AMG_SLOT_MULT"] = str(mult)
        for rep in range(3):
            eng.build(5)
        t = dict(eng.timings()); c = eng.counts()
        print(f"err={err} mult={mult} nodes={c['n_nodes']} node_slots={c['node_table_slots']} edge_slots={c['edge_table_slots']} "
              f"node_upsert={t['node_upsert']:.3f} edge_upsert={t['edge_upsert']:.3f} clears={t['node_table_clear']+t['edge_table_clear']:.3f}", flush=True)
