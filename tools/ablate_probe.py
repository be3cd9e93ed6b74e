"""Timing experiments on the node pass of the exact-key build (AMG_X_ABLATE bit mask:
1 no output stores, 2 no first-seen check, 4 no claim phase, 8 no table probe).
cfg3 stream with errors (first build: 10 % of the windows create a node) and error-free
(every window after the first few reads hits an existing node)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from amira_amd import Engine, synth, _ffi

def tokens(err, N=1_000_000, L=60, V=20000, seed=20250908):
    ids, sts = synth.block_reads(seed, 0, N, L, V, err)
    toks = np.where(sts == 1, V + ids, V - 1 - ids).astype(np.int32).ravel()
    offs = np.arange(0, (N + 1) * L, L, dtype=np.int64)
    return toks, offs, 2 * V

eng = Engine(0)
for err in (0.02, 0.0):
    toks, offs, two_v = tokens(err)
    eng.set_reads(toks, offs, two_v)
    for abl in (32,):
        os.environ["AMG_X_ABLATE"] = str(abl)
        res = []
        for rep in range(3):
            try:
                eng.build(5)
            except _ffi.AmgError:
                pass
            res.append(dict(eng.timings()).get("node_upsert"))
        print(f"err={err} ablate={abl:2d} node_upsert ms: {res}", flush=True)
    os.environ["AMG_X_ABLATE"] = "32"
    for mult in (3, 12, 48, 200, 800):
        os.environ["AMG_SLOT_MULT"] = str(mult)
        res = []
        for rep in range(3):
            try:
                eng.build(5)
            except _ffi.AmgError:
                pass
            res.append(round(dict(eng.timings()).get("node_upsert"), 3))
        print(f"err={err} ablate=32 slot_mult={mult} slots={eng.counts()['node_table_slots']} node_upsert ms: {res}", flush=True)
    os.environ.pop("AMG_SLOT_MULT")
    os.environ.pop("AMG_X_ABLATE")
