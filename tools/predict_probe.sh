#!/bin/bash
# usage: tools/predict_probe.sh — upper bound of what predicting a read's next node from the previous build's ids could
# buy the REBUILD node pass: a variant build (-DAMG_ABLATE_PREDICT) in which only a thread's first window probes its
# hashed slot and windows 1..3 "find" their key after a load from consecutive slots (the access pattern of a dense,
# id-ordered key array).  Error-free stream: every window after the first reads hits an existing node, as in a
# rebuild.  The graph is garbage; only the node_upsert times are meaningful.  Normal build first, for comparison.
cd $GRAFT_REPO_ROOT
for variant in "" "-DAMG_ABLATE_PREDICT"; do
make -C amira_amd/csrc clean > /dev/null
make -C amira_amd/csrc -j32 CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-result -Wno-unused-function -DAMG_EXPERIMENTS=0 $variant" 2>&1 | grep -E "error"
echo "== variant: ${variant:-normal}"
timeout 300 python3 - <<'PY'
import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
from amira_amd import Engine, synth
N, L, V = 1_000_000, 60, 20000
ids, sts = synth.block_reads(20250908, 0, N, L, V, 0.0)
toks = np.where(sts == 1, V + ids, V - 1 - ids).astype(np.int32).ravel()
offs = np.arange(0, (N + 1) * L, L, dtype=np.int64)
eng = Engine(0)
eng.set_reads(toks, offs, 2 * V)
for rep in range(4):
    try:
        eng.build(5)
    except Exception as e:
        print("build:", str(e)[:80])
    t = dict(eng.timings())
    print({n: round(t[n], 3) for n in ("node_upsert_head", "node_upsert", "edge_upsert") if n in t})
PY
done
make -C amira_amd/csrc clean > /dev/null
