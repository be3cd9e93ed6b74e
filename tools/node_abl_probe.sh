#!/bin/bash
# usage: tools/node_abl_probe.sh "<flags>" ... — node pass of the FIRST build of the cfg 3 stream per variant build of
# amg_build_x.o, claims from one counter / from the shard counters.  -DAMG_NODE_ABL=1: creators leave out their
# per-claim stores (first-seen word, slot).  The graph is garbage: timing only.
# Measured (round 4): 0.826 / 0.776 ms as shipped, 0.826 / 0.750 without the stores: what a first build pays over a
# rebuild (0.32 ms) is the compare-and-swap, the publication and the cold line of 5.4 M creations, not their bookkeeping.
cd $GRAFT_REPO_ROOT
for flags in "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-result -Wno-unused-function -DAMG_EXPERIMENTS=0 $flags -c amira_amd/csrc/amg_build_x.hip -o amira_amd/csrc/amg_build_x.o 2>&1 | grep -E "error"
  make -C amira_amd/csrc > /dev/null 2>&1
  for sh in 0 1; do
  echo "== flags: [$flags] AMG_CLAIM_SHARDS=$sh"
  AMG_CLAIM_SHARDS=$sh timeout 300 python3 - <<'PY'
import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, bench
from amira_amd import Engine
w = bench.WORKLOADS["cfg3"]
vocab, toks3, offs3 = bench.make_tokens(w, 0, w["N"])
eng = Engine(0)
eng.set_reads(toks3, offs3, vocab.two_v)
out = []
for rep in range(4):
    try:
        eng.build(5)
    except Exception as e:
        out.append("build: " + str(e)[:40])
    tm = dict(eng.timings())
    out.append({n: round(tm[n], 3) for n in ("node_upsert_head", "node_upsert") if n in tm})
print(out[-2:])
PY
  done
done
