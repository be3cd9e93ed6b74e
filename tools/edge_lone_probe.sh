#!/bin/bash
# usage: tools/edge_lone_probe.sh "<flags>" ... — edge pass of the FIRST build of the cfg 3 stream per variant build of
# amg_build_x.o, lone classes off / on.  -DAMG_LONE_ABL=1: the lone classes are not even claimed, 2: claimed but
# nothing stored for them, 3: only their 16-byte slot is not stored (the build fails after the pass: timing only)
cd $GRAFT_REPO_ROOT
for flags in "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-result -Wno-unused-function -DAMG_EXPERIMENTS=0 $flags -c amira_amd/csrc/amg_build_x.hip -o amira_amd/csrc/amg_build_x.o 2>&1 | grep -E "error"
  make -C amira_amd/csrc > /dev/null 2>&1
  for lone in 0 1; do
  echo "== flags: [$flags] AMG_EDGE_LONE=$lone"
  AMG_EDGE_LONE=$lone timeout 300 python3 - <<'PY'
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
        out.append("build: " + str(e)[:50])
    tm = dict(eng.timings())
    out.append({n: round(tm[n], 3) for n in ("node_count", "edge_table_clear", "edge_upsert_head", "edge_upsert", "edge_count", "edge_rank") if n in tm})
print(out[-2:])
PY
  done
done
