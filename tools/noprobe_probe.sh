#!/bin/bash
# usage: tools/noprobe_probe.sh — the table passes without their table: every window "finds" its key in the slot it
# would probe first, no load from the table (build made here with -DAMG_ABLATE_NOPROBE; the graph is garbage and
# later stages may fail — only the node_upsert / edge_upsert times of the first build are meaningful)
cd $GRAFT_REPO_ROOT
make -C amira_amd/csrc clean > /dev/null
make -C amira_amd/csrc -j32 CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-result -Wno-unused-function -DAMG_EXPERIMENTS=0 -DAMG_ABLATE_NOPROBE" 2>&1 | grep -E "error"
python3 - <<'PY'
import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, bench
from amira_amd import Engine
w = bench.WORKLOADS["cfg3"]
vocab, toks, offs = bench.make_tokens(w, 0, w["N"])
eng = Engine(0)
eng.set_reads(toks, offs, vocab.two_v)
for rep in range(3):
    try:
        eng.build(w["k"])
    except Exception as e:
        print("build:", str(e)[:80])
    print({n: round(m, 3) for n, m in eng.timings()})
PY
