#!/bin/bash
# usage: tools/ctr_probe.sh — is the single claim counter (one returning atomicAdd per workgroup that creates keys,
# 58 k of them on one word in a first build) what the first-build table passes wait for?  Four variant builds:
# {node pass, edge pass} x {a second returning atomic on the same word, 64 counters with disjoint claim ranges}.
# Mode 2 leaves garbage counts: the build stops after the pass, only that pass's time is meaningful.
cd $GRAFT_REPO_ROOT
for which in 1 2; do for mode in 1 2; do
make -C amira_amd/csrc clean > /dev/null
make -C amira_amd/csrc -j32 CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-result -Wno-unused-function -DAMG_EXPERIMENTS=0 -DAMG_EXP_CTR=$which -DAMG_EXP_MODE=$mode" 2>&1 | grep -E "error"
echo "== pass $which (1 nodes, 2 edges) mode $mode (1 extra atomic, 2 sharded counters)"
timeout 300 python3 - <<'PY'
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
    t = dict(eng.timings())
    print({n: round(t[n], 3) for n in ("node_upsert", "edge_upsert_head", "edge_upsert") if n in t})
PY
done; done
make -C amira_amd/csrc clean > /dev/null
