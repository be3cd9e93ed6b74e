#!/bin/bash
# usage: tools/bucket_probe2.sh "<flags>" ... — node pass of the FIRST build on an error-free stream (every window hits an
# existing node after the first reads: what a rebuild looks like) and on the cfg 3 stream, per variant build of
# amg_build_x.o (flags: -DAMG_BUCKET_PROBES=n, -DAMG_M_DIR_LDS, -DAMG_ABLATE_NOPROBE, ...), buckets on and off
cd $GRAFT_REPO_ROOT
for flags in "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-result -Wno-unused-function -DAMG_EXPERIMENTS=0 $flags -c amira_amd/csrc/amg_build_x.hip -o amira_amd/csrc/amg_build_x.o 2>&1 | grep -E "error"
  make -C amira_amd/csrc > /dev/null 2>&1
  for nb in 0 1; do
  echo "== flags: [$flags] AMG_NODE_BUCKETS=$nb"
  AMG_NODE_BUCKETS=$nb timeout 300 python3 - <<'PY'
import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, bench
from amira_amd import Engine, synth
N, L, V = 1_000_000, 60, 20000
ids, sts = synth.block_reads(20250908, 0, N, L, V, 0.0)
toks = np.where(sts == 1, V + ids, V - 1 - ids).astype(np.int32).ravel()
offs = np.arange(0, (N + 1) * L, L, dtype=np.int64)
w = bench.WORKLOADS["cfg3"]
vocab, toks3, offs3 = bench.make_tokens(w, 0, w["N"])
eng = Engine(0)
for name, (t, o, tv) in (("error-free", (toks, offs, 2 * V)), ("cfg3", (toks3, offs3, vocab.two_v))):
    eng.set_reads(t, o, tv)
    out = []
    for rep in range(4):
        try:
            eng.build(5)
        except Exception as e:
            out.append("build: " + str(e)[:60])
        tm = dict(eng.timings())
        out.append({n: round(tm[n], 3) for n in ("node_upsert_head", "node_upsert") if n in tm})
    print(name, out[-2:])
PY
  done
done
