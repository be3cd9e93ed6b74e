#!/bin/bash
# usage: tools/ab_sweep.sh <tag> ["ENV=1 ENV2=x" ...] — the cfg 3 sweep once per environment (A/B switches), one line each:
# ms per sweep and the per-stage HIP-event times of the instrumented step
tag=$1; shift
cd $GRAFT_REPO_ROOT
[ $# -eq 0 ] && set -- ""
for e in "$@"; do
  env $e timeout 200 python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-e2e --no-cfg4 --no-fused-line 2>/dev/null | tail -1 > gpurun_out/ab_${tag}.tmp
  python3 - "$e" gpurun_out/ab_${tag}.tmp >> gpurun_out/ab_${tag}.txt <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[2]))
    st = d.get("stages_ms_per_step", {})
    keep = {k: round(v, 3) for k, v in st.items() if v >= 0.05}
    print(f"[{sys.argv[1]}] ms_per_step {d['ms_per_step']:.3f} roofline {d['roofline']['kernel']} frac {d['roofline']['frac']:.3f}  {keep}")
except Exception as e:
    print(f"[{sys.argv[1]}] FAILED {e}: {open(sys.argv[2]).read()[:300]}")
PY
done
rm -f gpurun_out/ab_${tag}.tmp
cat gpurun_out/ab_${tag}.txt
