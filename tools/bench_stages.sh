#!/bin/bash
# usage: tools/bench_stages.sh <tag>  — sweep + build benches, stage table
tag=$1
timeout 150 python bench.py --steps 5 --warmup 1 2>&1 | tail -1 > gpurun_out/${tag}_sweep.json
timeout 150 python bench.py --workload cfg3 --steps 5 --warmup 1 2>&1 | tail -1 > gpurun_out/${tag}_build.json
python - <<PY
import json
for f in ("${tag}_sweep","${tag}_build"):
    d=json.load(open("gpurun_out/%s.json"%f)); print(f, round(d["ms_per_step"],3), d["stages_ms_per_step"])
PY
