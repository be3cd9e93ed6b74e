#!/bin/bash
# usage: tools/kstats.sh <tag> [bench args...] — rocprofv3 kernel-trace stats of one bench run
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ks_$tag -o out -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > $R/gpurun_out/ks_$tag.log 2>&1
cd $R
f=$(find gpurun_out/ks_$tag -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time {tot/1e6:.2f} ms over the run")
for r in rows[:28]:
    name = re.sub(r"\(.*", "", r["Name"]).replace("void ", "")[:60]
    print(f"{name:60s} calls {int(r['Calls']):5d} total {float(r['TotalDurationNs'])/1e6:8.3f} ms avg {float(r['AverageNs'])/1e3:9.1f} us {float(r['Percentage']):5.1f}%")
PY
