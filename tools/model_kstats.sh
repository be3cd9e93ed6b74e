#!/bin/bash
# usage: tools/model_kstats.sh <tag> <world> — per-kernel totals (rocprofv3 --kernel-trace --stats) of the scaling
# model's emulated sweep at ONE world size: which kernels grow with the rank count
tag=$1; W=$2
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/mk_$tag -o out -- python3 $R/tools/scaling_model.py 1000000 $W > $R/gpurun_out/mk_$tag.json 2> $R/gpurun_out/mk_$tag.err
cd $R
f=$(find gpurun_out/mk_$tag -name "*kernel_stats.csv" | head -1)
python3 - "$f" > gpurun_out/mk_$tag.txt <<'PY'
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
print("# kernel, calls, total us, mean us   (whole run: yardstick sweeps x3, model warm-up + measured step)")
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:60]:
    name = re.sub(r"\(.*", "", r["Name"]).replace("void ", "")[:90]
    print(f"{float(r['TotalDurationNs'])/1e3:12.1f} us  x{int(r['Calls']):6d}  mean {float(r['AverageNs'])/1e3:9.1f}  {name}")
PY
rm -rf gpurun_out/mk_$tag
head -45 gpurun_out/mk_$tag.txt
