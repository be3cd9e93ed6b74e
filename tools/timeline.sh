#!/bin/bash
# usage: tools/timeline.sh <tag> [bench args...] — every dispatch of ONE steady-state step of bench.py in stream order
# (name, start offset, duration, idle gap before it), from a rocprofv3 kernel trace; plus the per-name totals of that step
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tl_$tag -o out -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-e2e --no-fused-line "$@" > $R/gpurun_out/tl_$tag.log 2>&1
cd $R
f=$(find gpurun_out/tl_$tag -name "*kernel_trace.csv" | head -1)
python3 - "$f" > gpurun_out/tl_$tag.txt <<'PY'
import csv, sys, re
from collections import defaultdict
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")) for r in rows))
# a step starts with k_read_stats preceded by the clear of a build; steps = groups of three builds: find the
# k_read_stats launches and take the last full step (3 per step in the sweep, 1 in a build-only workload)
starts = [i for i, e in enumerate(ev) if e[2].startswith("k_read_stats")]
per = 3 if len(starts) >= 6 and len(starts) % 3 == 0 else 1
lo = starts[-per] - 1            # the clear launch before it
hi = len(ev)
# the step before the last one is bounded by the next step's start
if len(starts) >= 2 * per:
    lo, hi = starts[-2 * per] - 1, starts[-per] - 1
step = ev[lo:hi]
t0 = step[0][0]
busy = sum(e[1] - e[0] for e in step)
print(f"# {len(step)} dispatches, span {(step[-1][1] - t0) / 1e6:.3f} ms, busy {busy / 1e6:.3f} ms")
prev_end = t0
tot = defaultdict(lambda: [0, 0])
for s, e, n in step:
    print(f"{(s - t0) / 1e3:10.1f} us  dur {(e - s) / 1e3:8.1f}  gap {(s - prev_end) / 1e3:7.1f}  {n[:90]}")
    prev_end = max(prev_end, e)
    tot[n[:70]][0] += e - s
    tot[n[:70]][1] += 1
print("# per kernel")
for n, (d, c) in sorted(tot.items(), key=lambda kv: -kv[1][0]):
    print(f"{d / 1e3:10.1f} us  x{c:4d}  {n}")
small = [(e - s) for s, e, n in step if e - s < 20000]
print(f"# dispatches under 20 us: {len(small)}, {sum(small) / 1e6:.3f} ms")
PY
rm -rf gpurun_out/tl_$tag
tail -5 gpurun_out/tl_$tag.txt
