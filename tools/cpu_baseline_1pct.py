"""usage: python tools/cpu_baseline_1pct.py > profiles/r4_cpu_baseline_python_1pct.json
SURVEY 8(d)'s CPU yardstick at the sample it names: the pure-Python oracle (the reference's cost model, sha256 + pickle
per call) on the first 1 % of the cfg 3 stream (10 000 reads), the whole cleaning sweep, one core (~2 minutes).  The
default bench line keeps its ~12 s sample; this is the one committed run at the full 1 %."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
out = bench.cpu_baseline(bench.WORKLOADS["cfg3-sweep"], frac=0.01, budget_s=1e9)
out["host_cores"] = os.cpu_count()
print(json.dumps(out, indent=1))
