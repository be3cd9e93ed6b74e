"""usage: trace_window.py <kernel_trace.csv> <anchor kernel substring> [before_us after_us] — the dispatches around the LAST
occurrence of the anchor kernel, with durations and gaps (from a rocprofv3 --kernel-trace csv)"""
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
anchor = sys.argv[2]
before = float(sys.argv[3]) if len(sys.argv) > 3 else 3000.0
after = float(sys.argv[4]) if len(sys.argv) > 4 else 1000.0
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if anchor in r["Kernel_Name"]]
if not idx:
    sys.exit("anchor not found")
t_a = int(rows[idx[-1]]["Start_Timestamp"])
prev_end = None
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if s < t_a - before * 1e3 or s > t_a + after * 1e3:
        continue
    name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")[:70]
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    print(f"{(s - t_a) / 1e3:10.1f} us  dur {(e - s) / 1e3:8.1f}  gap {gap:7.1f}  {name}")
    prev_end = e
