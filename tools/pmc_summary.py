"""Summarise rocprofv3 --pmc results (rocpd sqlite `counters_collection` view or csv): per kernel
(short name) and counter, the mean over launches of the per-dispatch SUM over counter instances.
usage: pmc_summary.py DIR [DIR...]"""
import glob, re, sqlite3, sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(lambda: defaultdict(float)))   # kernel -> counter -> dispatch -> sum
meta, dur = {}, defaultdict(dict)
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*.db", recursive=True):
        cur = sqlite3.connect(f).cursor()
        q = ("select kernel_name, dispatch_id, counter_name, value, vgpr_count, sgpr_count, lds_block_size, "
             "workgroup_size, duration from counters_collection")
        for name, disp, cn, val, vg, sg, lds, wg, du in cur.execute(q):
            name = re.sub(r"\(.*", "", name).replace("void ", "")
            if not name.startswith("k_"):
                continue
            acc[name][cn][(f, disp)] += float(val)
            meta[name] = (vg, sg, lds, wg)
            dur[name][(f, disp)] = du
for name in sorted(acc):
    v, s, l, w = meta[name]
    n = len(dur[name])
    mean_us = sum(dur[name].values()) / n / 1e3
    print(f"{name} [dispatches {n}, mean {mean_us:.1f} us (under pmc), vgpr {v}, sgpr {s}, lds {l}, wg {w}]")
    for cn, per in sorted(acc[name].items()):
        vals = list(per.values())
        print(f"    {cn:22s} mean {sum(vals)/len(vals):14.5g}   first {vals[0]:14.5g}")
