"""FETCH_SIZE / WRITE_SIZE per launch of our kernels from two rocprofv3 --pmc passes (rocpd sqlite):
usage: pmc_fetch_write.py FETCH_DIR WRITE_DIR  -> JSON on stdout (the format bench.py reads)"""
import glob, json, re, sqlite3, sys
from collections import defaultdict


def per_launch(d, counter):
    out = defaultdict(lambda: defaultdict(float))
    order = {}
    for f in glob.glob(d + "/**/*.db", recursive=True):
        cur = sqlite3.connect(f).cursor()
        for name, disp, cn, val in cur.execute(
                "select kernel_name, dispatch_id, counter_name, value from counters_collection"):
            if cn != counter:
                continue
            name = re.sub(r"\(.*", "", name).replace("void ", "")
            if not name.startswith("k_"):
                continue
            out[name][disp] += float(val)
    return {k: [v[d_] for d_ in sorted(v)] for k, v in out.items()}


fetch = per_launch(sys.argv[1], "FETCH_SIZE")
write = per_launch(sys.argv[2], "WRITE_SIZE")
rows = []
for k in sorted(fetch, key=lambda k: -sum(fetch[k])):
    rows.append({"kernel": k, "launches": len(fetch[k]),
                 "FETCH_SIZE_KB_per_launch": [round(x) for x in fetch[k][:3]],
                 "WRITE_SIZE_KB_per_launch": [round(x) for x in write.get(k, [])[:3]],
                 "FETCH_SIZE_KB_mean": round(sum(fetch[k]) / len(fetch[k])),
                 "WRITE_SIZE_KB_mean": round(sum(write.get(k, [0])) / max(len(write.get(k, [0])), 1))})
print(json.dumps({
    "note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on `bench.py --steps 1 --warmup 0`; values in KB "
            "as reported (gfx950: FETCH_SIZE may read half of wide coalesced streams; uncorrected). The first three "
            "launches of a build kernel = the three builds of the instrumented sweep.",
    "kernels": rows}, indent=1))
