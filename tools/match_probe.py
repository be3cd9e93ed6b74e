"""where a k_match call spends its time at cfg 4's size: the two ctypes calls of Engine.match_patterns timed apart,
next to the device stages of the call (usage: match_probe.py [N])"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from amira_amd import Engine, _ffi
from amira_amd._ffi import check, ptr
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
w = dict(bench.WORKLOADS["cfg4"], N=N)
vocab, toks, offs = bench.make_tokens(w, 0, N)
e = Engine(0)
e.set_reads(toks, offs, vocab.two_v)
e.build(w["k"])
e.set_timing(True)
rng = np.random.default_rng(1)
for which, label in ((0, "gene lists"), (1, "one-node patterns")):
    for rep in range(3):
        if which == 0:
            starts = rng.integers(0, N, 50)
            pats = [toks[offs[r] + 5: offs[r] + 5 + 20].tolist() for r in starts]
        else:
            pats = [[int(i)] for i in rng.integers(0, e.graph_sizes()[0], 120)]
        n = len(pats)
        po = np.zeros(n + 1, np.int64); np.cumsum([len(p) for p in pats], out=po[1:])
        flat = np.fromiter((x for p in pats for x in p), dtype=np.int32, count=int(po[-1]))
        ho = np.zeros(n + 1, np.int64)
        t0 = time.perf_counter()
        check(_ffi.lib.amg_match_patterns(e._h, which, ptr(flat), ptr(po), n, ptr(ho), None, None))
        t1 = time.perf_counter()
        total = int(ho[-1]); hr, hp = np.empty(total, np.int32), np.empty(total, np.int32)
        check(_ffi.lib.amg_match_patterns(e._h, which, ptr(flat), ptr(po), n, ptr(ho), ptr(hr), ptr(hp)))
        t2 = time.perf_counter()
        st = e.timings()
        print(f"{label}: search {1e3 * (t1 - t0):.2f} ms, hand-out {1e3 * (t2 - t1):.2f} ms, hits {total}, device stages {st}")
# a second engine in the same process (kernels loaded already): what its first searches cost
e2 = Engine(0)
e2.set_reads(toks, offs, vocab.two_v)
t0 = time.perf_counter(); e2.build(w["k"]); t1 = time.perf_counter()
print(f"second engine: build {1e3 * (t1 - t0):.2f} ms")
for rep in range(2):
    starts = rng.integers(0, N, 50)
    pats = [toks[offs[r] + 5: offs[r] + 5 + 20].tolist() for r in starts]
    t0 = time.perf_counter(); e2.match_patterns(0, pats); t1 = time.perf_counter()
    pats1 = [[int(i)] for i in rng.integers(0, e2.graph_sizes()[0], 120)]
    e2.match_patterns(1, pats1); t2 = time.perf_counter()
    print(f"second engine, round {rep}: gene lists {1e3 * (t1 - t0):.2f} ms, one-node patterns {1e3 * (t2 - t1):.2f} ms")
