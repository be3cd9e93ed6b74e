import sys, time, json
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from amira_amd import Engine
w = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "cfg3"]
N = int(sys.argv[2]) if len(sys.argv) > 2 else w["N"]
vocab, toks, offs = bench.make_tokens(w, 0, N)
L, k = w["L"], w["k"]
gs = np.tile(np.arange(L, dtype=np.int64) * 1000, N); ge = gs + 899
rl = np.full(N, L * 1000 + 100, dtype=np.int64)
eng = Engine(0)
for it in range(3):
    tt = {}
    t0 = time.perf_counter()
    eng.set_reads(toks, offs, vocab.two_v); eng.set_positions(gs, ge, rl)
    t1 = time.perf_counter()
    def stage(name, fn):
        t = time.perf_counter(); r = fn(); eng.sync(); tt[name] = (time.perf_counter() - t) * 1e3
        tt[name + "_stages"] = {n: round(m, 3) for n, m in eng.timings()}
        return r
    stage("build1", lambda: eng.build(k)); c1 = eng.counts()
    stage("filter", lambda: eng.filter(3, 1))
    stage("correct1", lambda: eng.correct_reads())
    stage("adopt1", lambda: eng.adopt_corrected())
    stage("build2", lambda: eng.build(k)); c2 = eng.counts()
    rem = stage("clip", lambda: eng.remove_short_linear_paths(k))
    stage("correct2", lambda: eng.correct_reads())
    stage("adopt2", lambda: eng.adopt_corrected())
    stage("build3", lambda: eng.build(k)); c3 = eng.counts()
    t2 = time.perf_counter()
    print(json.dumps({"upload_ms": (t1 - t0) * 1e3, "sweep_ms": (t2 - t1) * 1e3,
                      "nodes": [c1["n_nodes"], c2["n_nodes"], c3["n_nodes"]], "to_correct": c1["n_reads_to_correct"],
                      "removed": len(rem), "t": tt}))
