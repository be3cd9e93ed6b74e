"""probe: bench.py's front_end leg alone (twice), with the native loader's / writers' own phase times (AMG_CALLS_TIMING=1)"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
w = bench.WORKLOADS["cfg3-sweep"]
vocab, toks, offs = bench.make_tokens(w, 0, w["N"])
for rep in range(2):
    t = time.perf_counter()
    out = bench.run_front_end(w, vocab, toks, offs, w["k"], w["N"] * (w["L"] - w["k"] + 1))
    print("leg", round(time.perf_counter() - t, 2), json.dumps({k: v for k, v in out.items() if k != "json_e2e"}), out["json_e2e"]["s_per_step"], flush=True)
