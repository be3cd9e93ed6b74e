"""bubble popping against the pure-Python oracle at a size with hundreds of junctions (one-off; PYTHONHASHSEED=0)
usage: bubbles_vs_oracle.py SEED N L V K ERR"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import json
import procedures as P
import seed0

seed, N, L, V, k = (int(x) for x in sys.argv[1:6])
err = float(sys.argv[6])
assert os.environ.get("PYTHONHASHSEED") == "0", "the reference's set order: run with PYTHONHASHSEED=0"
out = {}
for kind in ("product", "oracle"):
    impl = seed0._impl(kind)
    t = time.time()
    out[kind] = json.loads(json.dumps(P.p_bubbles_random(impl, seed, N, L, V, k, err)))
    print(kind, round(time.time() - t, 1), "s; junctions", sum(len(v) for v in out[kind]["starts"].values()),
          "unique paths", sum(c["n_unique"] for c in out[kind]["paths"]), "genes out", out[kind]["n_genes"], flush=True)
print("EQUAL" if out["product"] == out["oracle"] else "DIFFERENT")
