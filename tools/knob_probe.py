import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from amira_amd import Engine
w = bench.WORKLOADS["cfg3"]
vocab, toks, offs = bench.make_tokens(w, 0, w["N"])
eng = Engine(0)
eng.set_reads(toks, offs, vocab.two_v)
for it in range(3):
    try:
        eng.build(5)
    except Exception as e:
        pass
    print(os.environ.get("AMG_KNOB"), {n: round(m, 3) for n, m in eng.timings() if n in ("node_upsert", "node_table_clear")})
