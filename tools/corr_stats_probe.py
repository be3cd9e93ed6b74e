"""How many reads each correction of the cfg3 sweep changes (changed flag of the corrected set)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from amira_amd import Engine
w = bench.WORKLOADS["cfg3-sweep"]; N, L, k = w["N"], w["L"], w["k"]
vocab, toks, offs = bench.make_tokens(w, 0, N)
gs = np.tile(np.arange(L, dtype=np.int64) * 1000, N)
eng = Engine(0)
eng.set_reads(toks, offs, vocab.two_v); eng.set_positions(gs, gs + 899, np.full(N, L * 1000 + 100, np.int64))
eng.build(k); eng.filter(3, 1)
nr, nt = eng.correct_reads(); out = eng.corrected(nr, nt, False)
print("correction 1: reads out", nr, "changed", int(out["changed"].sum()), "tokens", nt)
eng.adopt_corrected(); eng.build(k); eng.remove_short_linear_paths(k)
nr2, nt2 = eng.correct_reads(); out = eng.corrected(nr2, nt2, False)
print("correction 2: reads out", nr2, "changed", int(out["changed"].sum()), "tokens", nt2, "dropped", nr - nr2)
