"""the node table of every build of the cfg 3 sweep: slots and retries (the rebuilds' tables are sized by the bound the
correction hands on, amg_correct_reads / amg_adopt_corrected)"""
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch, bench
from amira_amd import Engine
w = bench.WORKLOADS["cfg3-sweep"]; N = 1_000_000; L, k = w["L"], w["k"]
vocab, toks, offs = bench.make_tokens(w, 0, N)
gs = np.tile(np.arange(L, dtype=np.int64) * 1000, N); ge = gs + 899; rl = np.full(N, L * 1000 + 100, np.int64)
e = Engine(0)
e.set_reads(toks, offs, vocab.two_v); e.set_positions(gs, ge, rl)
e.build(k); c = e.counts(); print("build1 nodes", e.graph_sizes()[0], "slots", c["node_table_slots"], "retries", c["build_retries"])
e.filter(3, 1); e.correct_reads(); e.adopt_corrected(); e.build(k); c = e.counts(); print("build2 nodes", e.graph_sizes()[0], "slots", c["node_table_slots"], "retries", c["build_retries"])
e.remove_short_linear_paths(k); e.correct_reads(); e.adopt_corrected(); e.build(k); c = e.counts(); print("build3 nodes", e.graph_sizes()[0], "slots", c["node_table_slots"], "retries", c["build_retries"])
