"""probe: the stages of bench.py's json_e2e (process_pandora_json -> cleaning_sweep -> write_pandora_gene_calls), three runs"""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import bench
from amira_amd import graph_utils as gu, synth
from amira_amd.io import ReadLengths, write_gene_calls, write_gene_positions
from amira_amd.pre_processing import process_pandora_json
from amira_amd.result_utils import write_pandora_gene_calls
w = bench.WORKLOADS["cfg3-sweep"]
vocab, toks, offs = bench.make_tokens(w, 0, w["N"])
N, L, k = w["N"], w["L"], w["k"]
ids = synth.read_names(0, N)
gs = np.tile(np.arange(L, dtype=np.int64) * 1000, N); ge = gs + 899
base = os.environ.get("AMG_BENCH_TMP", "/dev/shm")
with tempfile.TemporaryDirectory(dir=base) as d:
    cj, pj, cj2, pj2 = (os.path.join(d, n) for n in ("c.json", "p.json", "c2.json", "p2.json"))
    write_gene_calls(cj, vocab, toks, offs, ids); write_gene_positions(pj, gs, ge, offs, ids)
    lengths = np.full(N, L * 1000 + 100, np.int64)
    wanted = [vocab.names[i] for i in range(0, vocab.V, max(vocab.V // 40, 1))] + ["not_in_the_reads"]
    for rep in range(4):
        T = [time.perf_counter()]
        r, genes, p = process_pandora_json(cj, wanted, pj); T.append(time.perf_counter())
        g, r2, p2 = gu.cleaning_sweep(r, p, k, ReadLengths(r.read_ids, lengths), 3); T.append(time.perf_counter())
        write_pandora_gene_calls(d, p2, r2, cj2, pj2); T.append(time.perf_counter())
        n = g.get_total_number_of_nodes(); g.close(); T.append(time.perf_counter())
        print("run", rep, "total %.3f" % (T[-1] - T[0]), "load %.3f sweep %.3f write %.3f nodes+close %.3f" % tuple(T[i + 1] - T[i] for i in range(4)), flush=True)
