"""cProfile of bench.py's json_e2e leg (gene calls + positions JSON in -> cleaning_sweep -> JSON out) at cfg 3's size.
usage: python tools/json_e2e_profile.py [cumulative|tottime]"""
import cProfile, os, pstats, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from amira_amd import graph_utils as gu, synth
from amira_amd.io import ReadLengths, TokenizedPositions, load_gene_calls, write_gene_calls, write_gene_positions

w = bench.WORKLOADS["cfg3-sweep"]
vocab, toks, offs = bench.make_tokens(w, 0, w["N"])
N, L, k = w["N"], w["L"], w["k"]
ids = synth.read_names(0, N)
gs = np.tile(np.arange(L, dtype=np.int64) * 1000, N)
ge = gs + 899
with tempfile.TemporaryDirectory(dir=os.environ.get("TMPDIR", "/tmp")) as d:
    cj, pj, cj2, pj2 = (os.path.join(d, n) for n in ("c.json", "p.json", "c2.json", "p2.json"))
    write_gene_calls(cj, vocab, toks, offs, ids)
    write_gene_positions(pj, gs, ge, offs, ids)
    lengths = np.full(N, L * 1000 + 100, np.int64)

    def whole():
        r, s_, e_ = load_gene_calls(cj, pj)
        g, r2, p2 = gu.cleaning_sweep(r, TokenizedPositions(r.read_ids, r.read_offsets, s_, e_), k,
                                      ReadLengths(r.read_ids, lengths), 3)
        write_gene_calls(cj2, r2.vocab, r2.tokens, r2.read_offsets, r2.read_ids)
        write_gene_positions(pj2, p2.gene_start, p2.gene_end, p2.read_offsets, p2.read_ids)
        g.close()

    whole()
    t = time.perf_counter(); whole(); print("wall", round(time.perf_counter() - t, 3))
    pr = cProfile.Profile(); pr.enable(); whole(); pr.disable()
    pstats.Stats(pr).sort_stats(sys.argv[1] if len(sys.argv) > 1 else "tottime").print_stats(22)
