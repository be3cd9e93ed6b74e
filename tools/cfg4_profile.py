import cProfile, pstats, os, sys, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from amira_amd import GeneMerGraph, synth
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
ids, sts = synth.block_reads(20250909, 0, N, 60, 20000, 0.0, n_amr=10)
reads = synth.to_read_dict(ids, sts, synth.gene_names(20000, 10))
pos = synth.positions_for(reads)
g = GeneMerGraph(reads, 5, pos)
pr = cProfile.Profile(); pr.enable()
g.assign_reads_to_genes([f"amr{j}" for j in range(10)], 1, {}, None)
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(28); print(s.getvalue()[-4500:])
