"""BASELINE config 4 probe: planted multi-copy AMR genes, build + read-path clustering through
the reference-compatible Python API (device build + K6 matching, host anchor/block logic)."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from amira_amd import GeneMerGraph, synth

N = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
L, V, k, seed = 60, 20_000, 5, 20250905 + 4
t = time.perf_counter()
ids, sts = synth.block_reads(seed, 0, N, L, V, 0.0, n_amr=10)
names = synth.gene_names(V, 10)
reads = synth.to_read_dict(ids, sts, names)
pos = synth.positions_for(reads)
t_gen = time.perf_counter() - t
t = time.perf_counter()
g = GeneMerGraph(reads, k, pos)
t_build = time.perf_counter() - t
c = g._engine.counts()
t = time.perf_counter()
clustered, path_reads = g.assign_reads_to_genes([f"amr{j}" for j in range(10)], 1, {}, None)
t_cluster = time.perf_counter() - t
alleles = {gene: sorted(len(v) for v in d.values()) for comp in clustered.values() for gene, d in comp.items()}
print(json.dumps({"N": N, "gen_s": round(t_gen, 1), "build_incl_tokenize_s": round(t_build, 2),
                  "device_build_ms": round(sum(m for _, m in g._engine.timings()), 2), "nodes": c["n_nodes"],
                  "cluster_s": round(t_cluster, 2), "alleles": alleles}))
