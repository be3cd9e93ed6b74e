"""debug: is amg_path_sketch_overlaps deterministic?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import numpy as np
import procedures as P
from amira_amd import GeneMerGraph, synth
from amira_amd.bubble_popping import _sequences_for

seed, N, L, V, k, err = 72, 400, 25, 90, 3, 0.05
ids, sts = synth.loop_reads(seed, N, L, V, err, 0)
calls = synth.to_read_dict(ids, sts, synth.gene_names(V, 0))
pos = {r: [(80 * i, 80 * i + 59) for i in range(len(g))] for r, g in calls.items()}
fq = P.synth_fastq(calls, pos, flank=40)
g = GeneMerGraph(calls, k, pos)
v = g._v()
alive = np.flatnonzero(v.arrays["nodes"]["alive"])
_, seqs, row_of, _ = _sequences_for(fq, 0)
po = np.arange(len(alive) + 1, dtype=np.int64)
for scaled in (10, 1):
    outs = []
    for rep in range(6):
        size, _ = g._engine.path_sketch_overlaps(seqs, None, 11, scaled, po, alive.astype(np.int32), [], [])
        outs.append(size.copy())
    print("scaled", scaled, "sum per call", [int(o.sum()) for o in outs], "calls equal to the first", [bool(np.array_equal(o, outs[0])) for o in outs], flush=True)
# one node per call
tok = g._engine.read_node_ids()
node = int(alive[0])
for rep in range(4):
    size, _ = g._engine.path_sketch_overlaps(seqs, None, 11, 10, np.asarray([0, 1], np.int64), np.asarray([node], np.int32), [], [])
    print("single node", node, int(size[0]), "windows", int((tok == node).sum()))
