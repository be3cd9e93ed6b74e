"""debug: per-node sketch sizes, device (amg_path_sketch_overlaps on one-node paths) vs the objects' way"""
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
g.filter_graph(3, 1)
calls, pos = g.correct_reads(fq)
g = GeneMerGraph(calls, k, pos)
v = g._v()
alive = np.flatnonzero(v.arrays["nodes"]["alive"])
print("nodes", len(alive), flush=True)
_, seqs, row_of, _ = _sequences_for(fq, 0)
rows = np.fromiter((row_of.get(r, -1) for r in g._read_ids), np.int32, len(g._read_ids))
print("rows identity", bool((rows == np.arange(len(rows))).all()), len(rows), seqs.n, flush=True)
path_off = np.arange(len(alive) + 1, dtype=np.int64)
size, _ = g._engine.path_sketch_overlaps(seqs, rows, 11, 10, path_off, alive.astype(np.int32), [], [])
print("device done", flush=True)
tok_node = g._engine.read_node_ids()
gs, ge = g._gs, g._ge
offs = g._read_off
bad = 0
for n_i, node in enumerate(alive.tolist()):
    h = v.hash_at(node)
    mh = {}
    g.get_minhash_of_nodes([h], mh, fq)
    want = len(mh[h])
    if want != int(size[n_i]):
        bad += 1
        if bad <= 3:
            ws = np.flatnonzero(tok_node == node)
            segs = []
            for w in ws.tolist():
                r = int(np.searchsorted(offs, w, side="right") - 1)
                segs.append(fq[g._read_ids[r]]["sequence"][int(gs[w]):int(ge[w + k - 1]) + 1])
            emu = g._engine.minhash(segs, [0] * len(segs), 11, 10)[0]
            obj_segs = g._node_segments(h, fq)
            print("node", node, "object", want, "device", int(size[n_i]), "emulated", len(emu), "windows", len(ws),
                  "object segments", len(obj_segs), "same segments", sorted(segs) == sorted(obj_segs), flush=True)
            print("  lens emu", sorted(len(s) for s in segs)[:10], "obj", sorted(len(s) for s in obj_segs)[:10])
print("differing nodes", bad, "of", len(alive))

# the same reads handed over as host dicts (positions uploaded, not gathered on the device)
calls2 = {r: list(v) for r, v in calls.items()}
pos2 = {r: [tuple(p) for p in v] for r, v in pos.items()}
g2 = GeneMerGraph(calls2, k, pos2)
v2 = g2._v()
alive2 = np.flatnonzero(v2.arrays["nodes"]["alive"])
rows2 = np.fromiter((row_of.get(r, -1) for r in g2._read_ids), np.int32, len(g2._read_ids))
size2, _ = g2._engine.path_sketch_overlaps(seqs, rows2, 11, 10, np.arange(len(alive2) + 1, dtype=np.int64), alive2.astype(np.int32), [], [])
bad2 = 0
for n_i, node in enumerate(alive2.tolist()):
    h = v2.hash_at(node)
    mh = {}
    g2.get_minhash_of_nodes([h], mh, fq)
    bad2 += len(mh[h]) != int(size2[n_i])
print("from host dicts: differing nodes", bad2, "of", len(alive2))
print("host positions equal", np.array_equal(g._gs, g2._gs), np.array_equal(g._ge, g2._ge))
# scaled 1: every k-mer
size3, _ = g2._engine.path_sketch_overlaps(seqs, rows2, 11, 1, np.arange(len(alive2) + 1, dtype=np.int64), alive2.astype(np.int32), [], [])
tok2 = g2._engine.read_node_ids()
for n_i, node in enumerate(alive2.tolist()[:5]):
    ws = np.flatnonzero(tok2 == node)
    segs = []
    for w in ws.tolist():
        r = int(np.searchsorted(g2._read_off, w, side="right") - 1)
        segs.append(fq[g2._read_ids[r]]["sequence"][int(g2._gs[w]):int(g2._ge[w + k - 1]) + 1])
    emu = g2._engine.minhash(segs, [0] * len(segs), 11, 1)[0]
    print("scaled 1 node", node, "device", int(size3[n_i]), "emulated", len(emu), "kmers", sum(max(0, len(s) - 10) for s in segs))
