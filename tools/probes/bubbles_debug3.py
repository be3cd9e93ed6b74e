"""debug: the intermediate arrays of amg_path_sketch_overlaps"""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import numpy as np
import procedures as P
from amira_amd import GeneMerGraph, synth, _ffi
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
size, _ = g._engine.path_sketch_overlaps(seqs, None, 11, 10, po, alive.astype(np.int32), [], [])
fn = _ffi.lib.amg_bubbles_debug_copy
fn.restype, fn.argtypes = C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int64]
def grab(which, n, dtype):
    a = np.empty(n, dtype)
    _ffi.check(fn(g._engine._h, which, a.ctypes.data_as(C.c_void_p), a.nbytes))
    return a
tok = g._engine.read_node_ids()
n_segs = int((tok >= 0).sum())   # every live node is in a path here
segs = grab(0, n_segs * 2, np.int64).reshape(-1, 2)
src, lennode = segs[:, 0], segs[:, 1]
slen, snode = (lennode & 0xffffffff).astype(np.int64), (lennode >> 32).astype(np.int64)
ws = np.flatnonzero(tok >= 0)
offs = g._read_off
seq_off = np.zeros(len(g._read_ids) + 1, np.int64)
np.cumsum([len(fq[r]["sequence"]) for r in fq], out=seq_off[1:])
rows = np.searchsorted(offs, ws, side="right") - 1
a, b = g._gs[ws].astype(np.int64), g._ge[ws + k - 1].astype(np.int64) + 1
Ls = seq_off[rows + 1] - seq_off[rows]
x, y = np.minimum(a, Ls), np.minimum(b, Ls)
want = sorted(zip((seq_off[rows] + x).tolist(), np.maximum(y - x, 0).tolist(), tok[ws].tolist()))
got = sorted(zip(src.tolist(), slen.tolist(), snode.tolist()))
print("segments", n_segs, "equal", want == got, flush=True)
D = len(v.arrays["nodes"]["alive"])
noff = grab(7, D + 1, np.int64)
print("noff last", int(noff[-1]), "paths", len(alive))
nl = grab(8, int(noff[-1]), np.int32)
print("nlist is a permutation of the paths", sorted(nl.tolist()) == list(range(len(alive))))
pstart = grab(9, len(alive) + 1, np.int64)
M = int(pstart[-1])
print("M", M, "pstart monotone", bool((np.diff(pstart) >= 0).all()))
out_h, srt_h, sp, si, h2, srt_p = grab(1, M, np.uint64), grab(2, M, np.uint64), grab(3, M, np.uint32), grab(4, M, np.uint32), grab(5, M, np.uint64), grab(6, M, np.uint32)
print("sort 1: keys ascending", bool((np.diff(srt_h.astype(np.float64)) >= 0).all()), "same multiset", np.array_equal(np.sort(out_h), srt_h))
print("sort 2: keys ascending", bool((np.diff(sp.astype(np.int64)) >= 0).all()), "srt_i a permutation", np.array_equal(np.sort(si), np.arange(M, dtype=np.uint32)),
      "keys follow", np.array_equal(srt_p[si], sp), "h2 = srt_h[si]", np.array_equal(srt_h[si], h2))
key = sp.astype(np.int64)
inorder = True
for p_ in range(len(alive)):
    seg = h2[pstart[p_]:pstart[p_ + 1]]
    if len(seg) > 1 and not (seg[1:] >= seg[:-1]).all():
        inorder = False
        break
print("hashes ascend inside every path", inorder)
dev_sets = {int(alive[p_]): set(h2[pstart[p_]:pstart[p_ + 1]].tolist()) for p_ in range(len(alive))}
print("sizes agree with the sets", all(len(dev_sets[int(alive[p_])]) == int(size[p_]) for p_ in range(len(alive))))
# expected sets from the segments on the host
bad = 0
for node in alive.tolist()[:40]:
    w = ws[tok[ws] == node]
    segs_ = []
    for t in w.tolist():
        r = int(np.searchsorted(offs, t, side="right") - 1)
        segs_.append(fq[g._read_ids[r]]["sequence"][int(g._gs[t]):int(g._ge[t + k - 1]) + 1])
    emu = g._engine.minhash(segs_, [0] * len(segs_), 11, 10)[0]
    if emu != dev_sets[node]:
        bad += 1
        print("node", node, "expected", len(emu), "device", len(dev_sets[node]), "missing", len(emu - dev_sets[node]), "extra", len(dev_sets[node] - emu))
print("nodes of the first 40 that differ", bad)
