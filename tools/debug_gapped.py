import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)
import numpy as np
import procedures as P
from amira_amd import Engine, tokenize
from test_gpu_sweep import flat_positions

name, k = sys.argv[1], int(sys.argv[2])
calls, pos = P.fixture(name)
vocab, toks, offs, read_ids = tokenize(calls)
def run(nofast):
    os.environ["AMG_NO_FAST_NW"] = "1" if nofast else "0"
    eng = Engine(0)
    eng.set_reads(toks, offs, vocab.two_v)
    gs, ge = flat_positions(read_ids, calls, pos)
    eng.set_positions(gs, ge, np.asarray([pos[r][-1][1] + 200 if pos[r] else 100 for r in read_ids], np.int64))
    eng.build(k); eng.filter(3, 1)
    nr, nt = eng.correct_reads()
    out = eng.corrected(nr, nt, True)
    tn, td = eng.read_nodes()
    return out, tn, td
a, tn, td = run(True)
b, _, _ = run(False)
print("reads", len(a["orig_read"]), len(b["orig_read"]))
for i in range(len(a["orig_read"])):
    x = a["tokens"][a["read_offsets"][i]:a["read_offsets"][i+1]]
    y = b["tokens"][b["read_offsets"][i]:b["read_offsets"][i+1]]
    gx = a["gene_start"][a["read_offsets"][i]:a["read_offsets"][i+1]]; gy = b["gene_start"][b["read_offsets"][i]:b["read_offsets"][i+1]]
    ex = a["gene_end"][a["read_offsets"][i]:a["read_offsets"][i+1]]; ey = b["gene_end"][b["read_offsets"][i]:b["read_offsets"][i+1]]
    if len(x) != len(y) or (x != y).any() or (gx != gy).any() or (ex != ey).any():
        r = a["orig_read"][i]
        print(" orig tokens:", toks[offs[r]:offs[r+1]].tolist())
        print(" slow pos:", list(zip(gx.tolist(), ex.tolist()))); print(" fast pos:", list(zip(gy.tolist(), ey.tolist())))
        r = a["orig_read"][i]
        print("DIFF read", i, r, read_ids[r], "len slow/fast", len(x), len(y), "orig len", offs[r+1]-offs[r])
        w = tn[offs[r]:offs[r+1]-k+1]
        print(" windows:", w.tolist())
        print(" slow:", x.tolist()); print(" fast:", y.tolist())
        break
else:
    print("identical")
