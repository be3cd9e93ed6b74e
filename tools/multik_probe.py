"""usage: python tools/multik_probe.py [N] — f3 (graph_utils.py:258-296 choose_kmer_size): the seven graphs k = 3, 5, .., 15 of
one read set as ONE amg_build_multi call (two passes over the tokens in all, fingerprint keys) against seven amg_build
calls (each its own two passes; exact keys where the tuple fits), and the hybrid: exact builds for the k that fit +
one multi call for the rest.  cfg 3 stream."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from amira_amd import Engine

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
w = dict(bench.WORKLOADS["cfg3"], N=N)
vocab, toks, offs = bench.make_tokens(w, 0, N)
ks = list(range(3, 16, 2))
engines = [Engine(0) for _ in ks]
engines[0].set_reads(toks, offs, vocab.two_v)


def sync_all():
    for e in engines:
        e.sync()


def timed(fn, reps=4):
    fn(); sync_all()
    t = time.perf_counter()
    for _ in range(reps):
        fn()
    sync_all()
    return (time.perf_counter() - t) / reps * 1e3


def multi(which):
    Engine.build_multi([engines[i] for i in which], [ks[i] for i in which])


# every engine its own copy of the reads for the single builds
for e in engines[1:]:
    e.set_reads(toks, offs, vocab.two_v)
out = {"reads": N, "ks": ks}
out["build_many_ms"] = round(timed(lambda: multi(range(len(ks)))), 3)
nodes_multi = [e.counts()["n_nodes"] for e in engines]
per_k = []
for i in range(len(ks)):
    per_k.append(round(timed(lambda i=i: engines[i].build(ks[i])), 3))
out["single_builds_ms"] = per_k
out["seven_builds_ms"] = round(sum(per_k), 3)
nodes_single = [e.counts()["n_nodes"] for e in engines]
exact = [i for i in range(len(ks)) if engines[i].counts()["exact_keys"]]
rest = [i for i in range(len(ks)) if i not in exact]
out["exact_ks"] = [ks[i] for i in exact]


def hybrid():
    for i in exact:
        engines[i].build(ks[i])
    if rest:
        # the multi call reads the first engine's reads: give the group its own leader
        Engine.build_multi([engines[i] for i in rest], [ks[i] for i in rest])


out["hybrid_ms"] = round(timed(hybrid), 3)
out["nodes"] = nodes_single
out["same_node_counts"] = nodes_single == nodes_multi
out["tokens_read"] = {"build_many": "2 x T", "seven_builds": "14 x T", "hybrid": f"{2 * len(exact) + 2} x T"}
print(json.dumps(out))
