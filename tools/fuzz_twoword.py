"""usage: python tools/fuzz_twoword.py SECONDS [SEED] — random read sets whose gene-mers need TWO-word exact keys
(k * bits > 63) with many keys that agree in their first 63 bits (the last canonical gene differs), built on the device
and compared with the sequential C oracle: nodes, coverages, first directions, edges, per-window ids.  Exercises the
slot protocol of amg_x.h (owner by the first key word, publication of the second, the lone continuation of a
half-equal key) with and without minimiser buckets, at table loads from sparse to crowded."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import token_oracle
from amira_amd import Engine

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
t_end = time.time() + budget
n_ok = 0
eng = Engine(0)
while time.time() < t_end:
    k = int(rng.choice([5, 5, 5, 7]))
    V = int(rng.choice([1 << 13, 20000, 30000, 1 << 16, 100000]))
    if k == 7 and V > 4000:          # 7 genes have to fit 94 bits: 13 bits each
        V = 4000
    genome = rng.integers(0, 2 * V, int(rng.choice([50, 400, 5000])))
    reads = []
    n_reads = int(rng.choice([2000, 20000, 60000]))
    p_tail = float(rng.choice([0.0, 0.3, 0.9]))     # how often a read is a family member [head..., X]
    heads = [rng.integers(0, 2 * V, k - 1) for _ in range(int(rng.integers(1, 8)))]
    for _ in range(n_reads):
        u = rng.random()
        if u < p_tail:
            h = heads[int(rng.integers(0, len(heads)))]
            r = np.concatenate([h, rng.integers(0, 2 * V, 1)])
            if rng.random() < 0.5:
                r = 2 * V - 1 - r[::-1]
        else:
            L = int(rng.integers(1, 40))
            s = int(rng.integers(0, len(genome)))
            r = np.take(genome, np.arange(s, s + L), mode="wrap").copy()
            e = rng.random(L) < 0.05
            r[e] = rng.integers(0, 2 * V, int(e.sum()))
            if rng.random() < 0.5:
                r = 2 * V - 1 - r[::-1]
        reads.append(r)
    toks = np.concatenate(reads).astype(np.int32)
    offs = np.zeros(len(reads) + 1, np.int64)
    np.cumsum([len(r) for r in reads], out=offs[1:])
    os.environ["AMG_NODE_BUCKETS"] = str(int(rng.integers(0, 2)))
    try:
        want = token_oracle.build(toks, offs, k, 2 * V)
    except AssertionError:
        continue                      # a palindromic window (even k only): nothing to compare
    for rep in range(2):
        eng.set_reads(toks, offs, 2 * V)
        eng.build(k)
        c = eng.counts()
        nodes, edges = eng.nodes(), eng.edges()
        tok_node, tok_dir = eng.read_nodes()
        ok = (c["n_windows"] == want["n_windows"] and np.array_equal(nodes["tokens"], want["tokens"])
              and np.array_equal(nodes["coverage"], want["coverage"]) and np.array_equal(nodes["first_dir"], want["first_dir"])
              and all(np.array_equal(edges[a], want[b]) for a, b in (("src", "src"), ("tgt", "tgt"), ("sdir", "sdir"),
                                                                        ("tdir", "tdir"), ("coverage", "ecov")))
              and np.array_equal(tok_node, want["tok_node"]) and np.array_equal(tok_dir, want["tok_dir"]))
        if not ok:
            dump = os.environ.get("FUZZ_DUMP", "gpurun_out/fuzz_twoword_fail.npz")
            np.savez_compressed(dump, toks=toks, offs=offs, k=k, two_v=2 * V)
            print(f"fuzz_twoword: MISMATCH (seed {seed}, case {n_ok}, k {k}, V {V}, buckets {os.environ['AMG_NODE_BUCKETS']}, "
                  f"exact {c['exact_keys']}); input saved to {dump}")
            sys.exit(1)
    n_ok += 1
eng.close()
print(f"fuzz_twoword: {n_ok} read sets equal (device vs C oracle, two builds each), 0 failures (seed {seed})")
