"""Differential fuzzing of the merged (multi-GPU) build: W emulated ranks on one GPU (loop-back
exchange, uneven shards, some of them empty) against the unsharded engine — plain merge and
merge with the fused coverage filter.  usage: fuzz_dist.py SECONDS [SEED]"""
import os, sys, time, traceback, importlib.util
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import numpy as np
from amira_amd import Engine, tokenize
from amira_amd.dist import dist_build_loopback
spec = importlib.util.spec_from_file_location("fz", os.path.join(ROOT, "tools", "fuzz_sweep.py"))
fz = importlib.util.module_from_spec(spec); spec.loader.exec_module(fz)
spec = importlib.util.spec_from_file_location("td", os.path.join(ROOT, "tests", "test_gpu_dist.py"))
td = importlib.util.module_from_spec(spec); spec.loader.exec_module(td)


def run(budget, seed, max_cases=None):
    rng = np.random.default_rng(seed)
    t_end = time.time() + budget
    n_ok = n_skip = n_fail = 0
    while time.time() < t_end and (max_cases is None or n_ok + n_skip + n_fail < max_cases):
        reads, k, info = fz.make_case(rng)
        world = int(rng.choice([2, 3, 5, 8]))
        thr = None if rng.random() < 0.5 else (int(rng.choice([2, 3])), int(rng.choice([1, 2])))
        info.update(world=world, thr=thr)
        vocab, toks, offs, _ = tokenize(reads)
        R = len(offs) - 1
        cuts = np.sort(rng.integers(0, R + 1, world - 1))          # uneven shards, possibly empty
        bounds = [0] + cuts.tolist() + [R]
        ref = Engine(0); engines = []
        try:
            ref.set_reads(toks, offs, vocab.two_v)
            palindrome = False
            try:
                ref.build(k)
            except Exception as e:  # noqa: BLE001
                if getattr(e, "code", None) != -4:
                    raise
                palindrome = True   # a palindromic gene-mer (even k): every rank of the merged build must say so too
            if thr and not palindrome:
                ref.filter(*thr)
            for r in range(world):
                lo, hi = bounds[r], bounds[r + 1]
                e = Engine(0)
                e.set_reads(toks[offs[lo]:offs[hi]], offs[lo:hi + 1] - offs[lo], vocab.two_v)
                engines.append(e)
            try:
                dist_build_loopback(engines, k, *(thr or (1, 1)))
                assert not palindrome, "the unsharded build found a palindromic gene-mer, the merged one did not"
            except Exception as e:  # noqa: BLE001
                if getattr(e, "code", None) != -4 or not palindrome:
                    raise
                n_skip += 1
                continue
            if thr:
                want = td.live_state(ref)
                for r, e in enumerate(engines):
                    got = td.live_state(e); lo, hi = bounds[r], bounds[r + 1]
                    for key in ("tokens", "coverage", "first_dir", "src", "tgt", "sdir", "tdir", "ecov"):
                        assert np.array_equal(got[key], want[key]), key
                    assert got["adj"] == want["adj"]
                    assert np.array_equal(got["tok_node"], want["tok_node"][offs[lo]:offs[hi]])
                    assert np.array_equal(got["to_correct"], want["to_correct"][lo:hi])
            else:
                want = td.graph_state(ref); wn, wd = ref.read_nodes()
                for r, e in enumerate(engines):
                    td.assert_same_graph(td.graph_state(e), want); lo, hi = bounds[r], bounds[r + 1]
                    n_, d_ = e.read_nodes()
                    assert np.array_equal(n_, wn[offs[lo]:offs[hi]]) and np.array_equal(d_, wd[offs[lo]:offs[hi]])
            n_ok += 1
        except Exception:  # noqa: BLE001
            print("MISMATCH:", info, flush=True); traceback.print_exc(); n_fail += 1
        finally:
            for e in engines + [ref]:
                e.close()
        if n_fail >= 5:
            break
    print(f"fuzz_dist: {n_ok} merged builds equal to the unsharded one, {n_skip} palindromic inputs refused by both, {n_fail} failures (seed {seed})")
    return n_ok, n_skip, n_fail


if __name__ == "__main__":
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 4711
    sys.exit(1 if run(budget, seed)[2] else 0)
