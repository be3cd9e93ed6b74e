"""Differential fuzzing of the reference-compatible Python layer (amira_amd.GeneMerGraph & co.
over the HIP engine) against the oracle: the golden-generating procedures (tests/golden/
procedures.py: whole sweep through the class API, planted-allele clustering) on random
parameters, product vs oracle.  Run under PYTHONHASHSEED=0 (clustering of the reference depends
on set order).  usage: PYTHONHASHSEED=0 fuzz_api.py SECONDS [SEED]"""
import json, os, sys, time, traceback, types
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import numpy as np
import procedures as P


def impls():
    import amira_amd
    from amira_amd.graph_utils import choose_kmer_size, get_overall_mean_node_coverages
    prod = types.SimpleNamespace(GeneMerGraph=amira_amd.GeneMerGraph, Gene=amira_amd.Gene, GeneMer=amira_amd.GeneMer,
                                 choose_kmer_size=choose_kmer_size,
                                 get_overall_mean_node_coverages=get_overall_mean_node_coverages)
    import amira_oracle
    from amira_oracle import driver
    orc = types.SimpleNamespace(GeneMerGraph=amira_oracle.GeneMerGraph, Gene=amira_oracle.Gene,
                                GeneMer=amira_oracle.GeneMer, choose_kmer_size=driver.choose_kmer_size,
                                get_overall_mean_node_coverages=driver.get_overall_mean_node_coverages)
    return prod, orc


def run(budget, seed, max_cases=None):
    assert os.environ.get("PYTHONHASHSEED") == "0", "run with PYTHONHASHSEED=0"
    prod, orc = impls()
    rng = np.random.default_rng(seed)
    t_end = time.time() + budget
    n_ok = n_fail = 0
    while time.time() < t_end and (max_cases is None or n_ok + n_fail < max_cases):
        k = int(rng.choice([3, 5, 7]))
        u = rng.random()
        only = os.environ.get("FUZZ_ONLY")   # e.g. FUZZ_ONLY=bubbles
        if only == "bubbles":
            u = 0.0
        elif only == "planted":   # read-path clustering (assign_reads_to_genes) on planted multi-copy genes
            u = 0.99
        if u < 0.15:   # bubble popping, device MinHash against the oracle's pure-Python sketch
            k = int(rng.choice([3, 5]))
            args = (int(rng.integers(1, 1 << 30)), int(rng.integers(80, 220)), int(rng.integers(k + 8, 36)),
                    int(rng.choice([40, 120, 400])), k, float(rng.choice([0.02, 0.05, 0.08])))
            proc = P.p_bubbles_random
        elif u < 0.75:
            args = (int(rng.integers(1, 1 << 30)), int(rng.integers(100, 500)), int(rng.integers(k + 6, 45)),
                    int(rng.choice([40, 120, 400, 1500])), k, float(rng.choice([0.0, 0.02, 0.05])))
            proc = P.p_sweep
        else:
            args = (int(rng.integers(1, 1 << 30)), int(rng.integers(150, 400)), int(rng.integers(25, 45)),
                    int(rng.choice([300, 1000])), k)
            proc = P.p_planted
        try:
            a = json.loads(json.dumps(proc(prod, *args)))
            b = json.loads(json.dumps(proc(orc, *args)))
            assert a == b
            n_ok += 1
        except Exception:  # noqa: BLE001
            print("MISMATCH:", proc.__name__, args, flush=True); traceback.print_exc(); n_fail += 1
            if n_fail >= 3:
                break
    print(f"fuzz_api: {n_ok} procedures equal (product vs oracle), {n_fail} failures (seed {seed})")
    return n_ok, n_fail


if __name__ == "__main__":
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 271828
    sys.exit(1 if run(budget, seed)[1] else 0)
