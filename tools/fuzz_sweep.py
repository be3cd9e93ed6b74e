"""Differential fuzzing on the GPU: random small read sets with nasty structure (tiny vocabularies,
tandem arrays, inverted repeats, insertions / deletions / substitutions, ragged read lengths)
through the whole cleaning sweep (build, filter, correct, build, clip, correct, build), every
stage compared with the pinned Python oracle.  usage: fuzz_sweep.py SECONDS [SEED]"""
import os, sys, time, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import numpy as np
import procedures as P
from amira_amd import Engine, synth
import importlib.util
spec = importlib.util.spec_from_file_location("sw", os.path.join(ROOT, "tests", "test_gpu_sweep.py"))
sw = importlib.util.module_from_spec(spec); spec.loader.exec_module(sw)


def make_case(rng):
    V = int(rng.choice([6, 12, 30, 80, 300]))
    long_reads = rng.random() < 0.2     # reads beyond one wave (64 windows) and beyond the fast kernels' limits
    glen = int(rng.integers(220, 420)) if long_reads else int(rng.integers(30, 160))
    genome = [(1 if rng.random() < 0.5 else -1, f"g{int(rng.integers(0, V))}") for _ in range(glen)]
    for _ in range(int(rng.integers(0, 4))):           # tandem arrays
        at, n = int(rng.integers(0, len(genome))), int(rng.integers(2, 10))
        genome[at:at] = [genome[at % len(genome)]] * n
    if rng.random() < 0.4:                               # inverted repeat
        a = int(rng.integers(0, len(genome) - 8)); seg = genome[a:a + int(rng.integers(3, 8))]
        at = int(rng.integers(0, len(genome)))
        genome[at:at] = [(-st, g) for st, g in reversed(seg)]
    names = sorted({g for _, g in genome})
    n_reads = int(rng.integers(120, 320)) if long_reads else int(rng.integers(150, 700))
    err = float(rng.choice([0.0, 0.01, 0.03, 0.06]))
    indel = float(rng.choice([0.0, 0.0, 0.01]))
    reads = {}
    for r in range(n_reads):
        L = int(rng.integers(1, 40)) if rng.random() < 0.1 else int(rng.integers(12, 45))
        if long_reads and rng.random() < 0.6:
            L = int(rng.integers(60, 200))
        L = min(L, len(genome))
        s0 = int(rng.integers(0, len(genome) - L + 1))
        seq = list(genome[s0:s0 + L])
        if rng.random() < 0.5:
            seq = [(-st, g) for st, g in reversed(seq)]
        out = []
        for st, g in seq:
            u = rng.random()
            if u < err:
                g = names[int(rng.integers(0, len(names)))]
            elif u < err + indel:
                continue
            elif u < err + 2 * indel:
                out.append(("+" if st > 0 else "-") + names[int(rng.integers(0, len(names)))])
            out.append(("+" if st > 0 else "-") + g)
        reads[f"r{r:05d}"] = out
    k = int(rng.choice([2, 3, 3, 4, 5, 5, 6, 7, 9]))
    return reads, k, dict(V=V, glen=len(genome), n_reads=n_reads, err=err, indel=indel, k=k)


def run(budget, seed, max_cases=None):
    """returns (sweeps equal to the oracle, palindrome assertions on both sides, failures)"""
    rng = np.random.default_rng(seed)
    eng = Engine(0)
    t_end = time.time() + budget
    n_ok = n_pal = n_fail = 0
    while time.time() < t_end and (max_cases is None or n_ok + n_pal + n_fail < max_cases):
        reads, k, info = make_case(rng)
        pos = synth.positions_for(reads)
        fq = P.FakeFastq(synth.fake_fastq_lengths(reads))
        try:
            mc = int(rng.choice([2, 3]))
            sw.run_sweep(eng, reads, pos, fq, k, min_cov=mc)
            n_ok += 1
        except AssertionError as e:
            msg = str(e)
            if "identical" in msg:      # palindromic gene-mer: the oracle asserts like the reference
                try:
                    from amira_amd import tokenize, _ffi
                    vocab, toks, offs, _ = tokenize(reads)
                    eng.set_reads(toks, offs, vocab.two_v)
                    eng.build(k)
                    print("MISMATCH: oracle asserted a palindrome, engine built", info, flush=True); n_fail += 1
                except Exception as e2:  # noqa: BLE001
                    if getattr(e2, "code", None) == -4:
                        n_pal += 1
                    else:
                        print("MISMATCH (palindrome case):", repr(e2), info, flush=True); n_fail += 1
            else:
                print("MISMATCH:", info, flush=True); traceback.print_exc(); n_fail += 1
                dump = os.environ.get("FUZZ_DUMP")
                if dump:  # keep the failing input for a stand-alone reproduction
                    import json
                    json.dump({"reads": reads, "k": k, "min_cov": mc, "info": info}, open(dump, "w"))
                    break
        except Exception as e:  # noqa: BLE001
            if getattr(e, "code", None) == -4:   # the engine met a palindromic gene-mer: so must the oracle,
                from amira_oracle import GeneMerGraph  # in the first build or in a rebuild on corrected reads
                try:
                    g1 = GeneMerGraph(reads, k, {r: list(v) for r, v in pos.items()})
                    g1.filter_graph(mc, 1)
                    r2, p2 = g1.correct_reads(fq)
                    g2 = GeneMerGraph(r2, k, p2)
                    g2.remove_short_linear_paths(k)
                    r3, p3 = g2.correct_reads(fq)
                    GeneMerGraph(r3, k, p3)
                    print("MISMATCH: engine asserted a palindrome, oracle finished the sweep", info, flush=True); n_fail += 1
                except AssertionError as e2:
                    if "identical" in str(e2):
                        n_pal += 1
                    else:
                        print("MISMATCH (palindrome case):", repr(e2), info, flush=True); n_fail += 1
            else:
                print("ERROR:", info, flush=True); traceback.print_exc(); n_fail += 1
        if n_fail >= 5:
            break
    print(f"fuzz: {n_ok} sweeps equal to the oracle, {n_pal} palindrome assertions on both sides, {n_fail} failures (seed {seed})")
    eng.close()
    return n_ok, n_pal, n_fail


if __name__ == "__main__":
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 12345
    sys.exit(1 if run(budget, seed)[2] else 0)
