"""Differential fuzzing of the merged (multi-GPU) CLEANING SWEEP: W emulated ranks on one GPU, uneven shards (some of them
empty), every build merged — the first one with or without the fused filter, the third one made from the second one's
live part whenever no rank re-threaded a read (amg_derive.hip; AMG_NO_DERIVE=1 for the other side) — against the
single-GPU sweep of the whole read set: final graph on every rank, corrected reads of both corrections laid end to end,
ids removed by the clip.  usage: fuzz_dist_sweep.py SECONDS [SEED]"""
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


def positions(reads, read_ids):
    gs = np.concatenate([np.arange(len(reads[r]), dtype=np.int64) * 1000 for r in read_ids]) if read_ids else np.zeros(0, np.int64)
    rl = np.asarray([len(reads[r]) * 1000 + 100 for r in read_ids], np.int64)
    return gs, gs + 899, rl


def run(budget, seed, max_cases=None):
    rng = np.random.default_rng(seed)
    t_end = time.time() + budget
    n_ok = n_skip = n_fail = n_derived = 0
    while time.time() < t_end and (max_cases is None or n_ok + n_skip + n_fail < max_cases):
        reads, k, info = fz.make_case(rng)
        world = int(rng.choice([2, 3, 5, 8]))
        fused = bool(rng.random() < 0.5)
        info.update(world=world, fused=fused)
        vocab, toks, offs, read_ids = tokenize(reads)
        gs, ge, rl = positions(reads, read_ids)
        R = len(offs) - 1
        cuts = np.sort(rng.integers(0, R + 1, world - 1))
        bounds = [0] + cuts.tolist() + [R]
        one = Engine(0); engines = []
        try:
            one.set_reads(toks, offs, vocab.two_v); one.set_positions(gs, ge, rl)
            # the single-GPU sweep; a palindromic gene-mer (even k) ends it at one of its three builds
            stopped_at, w1, w2, removed, want = None, None, None, None, None
            try:
                stage = 1; one.build(k)
                one.filter(3, 1)
                w1 = one.corrected(*one.correct_reads(), True); one.adopt_corrected()
                stage = 2; one.build(k)
                removed = np.sort(one.remove_short_linear_paths(k))
                w2 = one.corrected(*one.correct_reads(), True); one.adopt_corrected()
                stage = 3; one.build(k)
                want = td.graph_state(one)
            except Exception as e:  # noqa: BLE001
                if getattr(e, "code", None) != -4:
                    raise
                stopped_at = stage
            for r in range(world):
                lo, hi = bounds[r], bounds[r + 1]
                e = Engine(0)
                e.set_reads(toks[offs[lo]:offs[hi]], offs[lo:hi + 1] - offs[lo], vocab.two_v)
                e.set_positions(gs[offs[lo]:offs[hi]], ge[offs[lo]:offs[hi]], rl[lo:hi])
                engines.append(e)
            # the merged sweep must end at the same build with the same error
            at = None
            try:
                at = 1
                if fused:
                    dist_build_loopback(engines, k, 3, 1)
                else:
                    dist_build_loopback(engines, k)
                    for e in engines:
                        e.filter(3, 1)
                g1 = [e.corrected(*e.correct_reads(), True) for e in engines]
                for e in engines:
                    e.adopt_corrected()
                at = 2; dist_build_loopback(engines, k)
                for e in engines:
                    assert np.array_equal(np.sort(e.remove_short_linear_paths(k)), removed), "clip"
                g2 = [e.corrected(*e.correct_reads(), True) for e in engines]
                for e in engines:
                    e.adopt_corrected(); e.dist_stats(reset=True)
                at = 3; dist_build_loopback(engines, k)
                at = None
            except Exception as e:  # noqa: BLE001
                if getattr(e, "code", None) != -4:
                    raise
            assert at == stopped_at, f"single-GPU sweep stopped at build {stopped_at}, the merged one at {at}"
            if stopped_at is not None:
                n_skip += 1
                continue
            taken = {e.dist_stats()["derived_builds"] for e in engines}
            assert len(taken) == 1, taken
            n_derived += taken.pop()
            for e in engines:
                td.assert_same_graph(td.graph_state(e), want)
            for got, ref in ((g1, w1), (g2, w2)):
                for key in ("tokens", "gene_start", "gene_end", "changed"):
                    assert np.array_equal(np.concatenate([o[key] for o in got]), ref[key]), key
                assert np.array_equal(np.concatenate([np.diff(o["read_offsets"]) for o in got]), np.diff(ref["read_offsets"]))
            n_ok += 1
        except Exception:  # noqa: BLE001
            print("MISMATCH:", info, flush=True); traceback.print_exc(); n_fail += 1
        finally:
            for e in engines + [one]:
                e.close()
        if n_fail >= 5:
            break
    print(f"fuzz_dist_sweep: {n_ok} merged sweeps equal to the single-GPU one ({n_derived} third builds made from the second "
          f"graph's live part), {n_skip} sweeps ended by a palindromic gene-mer at the same build on both sides, {n_fail} failures (seed {seed})")
    return n_ok, n_skip, n_fail


if __name__ == "__main__":
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 8151
    sys.exit(1 if run(budget, seed)[2] else 0)
