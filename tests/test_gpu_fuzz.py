"""A short run of the differential fuzzer (tools/fuzz_sweep.py): random small read sets with
tiny vocabularies, tandem arrays, inverted repeats, indels and ragged lengths through the whole
cleaning sweep, every stage compared with the oracle; even k exercises the palindrome assertion
on both sides.  (Longer runs: `python tools/fuzz_sweep.py SECONDS SEED` on the GPU box.)"""
import importlib.util
import os

import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", [20250905, 31337])
def test_random_sweeps_equal_oracle(seed):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("fuzz_sweep", os.path.join(root, "tools", "fuzz_sweep.py"))
    fz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fz)
    n_ok, n_pal, n_fail = fz.run(budget=120.0, seed=seed, max_cases=14)
    assert n_fail == 0 and n_ok + n_pal == 14


def test_random_merged_builds_equal_unsharded():
    """tools/fuzz_dist.py: 2 - 8 emulated ranks with uneven (also empty) shards, plain merge and
    merge with the fused coverage filter, against the unsharded engine"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("fuzz_dist", os.path.join(root, "tools", "fuzz_dist.py"))
    fd = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fd)
    n_ok, n_skip, n_fail = fd.run(budget=120.0, seed=4711, max_cases=150)
    assert n_fail == 0 and n_ok > 100


def test_random_merged_sweeps_equal_single_gpu_sweep():
    """tools/fuzz_dist_sweep.py: the whole cleaning sweep with every build merged across 2 - 8 emulated ranks (the third
    one made from the second graph's live part whenever no rank re-threaded a read) against the single-GPU sweep"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("fuzz_dist_sweep", os.path.join(root, "tools", "fuzz_dist_sweep.py"))
    fd = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fd)
    n_ok, n_skip, n_fail = fd.run(budget=60.0, seed=8151, max_cases=120)
    assert n_fail == 0 and n_ok > 60


def test_random_bubble_popping_equals_oracle():
    """tools/fuzz_api.py restricted to bubble popping (f1): correct_low_coverage_paths on random reads, the device
    MinHash against the oracle's pure-Python sketch.  In a child interpreter with PYTHONHASHSEED=0 (the reference's
    set-order dependence)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PYTHONHASHSEED="0", FUZZ_ONLY="bubbles")
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_api.py"), "20", "424242"], env=env,
                         capture_output=True, text=True, timeout=600)
    last = [l for l in out.stdout.splitlines() if l.startswith("fuzz_api:")]
    assert out.returncode == 0 and last, out.stdout[-2000:] + out.stderr[-2000:]
    n_ok = int(last[-1].split()[1])
    assert n_ok >= 2 and " 0 failures" in last[-1]


def test_random_two_word_keys_equal_c_oracle():
    """tools/fuzz_twoword.py: read sets whose gene-mers need two-word exact keys, many of them agreeing in their first
    63 bits, with and without minimiser buckets, against the sequential C oracle (the slot protocol of amg_x.h: owner by
    the first key word, publication of the second, the lone continuation of a half-equal key)"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_twoword.py"), "20", "20261003"],
                         capture_output=True, text=True, timeout=600)
    last = [l for l in out.stdout.splitlines() if l.startswith("fuzz_twoword:")]
    assert out.returncode == 0 and last, out.stdout[-2000:] + out.stderr[-2000:]
    assert int(last[-1].split()[1]) >= 3 and " 0 failures" in last[-1]
