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
    n_ok, n_pal, n_fail = fz.run(budget=120.0, seed=seed, max_cases=20)
    assert n_fail == 0 and n_ok + n_pal == 20
