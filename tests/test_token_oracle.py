"""The C token-space oracle (oracle/token_oracle.c) against the Python oracle, which is
itself pinned to the reference by goldens.  CPU only."""
import numpy as np
import pytest

import procedures as P
import token_oracle
from helpers import oracle_arrays


def _tokenize(reads):
    # tests may not import the product without libamg.so; tokens.py is pure Python
    import importlib.util, os
    spec = importlib.util.spec_from_file_location(
        "_amg_tokens", os.path.join(os.path.dirname(os.path.dirname(__file__)), "amira_amd", "tokens.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m.tokenize(reads)


@pytest.mark.parametrize("case", [("synth", 7, 300, 30, 200, 5), ("synth", 11, 300, 24, 150, 3),
                                  ("synth", 13, 200, 40, 150, 7), ("fixture", "five", 3),
                                  ("fixture", "eight", 5), ("tiny",)])
def test_token_oracle_equals_python_oracle(case):
    from amira_oracle import GeneMerGraph
    if case[0] == "synth":
        _, seed, N, L, V, k = case
        reads, _, _ = P.synth_inputs(seed, N, L, V, 0.03)
    elif case[0] == "fixture":
        reads, _ = P.fixture(case[1])
        k = case[2]
    else:
        reads, k = {"a": ["-g4"] * 5, "b": ["+a", "+b", "-b", "-a"], "c": ["+x"], "d": []}, 3
    vocab, toks, offs, read_ids = _tokenize(reads)
    got = token_oracle.build(toks, offs, k, vocab.two_v)
    want = oracle_arrays(GeneMerGraph(reads, k), vocab, read_ids, offs, k)
    for key in ("tokens", "coverage", "first_dir", "src", "tgt", "sdir", "tdir", "ecov", "tok_node", "tok_dir"):
        assert np.array_equal(got[key], want[key]), key
    assert got["n_short"] == len(want["short"])
