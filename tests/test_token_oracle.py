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


# ---- oracle/token_sweep.c (the whole cleaning sweep, stateful) against the Python oracle, driven
# ---- through the very procedure the GPU engine is checked with (tests/test_gpu_sweep.py)
@pytest.fixture()
def ceng():
    e = token_oracle.SweepEngine()
    yield e
    e.close()


@pytest.mark.parametrize("seed,N,L,V,k,err", [(7, 400, 30, 300, 5, 0.03), (11, 400, 24, 200, 3, 0.03),
                                              (13, 300, 40, 250, 7, 0.02), (17, 800, 40, 150, 5, 0.05),
                                              (31, 600, 60, 400, 5, 0.04)])
def test_c_sweep_equals_python_oracle_synthetic(ceng, seed, N, L, V, k, err):
    from test_gpu_sweep import run_sweep
    reads, pos, fq = P.synth_inputs(seed, N, L, V, err)
    run_sweep(ceng, reads, pos, fq, k)


@pytest.mark.parametrize("name,k", [("nine", 3), ("nine", 5), ("four", 5), ("six", 5)])
def test_c_sweep_equals_python_oracle_fixture(ceng, name, k):
    from test_gpu_sweep import run_sweep
    calls, pos = P.fixture(name)
    lengths = {r: (pos[r][-1][1] + 200 if pos[r] else 100) for r in pos}
    run_sweep(ceng, calls, pos, P.FakeFastq(lengths), k)


@pytest.mark.parametrize("seed,k", [(3, 3), (4, 5)])
def test_c_sweep_equals_python_oracle_tandem(ceng, seed, k):
    from test_gpu_sweep import _tandem_reads, run_sweep
    reads, pos, fq = _tandem_reads(seed, 700, 26, 0.04)
    run_sweep(ceng, reads, pos, fq, k)


def test_c_sweep_nw_tie_fixture(ceng):
    import json, lzma, os
    from test_gpu_sweep import run_sweep
    path = os.path.join(os.path.dirname(__file__), "golden", "data", "nw_tie_case.json.xz")
    d = json.loads(lzma.open(path, "rt").read())
    spec_dir = os.path.dirname(os.path.dirname(__file__))
    import importlib.util
    spec = importlib.util.spec_from_file_location("_amg_synth", os.path.join(spec_dir, "amira_amd", "synth.py"))
    synth = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(synth)
    reads = d["reads"]
    run_sweep(ceng, reads, synth.positions_for(reads), P.FakeFastq(synth.fake_fastq_lengths(reads)), d["k"],
              min_cov=d["min_cov"])


def test_c_sweep_low_coverage_components(ceng):
    from amira_oracle import GeneMerGraph
    from helpers import compare_engine_to_oracle
    calls, _ = P.fixture("nine")
    vocab, toks, offs, read_ids = _tokenize(calls)
    ceng.set_reads(toks, offs, vocab.two_v)
    ceng.build(3)
    g = GeneMerGraph(calls, 3)
    for m in (5, 40):
        ceng.remove_low_coverage_components(m)
        g.remove_low_coverage_components(m)
        compare_engine_to_oracle(ceng, oracle_arrays(g, vocab, read_ids, offs, 3), live_only=True)


def test_threaded_correct_reads_equals_sequential():
    """the C sweep oracle corrects every read on its own: the chunks of tsw_set_threads(n) (used by the full-size GPU
    tests, where the sequential 8 M-read run was minutes) laid end to end are the sequential output"""
    reads, pos, fq = P.synth_inputs(11, 3000, 40, 400, 0.03)
    vocab, toks, offs, read_ids = _tokenize(reads)
    gs = np.fromiter((p[0] for r in read_ids for p in pos[r]), dtype=np.int64)
    ge = np.fromiter((p[1] for r in read_ids for p in pos[r]), dtype=np.int64)
    rl = np.asarray([len(fq[r]["sequence"]) for r in read_ids], np.int64)
    outs = []
    for threads in (1, 5):
        o = token_oracle.Sweep(toks, offs, vocab.two_v, gs, ge, rl)
        o.build(5)
        o.filter(3, 1)
        outs.append(o.corrected(*o.correct_reads(threads=threads), True))
        o.close()
    assert outs[0]["changed"].any()
    for key in outs[0]:
        assert np.array_equal(outs[0][key], outs[1][key]), key
