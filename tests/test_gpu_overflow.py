"""Table sizing: inputs whose distinct gene-mers exceed the initial table (half a slot per
token) must be rebuilt transparently and still match the oracle."""
import numpy as np
import pytest

from helpers import compare_engine_to_oracle, oracle_arrays

pytestmark = pytest.mark.gpu


def test_all_distinct_windows_trigger_table_growth():
    from amira_amd import Engine, tokenize
    from amira_oracle import GeneMerGraph
    rng = np.random.default_rng(5)
    # 3 000 reads of 12 random genes over a 50 000-gene vocabulary: every window is new
    reads = {f"r{i:05d}": [("+" if s else "-") + f"g{g}" for g, s in zip(rng.integers(0, 50000, 12), rng.integers(0, 2, 12))]
             for i in range(3000)}
    vocab, toks, offs, read_ids = tokenize(reads)
    eng = Engine(0)
    eng.set_reads(toks, offs, vocab.two_v)
    eng.build(3)
    c = eng.counts()
    assert c["n_nodes"] > c["n_tokens"] // 2 and c["build_retries"] >= 1
    compare_engine_to_oracle(eng, oracle_arrays(GeneMerGraph(reads, 3), vocab, read_ids, offs, 3))
    eng.close()


def test_fingerprint_collisions_are_detected_and_rebuilt(monkeypatch):
    """The node table is keyed by a 64-bit fingerprint that the edge pass verifies exactly.
    With the fingerprint crippled to 12 bits on the first attempt (test hook) distinct gene-mers
    are certain to share a slot: the verification must flag it and the rebuild (new seed, full
    fingerprint) must equal the oracle."""
    import procedures as P
    from amira_amd import Engine, tokenize
    from amira_oracle import GeneMerGraph
    monkeypatch.setenv("AMG_TEST_WEAK_FP", "1")
    reads, _, _ = P.synth_inputs(7, 400, 30, 300, 0.03)   # ~10 k distinct gene-mers >> 4096 fingerprints
    vocab, toks, offs, read_ids = tokenize(reads)
    eng = Engine(0)
    eng.set_reads(toks, offs, vocab.two_v)
    eng.build(5)
    assert eng.counts()["build_retries"] >= 1
    compare_engine_to_oracle(eng, oracle_arrays(GeneMerGraph(reads, 5), vocab, read_ids, offs, 5))
    eng.close()
