"""Table sizing: inputs whose distinct gene-mers exceed the initial table (half a slot per
token) must be rebuilt transparently and still match the oracle."""
import numpy as np
import pytest

from helpers import compare_engine_to_oracle, oracle_arrays

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n_genes,buckets", [(50000, "0"), (50000, "1"), (60, "1")])
def test_all_distinct_windows_trigger_table_growth(n_genes, buckets, monkeypatch):
    """3 000 reads of 12 random genes: (nearly) every window is new.  With hashed slots only the table (a quarter of a
    slot per token) overflows and is rebuilt larger; with minimiser buckets a 50 000-gene vocabulary brings 8 slots per
    gene of its own and needs no growth, a 60-gene vocabulary (k = 5 so that the windows stay distinct) does"""
    from amira_amd import Engine, tokenize
    from amira_oracle import GeneMerGraph
    monkeypatch.setenv("AMG_NODE_BUCKETS", buckets)
    rng = np.random.default_rng(5)
    k = 3 if n_genes > 1000 else 5
    reads = {f"r{i:05d}": [("+" if s else "-") + f"g{g}" for g, s in zip(rng.integers(0, n_genes, 12), rng.integers(0, 2, 12))]
             for i in range(3000)}
    vocab, toks, offs, read_ids = tokenize(reads)
    eng = Engine(0)
    eng.set_reads(toks, offs, vocab.two_v)
    eng.build(k)
    c = eng.counts()
    assert c["n_nodes"] > c["n_windows"] // 2
    if not (n_genes > 1000 and buckets == "1"):
        assert c["build_retries"] >= 1
    compare_engine_to_oracle(eng, oracle_arrays(GeneMerGraph(reads, k), vocab, read_ids, offs, k))
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


@pytest.mark.parametrize("filtered", [False, True])
def test_fingerprint_collisions_in_the_16_byte_slots_are_detected_and_rebuilt(monkeypatch, filtered):
    """k = 11 on a 300-gene vocabulary (110 bits): the tuple does not fit a slot and its 94-bit fingerprint is the key;
    every window is compared with its claim's first occurrence in the token stream (k_x_verify_fp).  With the fingerprint
    cut to 12 bits on the first attempt distinct gene-mers share keys: the check must flag it and the rebuild (next
    seed, full fingerprint) must equal the oracle — in the plain and in the filtered build"""
    import procedures as P
    from amira_amd import Engine, tokenize
    from amira_oracle import GeneMerGraph
    monkeypatch.setenv("AMG_TEST_WEAK_FP", "1")
    reads, _, _ = P.synth_inputs(7, 400, 30, 300, 0.03)
    vocab, toks, offs, read_ids = tokenize(reads)
    eng = Engine(0)
    eng.set_reads(toks, offs, vocab.two_v)
    g = GeneMerGraph(reads, 11)
    if filtered:
        eng.build_filtered(11, 2, 1)
        g.filter_graph(2, 1)
    else:
        eng.build(11)
    c = eng.counts()
    assert c["build_retries"] >= 1 and c["exact_keys"] == 2
    if filtered:
        # (a filtered build numbers the survivors only: compared through the plain build + filter of the same engine)
        other = Engine(0)
        other.set_reads(toks, offs, vocab.two_v)
        monkeypatch.delenv("AMG_TEST_WEAK_FP")
        other.build(11)
        other.filter(2, 1)
        compare_engine_to_oracle(other, oracle_arrays(g, vocab, read_ids, offs, 11), live_only=True)
        a, b = eng.nodes(), other.nodes()
        live = b["alive"] != 0
        assert np.array_equal(a["tokens"], b["tokens"][live]) and np.array_equal(a["coverage"], b["coverage"][live])
        other.close()
    else:
        compare_engine_to_oracle(eng, oracle_arrays(g, vocab, read_ids, offs, 11))
    eng.close()


@pytest.mark.parametrize("key_mode", ["exact", "fp"])
@pytest.mark.parametrize("damage", ["first_offset", "decreasing", "last_offset", "token_high", "token_negative"])
def test_malformed_device_csr_is_refused(monkeypatch, key_mode, damage):
    """device-resident (copied or borrowed) inputs cannot be validated on the host: the first
    kernel of the build checks the offsets and the token range and the build fails with
    AMG_E_ARG instead of indexing out of bounds or aliasing tuples"""
    import torch
    import procedures as P
    from amira_amd import Engine, tokenize, _ffi
    if key_mode == "fp":
        monkeypatch.setenv("AMG_KEY_MODE", "fp")
    reads, _, _ = P.synth_inputs(7, 200, 20, 100, 0.02)
    vocab, toks, offs, _ = tokenize(reads)
    toks, offs = toks.copy(), offs.copy()
    n_reads = len(offs) - 1
    if damage == "first_offset":
        offs[0] = 3
    elif damage == "decreasing":
        offs[50] = offs[49] - 2
    elif damage == "last_offset":
        offs = np.concatenate([offs, [offs[-1] - 5]])  # one more read whose end lies before its start
        n_reads += 1
    elif damage == "token_high":
        toks[1234] = vocab.two_v + 7
    else:
        toks[77] = -3
    d_t, d_o = torch.from_numpy(toks).cuda(), torch.from_numpy(offs).cuda()
    torch.cuda.synchronize()
    eng = Engine(0)
    try:
        for borrow in (False, True):
            eng.set_reads_device(d_t.data_ptr(), d_o.data_ptr(), n_reads, vocab.two_v, borrow=borrow)
            with pytest.raises(_ffi.AmgError) as ei:
                eng.build(5)
            assert ei.value.code == -2
        # the ctx is still usable afterwards
        vocab2, t2, o2, _ = tokenize(reads)
        eng.set_reads(t2, o2, vocab2.two_v)
        eng.build(5)
        assert eng.counts()["n_nodes"] > 0
    finally:
        eng.close()


def test_hashed_edge_region_overflow_is_rebuilt():
    """The edge table keeps the class joining node ids n and n + 1 in slot n and sizes its HASHED region for the rest
    to a quarter of the nodes.  A dense graph over a tiny vocabulary (a few thousand nodes, tens of thousands of edge
    classes, hardly any of them between consecutive ids) overflows that region: the build must grow it and still
    equal the sequential C oracle."""
    import token_oracle
    from amira_amd import Engine
    from helpers import compare_engine_to_sweep
    rng = np.random.default_rng(11)
    V, L, N, k = 10, 30, 20000, 3
    genes = rng.integers(0, V, (N, L))
    strands = rng.integers(0, 2, (N, L))
    toks = np.where(strands == 1, V + genes, V - 1 - genes).astype(np.int32).ravel()
    offs = np.arange(0, (N + 1) * L, L, dtype=np.int64)
    eng, orc = Engine(0), token_oracle.Sweep(toks, offs, 2 * V)
    try:
        eng.set_reads(toks, offs, 2 * V)
        eng.build(k)   # (an odd k has no palindromes)
        orc.build(k)
        c = eng.counts()
        assert c["n_edges"] > 8 * c["n_nodes"] and c["build_retries"] >= 1
        compare_engine_to_sweep(eng, orc, "dense graph")
    finally:
        eng.close()
        orc.close()
