"""GPU parity of the HIP build (K1-K4, K7) against the CPU oracle, through the C ABI."""
import numpy as np
import pytest

import dump as D
import procedures as P
from helpers import compare_engine_to_oracle, oracle_arrays

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from amira_amd import Engine
    e = Engine(0)
    yield e
    e.close()


def _run(eng, reads, k, filt=None):
    from amira_amd import tokenize
    from amira_oracle import GeneMerGraph
    vocab, toks, offs, read_ids = tokenize(reads)
    eng.set_reads(toks, offs, vocab.two_v)
    eng.build(k)
    g = GeneMerGraph(reads, k)
    want = oracle_arrays(g, vocab, read_ids, offs, k)
    c = eng.counts()
    assert c["n_short_reads"] == len(want["short"])
    assert c["n_windows"] == int((want["tok_node"] != -1).sum())
    assert c["n_components"] == len(set(want["component"].tolist()))
    compare_engine_to_oracle(eng, want)
    if filt:
        eng.filter(*filt)
        g.filter_graph(*filt)
        want = oracle_arrays(g, vocab, read_ids, offs, k)
        compare_engine_to_oracle(eng, want, live_only=True)
        flags = eng.reads_to_correct()
        assert sorted(r for r, f in zip(read_ids, flags) if f) == want["to_correct"]
    return c


def test_tiny_known_answers(eng):
    # tests/test_gene_mer_graph.py:38-68 shape: two reads sharing a gene-mer
    c = _run(eng, {"read1": ["+gene1", "-gene2", "+gene3", "-gene4"],
                   "read2": ["+gene1", "-gene2", "+gene3"]}, 3)
    assert (c["n_nodes"], c["n_edges"]) == (2, 2)
    # tandem self-loop and hairpin
    c = _run(eng, {"r": ["-gene4"] * 5}, 3)
    assert (c["n_nodes"], c["n_edges"]) == (1, 1)
    _run(eng, {"r": ["+a", "+b", "-b", "-a"], "s": ["+a", "+b", "-b", "-a", "+c"]}, 3)
    # k = 1: every gene is a node, canonical strand is '-'
    _run(eng, {"r1": ["+a", "-b", "+c", "+a"], "r2": ["-c", "+b"], "r3": ["+d"]}, 1)


def test_ragged_and_empty_reads(eng):
    reads = {"e0": [], "s1": ["+a"], "s2": ["+a", "-b"], "x": ["+a", "-b", "+c"], "e1": [],
             "y": ["-c", "+b", "-a", "+d", "+e", "-f", "+a", "-b", "+c"], "e2": []}
    c = _run(eng, reads, 3)
    assert c["n_short_reads"] == 5
    _run(eng, reads, 5)
    _run(eng, {"only_short": ["+a", "+b"]}, 3)
    _run(eng, {}, 3)


def test_palindrome_asserts_like_reference(eng):
    from amira_amd import _ffi, tokenize
    reads = {"p": ["+a", "-a", "+b"]}  # k = 2 window (+a, -a) equals its reverse complement
    vocab, toks, offs, _ = tokenize(reads)
    eng.set_reads(toks, offs, vocab.two_v)
    with pytest.raises(_ffi.AmgError) as ei:
        eng.build(2)
    assert ei.value.code == _ffi.E_PALINDROME


@pytest.mark.parametrize("name,k", [("five", 3), ("five", 5), ("seven", 3), ("eight", 5),
                                     ("four", 3), ("nine", 5), ("three", 3)])
def test_fixture_build_and_filter(eng, name, k):
    calls, _ = P.fixture(name)
    _run(eng, calls, k, filt=(3, 1))


@pytest.mark.parametrize("seed,N,L,V,k,err", [(7, 400, 30, 300, 5, 0.03), (11, 400, 24, 200, 3, 0.03),
                                              (13, 300, 40, 250, 7, 0.02), (17, 800, 40, 150, 5, 0.05),
                                              (19, 500, 33, 400, 9, 0.02), (23, 300, 50, 300, 15, 0.01)])
def test_synthetic_build_and_filter(eng, seed, N, L, V, k, err):
    reads, _, _ = P.synth_inputs(seed, N, L, V, err)
    _run(eng, reads, k, filt=(3, 1))
    _run(eng, reads, k, filt=(2, 2))


@pytest.mark.parametrize("k", [3, 5])
def test_fingerprint_key_path(eng, monkeypatch, k):
    """small k normally takes the exact-key build (the packed tuple is the key); the
    fingerprint + verification build that large k / large vocabularies and the multi-GPU merge
    use must give the same graph on the same reads"""
    reads, _, _ = P.synth_inputs(7, 400, 30, 300, 0.03)
    assert _run(eng, reads, k, filt=(3, 1))["exact_keys"] == 1
    monkeypatch.setenv("AMG_KEY_MODE", "fp")
    assert _run(eng, reads, k, filt=(3, 1))["exact_keys"] == 0


def test_key_mode_follows_tuple_width(eng):
    reads, _, _ = P.synth_inputs(19, 300, 33, 400, 0.02)   # 2V = 800 -> 10 bits per token
    assert _run(eng, reads, 6)["exact_keys"] == 1           # 60 bits: one word
    assert _run(eng, reads, 7)["exact_keys"] == 1           # 70 bits: spills into the tag
    assert _run(eng, reads, 9)["exact_keys"] == 1           # 90 bits
    assert _run(eng, reads, 10)["exact_keys"] == 2          # 100 bits: the same slots keyed by a verified 94-bit fingerprint
    assert _run(eng, reads, 13)["exact_keys"] == 2          # 130 bits


def test_rebuild_on_same_context(eng):
    reads, _, _ = P.synth_inputs(3, 200, 30, 100, 0.02)
    a = _run(eng, reads, 5)
    b = _run(eng, reads, 3)
    c = _run(eng, reads, 5)
    assert a["n_nodes"] == c["n_nodes"] and b["n_nodes"] != 0


def test_match_patterns_against_brute_force(eng):
    """K6 (amg_match_patterns) vs find_sublist_indices over every read, tokens and node ids."""
    from amira_amd import tokenize
    import random
    reads, _, _ = P.synth_inputs(31, 300, 30, 60, 0.05)
    vocab, toks, offs, read_ids = tokenize(reads)
    eng.set_reads(toks, offs, vocab.two_v)
    eng.build(3)
    tok_node, _ = eng.read_nodes()
    rng = random.Random(3)
    for which, seq, tail in ((0, toks, 0), (1, tok_node, 2)):
        rows = [seq[offs[r]:offs[r + 1] - tail].tolist() for r in range(len(read_ids))]
        pats = []
        for _ in range(200):
            row = rows[rng.randrange(len(rows))]
            if len(row) < 2:
                continue
            m = rng.randint(1, min(6, len(row)))
            s = rng.randrange(len(row) - m + 1)
            pats.append(row[s:s + m])
        pats += [[10 ** 6], [rows[0][0], 10 ** 6], []]  # cannot match / empty
        off, hr, hp = eng.match_patterns(which, pats)
        for j, p in enumerate(pats):
            want = [(r, i) for r, row in enumerate(rows) for i in range(len(row) - len(p) + 1)
                    if p and row[i:i + len(p)] == p]
            got = list(zip(hr[off[j]:off[j + 1]].tolist(), hp[off[j]:off[j + 1]].tolist()))
            assert got == want, (which, j)


@pytest.mark.parametrize("case", [("fixture", "nine"), ("fixture", "five"), ("synth", 13, 300, 40, 250, 0.02),
                                  ("ragged",)])
def test_build_multi_equals_single_builds(case):
    """amg_build_multi (choose_kmer_size's seven graphs, graph_utils.py:258-296, from two passes over the tokens):
    graph i must be exactly what amg_build(k_i) gives — nodes, edges, coverages, adjacency, components, window ids"""
    from amira_amd import Engine, tokenize
    from test_gpu_dist import assert_same_graph, graph_state
    if case[0] == "fixture":
        reads, _ = P.fixture(case[1])
    elif case[0] == "synth":
        reads, _, _ = P.synth_inputs(*case[1:])
    else:
        reads = {"e0": [], "s1": ["+a"], "x": ["+a", "-b", "+c"], "y": ["-c", "+b", "-a", "+d", "+e", "-f", "+a", "-b", "+c"],
                 "z": ["+a", "-b", "+c", "+d", "+e", "-f", "+g", "+h", "+i", "-j", "+k", "+l", "-m", "+n", "+o", "+p", "-q"]}
    vocab, toks, offs, _ = tokenize(reads)
    ks = list(range(3, 16, 2))
    engines = [Engine(0) for _ in ks]
    single = Engine(0)
    try:
        engines[0].set_reads(toks, offs, vocab.two_v)
        Engine.build_multi(engines, ks)
        single.set_reads(toks, offs, vocab.two_v)
        for e, k in zip(engines, ks):
            single.build(k)
            assert_same_graph(graph_state(e), graph_state(single))
            assert np.array_equal(e.read_nodes()[0], single.read_nodes()[0])
            assert np.array_equal(e.read_nodes()[1], single.read_nodes()[1])
            ca, cb = e.counts(), single.counts()
            for key in ("n_nodes", "n_edges", "n_windows", "n_short_reads", "n_components"):
                assert ca[key] == cb[key], (k, key)
    finally:
        for e in engines + [single]:
            e.close()


def test_build_many_graphs_equal_separate_graphs():
    """GeneMerGraph.build_many == [GeneMerGraph(reads, k, positions) for k in ...] through the object API, including a
    correction on one of the later graphs (its engine gets the positions only then)"""
    from amira_amd import GeneMerGraph
    reads, pos, fq = P.synth_inputs(17, 400, 30, 100, 0.04)
    many = GeneMerGraph.build_many(dict(reads), [3, 5, 7], {r: list(v) for r, v in pos.items()})
    for g, k in zip(many, (3, 5, 7)):
        one = GeneMerGraph(dict(reads), k, {r: list(v) for r, v in pos.items()})
        assert list(g.get_nodes()) == list(one.get_nodes()) and list(g.get_edges()) == list(one.get_edges())
        assert [n.get_node_coverage() for n in g.all_nodes()] == [n.get_node_coverage() for n in one.all_nodes()]
        assert g.components() == one.components()
        assert {r: list(v) for r, v in g.get_readNodePositions().items()} == \
               {r: list(v) for r, v in one.get_readNodePositions().items()}
        assert g.get_short_read_annotations() == one.get_short_read_annotations()
        if k == 5:
            g.filter_graph(3, 1)
            one.filter_graph(3, 1)
            assert g.correct_reads(fq) == one.correct_reads(fq)
        one.close()
    many[0].close()          # the first graph's engine is only handed back when the graphs that read its arrays are closed
    assert many[0]._engine is not None
    many[1].close()
    many[2].close()
    assert many[0]._engine is None


@pytest.mark.parametrize("buckets,shards", [("1", "0"), ("0", "0"), ("1", "1")])
def test_half_equal_two_word_keys(buckets, shards, monkeypatch):
    """Two-word exact keys (k * bits > 63) whose first 63 bits agree and whose last gene differs: the slot belongs to
    whoever takes the first key word, everybody else learns from the second word — published later, in one store —
    whether the slot holds ITS key, and goes on probing if not.  Tens of thousands of such keys are created at once
    here (short reads [A, B, C, D, X_i]: with minimiser buckets they all aim at ONE home slot), each seen two or three
    times, forwards and as reverse complements, against the sequential C oracle."""
    import token_oracle
    from amira_amd import Engine
    monkeypatch.setenv("AMG_NODE_BUCKETS", buckets)
    monkeypatch.setenv("AMG_CLAIM_SHARDS", shards)  # (1: claim ids from the 64 shard counters, as large inputs take them)
    V, k = 30000, 5                      # 16 bits per gene: 80-bit tuples
    rng = np.random.default_rng(11)
    reads = []
    for fam in range(6):                 # six families of keys sharing their first four genes
        head = rng.choice(V, 4, replace=False)
        strands = rng.integers(0, 2, 5)
        for x in rng.choice(V, 9000, replace=False):
            genes = np.concatenate([head, [x]])
            fwd = np.where(strands == 1, V + genes, V - 1 - genes)
            for _ in range(int(rng.integers(2, 4))):
                reads.append(fwd if rng.random() < 0.5 else (2 * V - 1 - fwd[::-1]))
    # longer reads over the same families (edges between half-equal keys' nodes and ordinary ones)
    for _ in range(3000):
        reads.append(rng.integers(0, 2 * V, int(rng.integers(5, 12))))
    order = rng.permutation(len(reads))
    reads = [reads[i] for i in order]
    toks = np.concatenate(reads).astype(np.int32)
    offs = np.zeros(len(reads) + 1, np.int64)
    np.cumsum([len(r) for r in reads], out=offs[1:])
    eng = Engine(0)
    try:
        for rep in range(3):             # (creation races differ from run to run)
            eng.set_reads(toks, offs, 2 * V)
            eng.build(k)
            c = eng.counts()
            assert c["exact_keys"] == 1
            want = token_oracle.build(toks, offs, k, 2 * V)
            nodes, edges = eng.nodes(), eng.edges()
            tok_node, tok_dir = eng.read_nodes()
            assert c["n_windows"] == want["n_windows"]
            assert np.array_equal(nodes["tokens"], want["tokens"])
            assert np.array_equal(nodes["coverage"], want["coverage"])
            assert np.array_equal(nodes["first_dir"], want["first_dir"])
            for a, b in (("src", "src"), ("tgt", "tgt"), ("sdir", "sdir"), ("tdir", "tdir"), ("coverage", "ecov")):
                assert np.array_equal(edges[a], want[b]), a
            assert np.array_equal(tok_node, want["tok_node"]) and np.array_equal(tok_dir, want["tok_dir"])
    finally:
        eng.close()


@pytest.mark.parametrize("lone,shards", [("0", "0"), ("1", "0"), ("0", "1"), ("1", "1")])
def test_edge_classes_of_single_nodes(lone, shards, monkeypatch):
    """Edge classes with an end node of coverage 1 bypass the edge table (k_edges_v<.., LONE>): they occur once — or
    twice, when the windows either side of the single window are the same node in the same direction (a period-2
    stretch: a b a b a b a at k = 5).  Tiny vocabularies make every such arrangement, across tile boundaries and read
    ends, next to planted period-2 reads; against the sequential C oracle, with the shortcut forced on and off — and
    with the claim ids of both table passes taken from one counter or from the 64 shard counters (x_chunk_claim: ids
    nobody took in between), which large inputs use."""
    import token_oracle
    from amira_amd import Engine
    monkeypatch.setenv("AMG_EDGE_LONE", lone)
    monkeypatch.setenv("AMG_CLAIM_SHARDS", shards)
    if lone == shards == "1":  # and with a head launch: the first tiles take their claims densely from a counter of their own
        monkeypatch.setenv("AMG_X_HEAD_TILES", "2")
    eng = Engine(0)
    try:
        for seed, V, n_reads, k in ((1, 3, 700, 5), (2, 2, 300, 5), (3, 12, 2500, 3), (4, 40, 3000, 5), (5, 6, 300, 3)):
            rng = np.random.default_rng(seed)
            reads = [rng.integers(0, 2 * V, int(rng.integers(1, 15))) for _ in range(n_reads)]
            for _ in range(n_reads // 10):  # a b a b ... with strands of their own, forwards or reverse-complemented
                a, b = rng.integers(0, 2 * V, 2)
                r = np.where(np.arange(int(rng.integers(k + 2, k + 6))) % 2 == 0, a, b)
                reads.append(r if rng.random() < 0.5 else 2 * V - 1 - r[::-1])
            order = rng.permutation(len(reads))
            reads = [reads[i] for i in order]
            toks = np.concatenate(reads).astype(np.int32)
            offs = np.zeros(len(reads) + 1, np.int64)
            np.cumsum([len(r) for r in reads], out=offs[1:])
            eng.set_reads(toks, offs, 2 * V)
            eng.build(k)
            want = token_oracle.build(toks, offs, k, 2 * V)
            nodes, edges = eng.nodes(), eng.edges()
            tok_node, tok_dir = eng.read_nodes()
            assert (nodes["coverage"] == 1).sum() > 20, "the case has no single nodes"
            assert np.array_equal(nodes["tokens"], want["tokens"])
            assert np.array_equal(nodes["coverage"], want["coverage"])
            for a, b in (("src", "src"), ("tgt", "tgt"), ("sdir", "sdir"), ("tdir", "tdir"), ("coverage", "ecov")):
                assert np.array_equal(edges[a], want[b]), (seed, a)
            assert np.array_equal(tok_node, want["tok_node"]) and np.array_equal(tok_dir, want["tok_dir"])
    finally:
        eng.close()
