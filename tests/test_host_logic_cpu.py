"""CPU tests of the product's host-side Python (no GPU needed): value objects against the
reference's golden hashes, path_finding_utils against the oracle's restatement on random
inputs, synthetic generators."""
import json
import os
import random
import types

import numpy as np

import procedures as P

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "goldens.json")))


def test_product_value_objects_match_reference_hashes():
    from amira_amd import Gene, GeneMer
    got = json.loads(json.dumps(P.p_values(types.SimpleNamespace(Gene=Gene, GeneMer=GeneMer))))
    assert got == GOLD["values"]


def test_product_value_objects_behave_like_oracle():
    from amira_amd import Edge, Gene, GeneMer, Node, Read
    import amira_oracle as O
    rng = random.Random(1)
    names = [f"g{i}" for i in range(12)]
    for _ in range(200):
        genes = [rng.choice("+-") + rng.choice(names) for _ in range(rng.choice([1, 3, 5]))]
        try:
            a = GeneMer([Gene(g) for g in genes])
        except AssertionError:
            try:
                O.GeneMer([O.Gene(g) for g in genes])
                raise AssertionError("oracle accepted a gene-mer the product rejected")
            except AssertionError:
                continue
        b = O.GeneMer([O.Gene(g) for g in genes])
        assert a.__hash__() == b.__hash__() and a.get_geneMerDirection() == b.get_geneMerDirection()
        assert [str(g) for g in a.get_canonical_geneMer()] == [g.as_string() for g in b.get_canonical_geneMer()]
        assert [str(g) for g in a.get_rc_geneMer()] == [g.as_string() for g in b.get_rc_geneMer()]
    genes = ["+a", "-b", "+c", "+d", "-e"]
    ra, rb = Read("r", genes, [(i, i + 1) for i in range(5)]), O.Read("r", genes, [(i, i + 1) for i in range(5)])
    ma, sa = ra.get_geneMers(3)
    mb, sb = rb.get_geneMers(3)
    assert sa == sb and [m.__hash__() for m in ma] == [m.__hash__() for m in mb]
    assert Read("r", genes[:2]).get_geneMers(3) == ([], [])
    na, nb = [Node(m) for m in ma], [O.Node(m) for m in mb]
    for d1, d2 in ((1, 1), (1, -1), (-1, 1), (-1, -1)):
        assert Edge(na[0], na[1], d1, d2).__hash__() == O.Edge(nb[0], nb[1], d1, d2).__hash__()
    assert Edge(na[0], na[1], 1, 1).__hash__() == Edge(na[0], na[1], -1, -1).__hash__()
    n = na[0]
    n.add_read("x"); n.add_read("x"); n.add_read("y")
    assert n.get_list_of_reads() == ["x", "y"] and n.increment_node_coverage() == 1
    n.add_forward_edge_hash(5); n.add_forward_edge_hash(5); n.add_backward_edge_hash(7)
    assert n.get_forward_edge_hashes() == [5] and n.get_backward_edge_hashes() == [7]
    n.remove_forward_edge_hash(5)
    assert n.get_forward_edge_hashes() == []


def test_path_finding_utils_match_oracle_on_random_inputs():
    from amira_amd import path_finding_utils as A
    from amira_oracle import paths as B
    rng = random.Random(7)
    for _ in range(60):
        seqs = {f"r{i}": [rng.randrange(1, 9) for _ in range(rng.randrange(1, 12))] for i in range(8)}
        ta, tb = A.construct_suffix_tree(dict(seqs)), B.construct_suffix_tree(dict(seqs))
        for a1 in range(1, 9):
            assert A.get_suffixes_from_initial_tree(ta, a1) == B.get_suffixes_from_initial_tree(tb, a1)
        anchors = set(rng.sample(range(1, 9), 3))
        fa, fb = {}, {}
        for a1 in sorted(anchors):
            sa = A.get_suffixes_from_initial_tree(ta, a1)
            A.process_anchors(A.Tree({r: list(reversed(s)) for r, s in sa.items()}), anchors, a1, fa, seqs, ta, 1)
            sb = B.get_suffixes_from_initial_tree(tb, a1)
            B.process_anchors(B.Tree({r: list(reversed(s)) for r, s in sb.items()}), anchors, a1, fb, seqs, tb, 1)
        assert fa == fb
        assert A.filter_blocks(fa) == B.filter_blocks(fb)
        paths = {tuple(rng.randrange(1, 5) for _ in range(rng.randrange(0, 5))) for _ in range(10)}
        assert A.cluster_downstream_adjacent_paths(paths) == B.cluster_downstream_adjacent_paths(paths)
        assert A.cluster_upstream_adjacent_paths(paths) == B.cluster_upstream_adjacent_paths(paths)
        genes = {f"r{i}": [rng.choice("+-") + rng.choice("abcX") for _ in range(rng.randrange(2, 9))] for i in range(10)}
        lst = genes["r0"] + ["+X"]
        args = (rng.randrange(1, len(lst) + 1), 1, "X", lst, genes)
        assert A.process_combinations_for_i(args) == B.process_combinations_for_i(args)
        main = [rng.randrange(3) for _ in range(12)]
        sub = main[3:5]
        assert A.find_sublist_indices(main, sub) == B.find_sublist_indices(main, sub)
        assert A.is_sublist(main, sub) and A.is_sublist(main, [9]) == B.is_sublist(main, [9])


def test_block_generator_matches_read_ranges_and_loop_generator_known_answer():
    from amira_amd import synth
    a, sa = synth.block_reads(5, 0, 300, 20, 100, 0.05)
    b, sb = synth.block_reads(5, 100, 250, 20, 100, 0.05)
    assert np.array_equal(a[100:250], b) and np.array_equal(sa[100:250], sb)  # shards agree with the whole
    ids, sts = synth.loop_reads(20250905, 50, 40, 5000)
    reads = synth.to_read_dict(ids, sts, synth.gene_names(5000))
    assert len(reads) == 50 and all(len(v) == 40 for v in reads.values())
    assert synth.positions_for(reads)["r0000000"][1] == [1000, 1899]
    assert synth.fake_fastq_lengths(reads)["r0000000"] == 40100


def test_lazy_clip_hashes_behave_like_the_list():
    """remove_short_linear_paths with array-backed inputs returns its node hashes as a sequence that computes them when
    looked at: length without hashing, then list semantics (iteration, indexing, equality, sorted)"""
    import numpy as np
    from amira_amd.construct_gene import hashlib_hash
    from amira_amd.construct_graph import _LazyHashes
    from amira_amd.tokens import Vocabulary
    vocab = Vocabulary([f"g{i}" for i in range(9)])
    rows = np.array([[9, 10, 11], [8, 7, 6], [12, 3, 14]], np.int32)
    want = [hashlib_hash(tuple(vocab.signed_hash(int(t)) for t in row)) for row in rows]
    lazy = _LazyHashes(rows, vocab)
    assert len(lazy) == 3 and lazy._made is None          # nothing hashed yet
    assert list(lazy) == want and lazy[1] == want[1] and lazy == want and sorted(lazy) == sorted(want)
    assert want[2] in lazy and lazy == _LazyHashes(rows.copy(), vocab)
    assert len(_LazyHashes(rows[:0], vocab)) == 0 and list(_LazyHashes(rows[:0], vocab)) == []
