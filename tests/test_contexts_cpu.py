"""generate_contexts of the product (amira_amd/path_finding_utils.py: only the last non-canonical read of a block
and the canonical reads after it are looked at, lists already in a suffix-closed context set are skipped) against
the read-by-read loop of the reference as the oracle restates it (path_finding_utils.py:150-215): same contexts (keys
in the same order, same sets, iterating alike) and same duplicate flags on random inputs with repeats, reversed
blocks and blocks that occur more than once.  CPU only; importing the product's module needs no device."""
import importlib.util
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def _product_module():
    # path_finding_utils alone: the package __init__ would load libamg.so (fine here, but not needed)
    spec = importlib.util.spec_from_file_location("pfu_product", os.path.join(ROOT, "amira_amd", "path_finding_utils.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _case(rng):
    alphabet = rng.choice([4, 9, 30])
    genome = [rng.randrange(alphabet) for _ in range(rng.randint(12, 60))]
    if rng.random() < 0.5:   # a repeat, so that some blocks occur twice on a read
        at = rng.randrange(len(genome))
        genome[at:at] = genome[max(0, at - 4):at]
    a, b = sorted(rng.sample(range(len(genome)), 2))
    block = genome[a:b + 1]
    reads, block_reads = {}, {}
    for i in range(rng.randint(1, 25)):
        lo = rng.randint(0, a)
        hi = rng.randint(b, len(genome) - 1)
        nodes = genome[lo:hi + 1]
        flipped = rng.random() < 0.5
        if flipped:
            nodes = nodes[::-1]
        if rng.random() < 0.1:
            nodes = nodes + [None] + nodes[:2]
        rid = f"r{i}"
        reads[rid] = nodes
        blk = block[::-1] if flipped else list(block)
        if rng.random() < 0.15:
            blk = blk[::-1] if blk[::-1] in [nodes[j:j + len(blk)] for j in range(len(nodes))] else blk
        block_reads[rid] = blk
    return block_reads, reads


def test_generate_contexts_equals_the_read_by_read_loop():
    from amira_oracle import paths as ref
    prod = _product_module()
    rng = random.Random(20261003)
    n = 0
    for _ in range(3000):
        block_reads, reads = _case(rng)
        dup_a = {tuple(ref.get_canonical_representation(b)): False for b in block_reads.values()}
        dup_b = dict(dup_a)
        try:
            want = ref.generate_contexts({k: list(v) for k, v in block_reads.items()}, dup_a, reads)
        except AssertionError:
            continue   # a block that is not on its read: the reference asserts, nothing to compare
        got = prod.generate_contexts({k: list(v) for k, v in block_reads.items()}, dup_b, reads)
        assert list(got) == list(want)
        for key in want:
            for side in ("upstream", "downstream"):
                assert got[key][side] == want[key][side], (key, side)
                assert list(got[key][side]) == list(want[key][side]), (key, side, "iteration order")
        assert dup_b == dup_a
        n += 1
    assert n > 2000


def _reads_case(rng):
    """node lists of reads over a small 'genome' with a second locus sharing a stretch, some reversed"""
    alphabet = rng.choice([12, 40])
    genome = list(range(100, 100 + rng.randint(14, 40)))
    if rng.random() < 0.6:   # a second locus sharing a stretch (copies of a gene in different contexts)
        a = rng.randrange(2, len(genome) - 6)
        other = [200 + i for i in range(rng.randint(3, 8))] + genome[a:a + rng.randint(2, 5)] + [300 + i for i in range(rng.randint(3, 8))]
    else:
        other = []
    reads = {}
    for i in range(rng.randint(4, 40)):
        src = other if (other and rng.random() < 0.4) else genome
        lo = rng.randrange(0, max(len(src) - 3, 1))
        hi = rng.randint(lo + 1, len(src) - 1) if lo + 1 < len(src) else lo
        nodes = src[lo:hi + 1]
        if rng.random() < 0.5:
            nodes = nodes[::-1]
        reads[f"r{i:03d}"] = list(nodes)   # (no masked nodes: the reference's int() parse of a suffix would raise)
    pool = sorted({n for v in reads.values() for n in v if n is not None})
    anchors = set(rng.sample(pool, min(len(pool), rng.randint(2, 5))))
    return reads, anchors


def test_anchor_blocks_equal_the_reference_procedure():
    """get_full_paths' first stage (suffix tree -> sub-tree per anchor -> process_anchors) in the product — through
    the scan tree's generic calls and through its one-step `reversed_suffix_tree` — against the oracle's restatement:
    the same full blocks in the same order with the same supporting reads"""
    from amira_oracle import paths as ref
    prod = _product_module()
    rng = random.Random(424242)
    compared = 0
    for _ in range(1500):
        reads, anchors = _reads_case(rng)
        anchors_list = sorted(anchors)
        want = {}
        tree_r = ref.construct_suffix_tree({r: list(v) for r, v in reads.items()})
        for a1 in anchors_list:
            suf = ref.get_suffixes_from_initial_tree(tree_r, a1)
            sub = ref.Tree({r: list(reversed(s)) for r, s in suf.items()})
            ref.process_anchors(sub, anchors, a1, want, reads, tree_r, 1)
        for fast in (False, True):
            got = {}
            tree_p = prod.construct_suffix_tree({r: list(v) for r, v in reads.items()})
            for a1 in anchors_list:
                if fast:
                    sub = tree_p.reversed_suffix_tree(a1)
                else:
                    suf = prod.get_suffixes_from_initial_tree(tree_p, a1)
                    sub = prod.Tree({r: list(reversed(s)) for r, s in suf.items()})
                prod.process_anchors(sub, anchors, a1, got, reads, tree_p, 1)
            assert list(got) == list(want), fast
            assert got == want, fast
        compared += len(want)
    assert compared > 500


def test_blocks_from_coded_hits_equal_the_plain_loop():
    """get_blocks_from_subtree over find_all_coded (one shared block list per distinct suffix, found again through the
    integer codes) against the hit-by-hit loop over find_all: same block per read, same duplicates dict, same orders;
    and generate_contexts gives the same contexts whether the reads of a block share its list or not"""
    import random
    from amira_amd import path_finding_utils as pf

    class Plain(pf._ScanTree):
        find_all_coded = None   # hides the coded search: get_blocks_from_subtree takes the plain loop

    rng = random.Random(5)
    big = [10 ** 40 + i for i in range(7)]
    for _ in range(300):
        reads = {}
        for i in range(rng.randint(1, 9)):
            seq = [rng.choice(big + [None]) for _ in range(rng.randint(1, 10))]
            reads[f"r{i}"] = seq
        data = dict(reads)
        data.update({r + "_reverse": s[::-1] for r, s in reads.items() if len(set(s)) > 1})
        a1, a2 = rng.choice(big), rng.choice(big)
        if a1 == a2:
            continue
        anchors = [x for x in big]
        got_tree, want_tree = pf._ScanTree(data), Plain(data)
        sub_got, sub_want = got_tree.reversed_suffix_tree(a1), want_tree.reversed_suffix_tree(a1)
        sub_want.__class__ = Plain
        got = pf.get_blocks_from_subtree(sub_got, a2, anchors)
        want = pf.get_blocks_from_subtree(sub_want, a2, anchors)
        assert list(got[0].items()) == list(want[0].items()) and list(got[1].items()) == list(want[1].items())
        lists = {r: list(s) for r, s in data.items()}
        c_got = pf.generate_contexts(got[0], dict(got[1]), lists)
        c_want = pf.generate_contexts(want[0], dict(want[1]), lists)
        assert list(c_got) == list(c_want)
        for key in c_want:
            for side in ("upstream", "downstream"):
                assert list(c_got[key][side]) == list(c_want[key][side])   # same elements in the same iteration order
