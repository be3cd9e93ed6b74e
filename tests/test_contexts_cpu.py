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
