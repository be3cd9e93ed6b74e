"""The native block search (amg_cluster_full_blocks, host C++ behind the C ABI) against the reference's procedure as
the oracle restates it (path_finding_utils.py:88-247) and against the product's pure-Python block search, on random
read sets: the same full blocks in the same ORDER — the order comes from how Python sets of node-hash tuples iterate,
which the native code reproduces.  Node "hashes" are random 256-bit integers, as the real ones are.  CPU only."""
import os
import random
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def _reads_case(rng, with_none):
    n_genome = rng.randint(14, 60)
    label = {}

    def big(x):   # a stable random 256-bit "node hash" per label
        if x not in label:
            label[x] = rng.getrandbits(256) - (1 << 255)
        return label[x]

    genome = list(range(100, 100 + n_genome))
    loci = [genome]
    for _ in range(rng.randint(0, 3)):   # further loci sharing a stretch (copies of a gene in different contexts)
        a = rng.randrange(2, len(genome) - 6)
        loci.append([1000 * len(loci) + i for i in range(rng.randint(3, 12))] + genome[a:a + rng.randint(2, 6)] +
                    [2000 * len(loci) + i for i in range(rng.randint(3, 12))])
    if rng.random() < 0.3:   # a tandem repeat: blocks that occur twice on a read
        at = rng.randrange(3, len(genome))
        genome[at:at] = genome[max(0, at - 3):at]
    reads = {}
    for i in range(rng.randint(4, 80)):
        src = rng.choice(loci)
        lo = rng.randrange(0, max(len(src) - 3, 1))
        hi = rng.randint(lo + 1, len(src) - 1) if lo + 1 < len(src) else lo
        nodes = [big(x) for x in src[lo:hi + 1]]
        if rng.random() < 0.5:
            nodes = nodes[::-1]
        if with_none and rng.random() < 0.15:
            nodes[rng.randrange(len(nodes))] = None
        if rng.random() < 0.05:
            nodes = [nodes[0]] * rng.randint(1, 3)        # one distinct node: no "_reverse" entry
        reads[f"r{i:03d}"] = nodes
    pool = sorted({n for v in reads.values() for n in v if n is not None})
    anchors = set(rng.sample(pool, min(len(pool), rng.randint(2, 6))))
    return reads, anchors


def _native(reads, anchors):
    from amira_amd import clustering
    assert clustering.emulation_ok()
    ids = {}
    for nodes in reads.values():
        for x in nodes:
            if x is not None:
                ids.setdefault(x, len(ids))
    for a in anchors:
        ids.setdefault(a, len(ids))
    back = [None] * len(ids)
    for h, i in ids.items():
        back[i] = h
    seq = np.asarray([(-2 if x is None else ids[x]) for nodes in reads.values() for x in nodes], np.int32)
    off = np.zeros(len(reads) + 1, np.int64)
    np.cumsum([len(v) for v in reads.values()], out=off[1:])
    anchor_list = list(anchors)
    rank = {h: i for i, h in enumerate(sorted(anchor_list))}
    py_hash = np.asarray([hash(h) for h in back], np.int64) if back else np.zeros(1, np.int64)
    blocks = clustering.full_block_ids(seq, off, [ids[a] for a in anchor_list], [rank[a] for a in anchor_list],
                                       py_hash, hash(None))
    return [tuple(None if i == -2 else back[i] for i in b.tolist()) for b in blocks]


def test_native_blocks_equal_the_reference_procedure():
    from amira_oracle import paths as ref
    rng = random.Random(20261004)
    compared = 0
    for _ in range(1200):
        reads, anchors = _reads_case(rng, with_none=False)
        want = {}
        tree_r = ref.construct_suffix_tree({r: list(v) for r, v in reads.items()})
        for a1 in anchors:   # (the reference iterates the anchor SET in both loops)
            suf = ref.get_suffixes_from_initial_tree(tree_r, a1)
            sub = ref.Tree({r: list(reversed(s)) for r, s in suf.items()})
            ref.process_anchors(sub, anchors, a1, want, reads, tree_r, 1)
        got = _native(reads, anchors)
        assert got == list(want)
        compared += len(want)
    assert compared > 2000


def test_native_blocks_equal_the_python_block_search_with_masked_nodes():
    from amira_amd import path_finding_utils as pf
    rng = random.Random(77)
    compared = 0
    for _ in range(1200):
        reads, anchors = _reads_case(rng, with_none=True)
        want = {}
        tree = pf.construct_suffix_tree({r: list(v) for r, v in reads.items()})
        for a1 in anchors:
            sub = tree.reversed_suffix_tree(a1)
            pf.process_anchors(sub, anchors, a1, want, reads, tree, 1)
        got = _native(reads, anchors)
        assert got == list(want)
        compared += len(want)
    assert compared > 2000


def test_native_anchor_stats_equal_the_reference_read_loop():
    """amg_cluster_anchor_stats against the read loop of get_AMR_anchors (construct_graph.py:2644-2676) written out
    per node: reads in read order, positions ascending, stop at a single-window read or at an interior occurrence
    with a non-AMR neighbour"""
    from amira_amd import clustering
    rng = random.Random(99)
    for _ in range(400):
        n_nodes = rng.randint(3, 30)
        reads = []
        for _ in range(rng.randint(1, 40)):
            n = rng.choice([1, 1, 2, 3, 5, 9, 20])
            reads.append([rng.choice([-2] + list(range(n_nodes))) for _ in range(n)])
        amr = rng.sample(range(n_nodes), rng.randint(1, min(6, n_nodes)))
        amr_set = set(amr)
        want = []
        for h in amr:
            flags, singletons, is_anchor = [], [], False
            for on_read in reads:
                if h not in on_read:
                    continue
                if len(on_read) == 1 and on_read[0] == h:
                    singletons.append(True)
                    flags.append(True)
                    break
                singletons.append(False)
                for idx, x in enumerate(on_read):
                    if x != h:
                        continue
                    if idx == 0 or idx == len(on_read) - 1:
                        flags.append(True)
                        continue
                    if on_read[idx - 1] not in amr_set or on_read[idx + 1] not in amr_set:
                        is_anchor = True
                        break
                    flags.append(False)
                if is_anchor:
                    break
            # all(singletons) as the vectorised product code states it: no occurrence at all, or the first is a
            # single-window read (then the loop stopped there)
            want.append([int(is_anchor), int(all(singletons)), len(flags), flags.count(True)])
        # the reads reach the native code in another order (a Python set's), with the permutation back to read order
        perm = list(range(len(reads)))
        rng.shuffle(perm)
        shuffled = [reads[i] for i in perm]
        order = np.argsort(np.asarray(perm), kind="stable")
        seq = np.asarray([x for r in shuffled for x in r], np.int32)
        off = np.zeros(len(shuffled) + 1, np.int64)
        np.cumsum([len(r) for r in shuffled], out=off[1:])
        got = clustering.anchor_stats(seq, off, order, amr, n_nodes)
        assert got.tolist() == want
