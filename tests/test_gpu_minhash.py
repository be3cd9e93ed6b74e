"""amg_minhash (HIP) against the restated sourmash MinHash of the oracle, and the reference-held
containments through the product's bubble-popping API."""
import numpy as np
import pytest

import procedures as P

pytestmark = pytest.mark.gpu


def _random_segments(rng, n, lo, hi):
    alphabet = np.frombuffer(b"ACGTacgtNRY", dtype=np.uint8)
    probs = np.array([0.23, 0.23, 0.23, 0.23, 0.015, 0.015, 0.015, 0.015, 0.01, 0.005, 0.005])
    segs = []
    for _ in range(n):
        L = int(rng.integers(lo, hi))
        segs.append(bytes(rng.choice(alphabet, size=L, p=probs / probs.sum())).decode())
    return segs


@pytest.mark.parametrize("ksize,scaled", [(11, 10), (9, 1), (21, 3), (4, 1), (32, 2),
                                          (1, 1), (8, 1), (16, 1), (17, 1), (24, 2), (25, 1), (31, 1)])   # (word boundaries of amg_kmer.h)
def test_device_sketch_equals_oracle(ksize, scaled):
    from amira_amd import Engine
    from amira_oracle.minhash import MinHash
    rng = np.random.default_rng(ksize * 100 + scaled)
    segs = _random_segments(rng, 200, 0, 3000) + ["", "ACG", "N" * 50, "ACGT" * 700]
    sets = [int(x) for x in rng.integers(0, 17, len(segs))]
    eng = Engine(0)
    try:
        got = eng.minhash(segs, sets, ksize, scaled)
    finally:
        eng.close()
    want = {s: MinHash(n=0, ksize=ksize, scaled=scaled) for s in set(sets)}
    for seg, s in zip(segs, sets):
        want[s].add_sequence(seg, force=True)
    assert set(got) == set(want)
    for s in want:
        assert got[s] == set(want[s].hashes), s
    assert sum(len(v) for v in got.values()) > 0


def test_reference_held_containments_through_the_product():
    """tests/test_gene_mer_graph.py:5119-5155 through amira_amd.GeneMerGraph (device sketches)"""
    import dump as D
    from amira_amd import GeneMerGraph
    calls, pos = D.load_fixture("test_path_calls"), D.load_fixture("test_path_positions")
    g = GeneMerGraph(calls, 3, pos)
    fq = P.real_fastq()
    starts = g.identify_potential_bubble_starts()
    checked = 0
    for component in g.components():
        if component not in starts:
            continue
        unique = g.get_all_paths_between_junctions_in_component(starts[component], g.get_kmerSize() * 3, 1)
        filtered = sorted(g.filter_paths_between_bubble_starts(unique), key=lambda x: len(x[0]), reverse=True)
        sketches = g.get_minhashes_for_paths(filtered, fq, 1)
        m1 = g.get_minimizers_from_minhashes([n[0] for n in filtered[0][0]], sketches)
        m2 = g.get_minimizers_from_minhashes([n[0] for n in filtered[1][0]], sketches)
        assert len(m1 & m2) / len(m1) == 0.9105839416058394
        assert len(m1 & m2) / len(m2) == 0.9091323161011159
        checked += 1
    assert checked == 1
