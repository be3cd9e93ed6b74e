"""The restated sourmash MinHash (oracle/amira_oracle/minhash.py) against the vectors that pin it:
MurmurHash3 known answers, sourmash's own `hash_murmur("ACG")` value, and the containments the
reference's test suite holds for its FASTQ fixture (tests/test_gene_mer_graph.py:5119-5155 and
:4528-4607).  CPU only."""
import procedures as P
from amira_oracle import GeneMerGraph
from amira_oracle.minhash import MinHash, max_hash_for_scaled, murmurhash3_x64_128_h1


def test_murmurhash3_known_answers():
    assert murmurhash3_x64_128_h1(b"hello", 0) == 0xCBD8A7B341BD9B02
    assert murmurhash3_x64_128_h1(b"The quick brown fox jumps over the lazy dog", 0) == 0xE34BBC7BBC071B6C
    assert murmurhash3_x64_128_h1(b"ACG", 42) == 1731421407650554201       # sourmash tests: hash_murmur("ACG")
    assert max_hash_for_scaled(1) == 2 ** 64 - 1 and max_hash_for_scaled(10) == 1844674407370955264


def test_sketch_rules():
    a, b = MinHash(n=0, ksize=5, scaled=1), MinHash(n=0, ksize=5, scaled=1)
    a.add_sequence("ACGTTGCATG", force=True)
    b.add_sequence("catgcaacgt", force=True)                 # reverse complement, lower case: same k-mers
    assert set(a.hashes) == set(b.hashes) and len(a) == 6
    c = MinHash(n=0, ksize=5, scaled=1)
    c.add_sequence("ACGTTNGCATG", force=True)                # windows over the N are skipped
    assert len(c) == 2
    d = MinHash(n=0, ksize=5, scaled=1)
    d.add_sequence("ACG", force=True)                        # shorter than k: nothing
    assert len(d) == 0


def test_reference_held_containments_on_the_fastq_fixture():
    """tests/test_gene_mer_graph.py:5119-5155, restated: the two filtered bubble paths of the fixture
    share 0.9105839416058394 / 0.9091323161011159 of their node sketches' hashes"""
    import dump as D
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


def test_assess_connectivity_vectors():
    """tests/test_gene_mer_graph.py:4528-4607: seq1 / seq2 differ by two bases, seq3 is unrelated"""
    seqs = ["ATGGTCTCCGAGCTGCAGCGCCAGCTGGCGCTGCATCGGCAGACCCGCGGTGTAGGGTCTTCGTCGACTGCTT",
            "ATGGTCTCCGAGCTGCAGCGCCAGCTTTCGCTGCATCGGCAGACCCGCGGTGTAGGGTCTTCGTCGACTGCTT",
            "ATGAGTAGTAGGTCGTCGATCGTCAGCTGGATCTGAGATTCGGATTCGGCGGCTATCGGCTAGTCGACTGCTT"]
    m = []
    for s in seqs:
        mh = MinHash(n=0, ksize=9, scaled=1)
        mh.add_sequence(s, force=True)
        m.append(mh)
    c12 = max(m[0].contained_by(m[1]), m[1].contained_by(m[0]))
    c13 = max(m[0].contained_by(m[2]), m[2].contained_by(m[0]))
    assert 0 < c13 < 0.9 <= c12 < 1        # the reference's thresholds 0 / 0.9 / 1 separate exactly these
