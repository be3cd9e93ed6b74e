"""Pin the CPU oracle against goldens produced by the REAL reference.

goldens.json was written by tests/golden/gen_goldens.py, which imports
/root/reference/amira in the build container and runs tests/golden/procedures.py
against it.  Here the same procedures run against oracle/amira_oracle and must give
identical digests, counts, samples and 256-bit hashes.
"""
import json
import os
import types

import pytest

import procedures as P
from seed0 import run_case_seed0
from amira_oracle import Gene, GeneMer, GeneMerGraph
from amira_oracle.driver import choose_kmer_size, get_overall_mean_node_coverages, iterative_bubble_popping
from amira_oracle.front_end import process_pandora_json, write_pandora_gene_calls

ORACLE = types.SimpleNamespace(GeneMerGraph=GeneMerGraph, Gene=Gene, GeneMer=GeneMer,
                               choose_kmer_size=choose_kmer_size,
                               get_overall_mean_node_coverages=get_overall_mean_node_coverages,
                               iterative_bubble_popping=iterative_bubble_popping,
                               process_pandora_json=process_pandora_json,
                               write_pandora_gene_calls=write_pandora_gene_calls)
GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "goldens.json")))

# every case runs by default (fixture_one_k3 = BASELINE config 1: 13 s, sweep_s20250908: 23 s, planted_dense_k5: 4 s)
# except the one that takes minutes in pure Python (bubble popping on fixture nine, ~200 s): AMG_SLOW=1 runs it too
HEAVY = {"bubbles_synth_nine_k3"}


def _cases():
    for name in P.CASES:
        marks = [pytest.mark.slow] if name in HEAVY else []
        yield pytest.param(name, marks=marks)


@pytest.mark.parametrize("name", list(_cases()))
def test_oracle_matches_reference(name):
    if name in HEAVY and not os.environ.get("AMG_SLOW"):
        pytest.skip("set AMG_SLOW=1 to run the largest oracle case (~200 s)")
    proc, args, _ = P.CASES[name]
    assert name in GOLD, "golden missing: regenerate with tests/golden/gen_goldens.py"
    if proc in (P.p_planted, P.p_cluster_fixture, P.p_front_end) and os.environ.get("PYTHONHASHSEED") != "0":
        # the reference's clustering leaks set-of-str iteration order into its result
        # (construct_graph.py:1497-1504, :2769-2777), and so does process_pandora_json's list(set)
        # (pre_processing.py:61); goldens were taken at PYTHONHASHSEED=0, so these cases re-run in a child
        # with that seed
        got = run_case_seed0("oracle", name)
    else:
        got = json.loads(json.dumps(proc(ORACLE, *args)))
    assert got == GOLD[name]


def test_known_answers_from_reference_tests():
    # tests/test_gene_mer_graph.py:64-68, :116-118 and SURVEY Appendix G
    g = GeneMerGraph({"read1": ["+gene1", "-gene2", "+gene3", "-gene4"],
                      "read2": ["+gene1", "-gene2", "+gene3"]}, 3)
    covs = sorted(n.get_node_coverage() for n in g.all_nodes())
    assert len(g.get_nodes()) == 2 and covs == [1, 2]
    b = GOLD["fixture_three_k3"]["build"]
    assert (b["n_nodes"], b["n_edges"], b["sum_node_cov"], b["sum_edge_cov"]) == (2079, 4160, 84459, 167762)
    b = GOLD["fixture_nine_k5"]
    assert (b["build"]["n_nodes"], b["build"]["n_edges"], b["build"]["n_short"]) == (1031, 2178, 28)
    assert (b["filter_3_1"]["n_nodes"], b["filter_3_1"]["n_edges"], b["filter_3_1"]["n_to_correct"]) == (914, 1886, 62)
    s = GOLD["sweep_s20250908"]
    assert (s["build1"]["n_nodes"], s["filtered1"]["n_nodes"], s["filtered1"]["n_to_correct"]) == (12410, 2000, 1659)
    assert (s["build2"]["n_nodes"], s["n_removed"], s["build3"]["n_nodes"], s["build3"]["n_edges"]) == (2876, 33, 2843, 5876)


def test_self_loop_and_hairpin_rules():
    # tandem self-loop = one edge, +2 per traversal
    g = GeneMerGraph({"r": ["-gene4"] * 5}, 3)
    (node,) = g.all_nodes()
    (edge,) = g.get_edges().values()
    assert node.get_node_coverage() == 3 and edge.get_edge_coverage() == 4
    assert len(node.get_forward_edge_hashes()) + len(node.get_backward_edge_hashes()) == 1
    # hairpin: one node visited in both directions
    h = GeneMerGraph({"r": ["+a", "+b", "-b", "-a"]}, 3)
    assert len(h.get_nodes()) == 1 and len(h.get_edges()) == 1
    assert h.get_readNodeDirections()["r"] == [1, -1] or h.get_readNodeDirections()["r"] == [-1, 1]
