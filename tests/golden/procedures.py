"""The procedures whose results are pinned in goldens.json.

Each takes the implementation under test as a parameter (``impl``: an object with
``GeneMerGraph``, ``Gene``, ``GeneMer`` attributes), so the SAME code is run against
the real reference (gen_goldens.py, build container only), the CPU oracle
(tests/test_oracle_goldens.py) and the HIP product (tests/test_product_goldens.py).
"""
import importlib.util
import os

import dump as D

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))

_spec = importlib.util.spec_from_file_location(
    "_amg_synth", os.path.join(ROOT, "amira_amd", "synth.py")
)
synth = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(synth)


class FakeFastq(dict):
    """Only len(fastq[read]["sequence"]) is consulted (construct_graph.py:1685)."""

    def __init__(self, lengths):
        super().__init__({r: {"sequence": range(n)} for r, n in lengths.items()})


def fixture(name):
    calls = D.load_fixture(f"complex_gene_calls_{name}")
    try:
        pos = D.load_fixture(f"complex_gene_positions_{name}")
    except FileNotFoundError:
        pos = None
    return calls, pos


def hash_samples(g, n=60):
    nodes = [[g.get_gene_mer_label(v), str(h)] for h, v in list(g.get_nodes().items())[:n]]
    edges = [D._edge_desc(g, e)[:4] + [str(h)] for h, e in list(g.get_edges().items())[:n]]
    return {"node_hashes": nodes, "edge_hashes": edges}


def p_fixture(impl, name, k):
    """build -> filter_graph(3,1) -> correct_reads on one of the reference's JSON fixtures."""
    calls, pos = fixture(name)
    if pos is not None:
        pos = {r: list(v) for r, v in pos.items()}
    g = impl.GeneMerGraph(calls, k, pos)
    entry = {"k": k, "build": D.summarise(D.dump_graph(g)), "hashes": hash_samples(g)}
    g.filter_graph(3, 1)
    entry["filter_3_1"] = D.summarise(D.dump_graph(g))
    if pos is not None:
        lengths = {r: (pos[r][-1][1] + 200 if pos[r] else 100) for r in pos}
        genes, gpos = g.correct_reads(FakeFastq(lengths))
        entry["correct_after_filter"] = D.summarise_corrected(D.dump_corrected(genes, gpos))
    return entry


def synth_inputs(seed, N, L, V, err, n_amr=0):
    ids, sts = synth.loop_reads(seed, N, L, V, err, n_amr)
    reads = synth.to_read_dict(ids, sts, synth.gene_names(V, n_amr))
    return reads, synth.positions_for(reads), FakeFastq(synth.fake_fastq_lengths(reads))


def p_sweep(impl, seed, N, L, V, k, err):
    """SURVEY Appendix C cfg-3 sweep (graph_utils.py:145-166)."""
    reads, pos, fq = synth_inputs(seed, N, L, V, err)
    entry = {"seed": seed, "N": N, "L": L, "V": V, "k": k, "err": err}
    g1 = impl.GeneMerGraph(reads, k, pos)
    entry["build1"] = D.summarise(D.dump_graph(g1))
    g1.filter_graph(3, 1)
    entry["filtered1"] = D.summarise(D.dump_graph(g1))
    r2, p2 = g1.correct_reads(fq)
    entry["corrected1"] = D.summarise_corrected(D.dump_corrected(r2, p2))
    g2 = impl.GeneMerGraph(r2, k, p2)
    entry["build2"] = D.summarise(D.dump_graph(g2))
    removed = g2.remove_short_linear_paths(k)
    entry["n_removed"] = len(removed)
    entry["removed_digest"] = D.digest(sorted(str(h) for h in removed))
    entry["clipped2"] = D.summarise(D.dump_graph(g2))
    r3, p3 = g2.correct_reads(fq)
    entry["corrected2"] = D.summarise_corrected(D.dump_corrected(r3, p3))
    g3 = impl.GeneMerGraph(r3, k, p3)
    entry["build3"] = D.summarise(D.dump_graph(g3))
    return entry


def _cluster_entry(g, genes):
    clustered, path_reads = g.assign_reads_to_genes(genes, 1, {}, None)
    canon = D.canon_clusters(clustered, path_reads)
    anon = D.anon_clusters(clustered, path_reads, genes)
    return {
        "anon_clusters_digest": D.digest(anon["clusters"]),
        "anon_path_reads_digest": D.digest(anon["path_reads"]),
        "genes": genes,
        "n_alleles": len(canon["clusters"]),
        "allele_sizes": [[c[0], c[1], c[2], len(c[3])] for c in canon["clusters"]],
        "clusters_digest": D.digest(canon["clusters"]),
        "path_reads_digest": D.digest(canon["path_reads"]),
        "path_read_sizes": sorted(len(v) for _, v in canon["path_reads"]),
    }


def p_cluster_fixture(impl, name, k, genes):
    calls, pos = fixture(name)
    return _cluster_entry(impl.GeneMerGraph(calls, k, pos), genes)


def p_planted(impl, seed, N, L, V, k):
    reads, pos, _ = synth_inputs(seed, N, L, V, 0.0, n_amr=10)
    return _cluster_entry(impl.GeneMerGraph(reads, k, pos), [f"amr{j}" for j in range(10)])


def p_misc_passes(impl, name, k):
    """remove_low_coverage_components / remove_junk_reads / get_valid_reads_only /
    remove_non_AMR_associated_nodes on a fixture (pipeline order of __main__.py:576-597)."""
    calls, pos = fixture(name)
    pos = {r: list(v) for r, v in pos.items()}
    g = impl.GeneMerGraph(calls, k, pos)
    entry = {}
    g.remove_low_coverage_components(5)
    entry["after_low_cov_components"] = D.summarise(D.dump_graph(g))
    g.filter_graph(2, 1)
    keep, keep_pos, drop, drop_pos = g.remove_junk_reads(0.80)
    entry["junk"] = {"kept": len(keep), "dropped": len(drop),
                     "kept_digest": D.digest(sorted(keep)), "dropped_digest": D.digest(sorted(drop))}
    entry["valid_reads_digest"] = D.digest(sorted(g.get_valid_reads_only()))
    return entry


def p_drivers(impl, name):
    """graph_utils drivers on a fixture: get_overall_mean_node_coverages (:299-313),
    choose_kmer_size (:258-296, seven builds k = 3..15) and remove_non_AMR_associated_nodes
    (construct_graph.py:2941-2959) for the three most frequent genes of the fixture."""
    from collections import Counter
    calls, pos = fixture(name)
    pos = {r: list(v) for r, v in pos.items()}
    counts = Counter(g[1:] for genes in calls.values() for g in genes)
    genes = [g for g, _ in sorted(counts.items(), key=lambda kv: (-kv[1], kv[0]))[:3]]
    g = impl.GeneMerGraph(calls, 3, pos)
    cov = impl.get_overall_mean_node_coverages(g)
    entry = {"genes": genes, "mean_cov": {str(k): float(v) for k, v in cov.items()}}
    entry["chosen_k"] = impl.choose_kmer_size(cov[3], calls, 1, pos, genes)
    entry["chosen_k_low_cov"] = impl.choose_kmer_size(1, calls, 1, pos, genes)
    g.remove_non_AMR_associated_nodes(genes[:1])
    entry["after_remove_non_AMR"] = D.summarise(D.dump_graph(g))
    return entry


def p_values(impl):
    """Known-answer hashes of the value objects (pins the host-side sha256 hashing)."""
    names = ["+gene1", "-gene2", "+blaTEM-1", "-group_1234", "+g0", "+two words"]
    out = {"gene_hashes": [[n, str(impl.Gene(n).__hash__())] for n in names]}
    rows = []
    for m in (["+gene1", "-gene2", "+gene3"], ["-g5", "-g4", "+g3", "+g2", "-g1"], ["+a"]):
        gm = impl.GeneMer([impl.Gene(x) for x in m])
        rows.append(
            [
                m,
                gm.get_geneMerDirection(),
                [("+" if g.get_strand() == 1 else "-") + g.get_name()
                 for g in gm.get_canonical_geneMer()],
                str(gm.__hash__()),
            ]
        )
    out["genemer_pins"] = rows
    return out


# name -> (procedure, args, slow?)   slow cases are skipped by `gen_goldens.py --quick`
CASES = {"values": (p_values, (), False)}
for _n in ("five", "six", "seven", "eight", "four", "three", "nine"):
    for _k in (3, 5):
        CASES[f"fixture_{_n}_k{_k}"] = (p_fixture, (_n, _k), _n in ("three", "nine"))
CASES["fixture_one_k3"] = (p_fixture, ("one", 3), True)
CASES["sweep_s20250908"] = (p_sweep, (20250908, 3000, 40, 2000, 5, 0.02), True)
CASES["sweep_small_k5"] = (p_sweep, (7, 400, 30, 300, 5, 0.03), False)
CASES["sweep_small_k3"] = (p_sweep, (11, 400, 24, 200, 3, 0.03), False)
CASES["sweep_small_k7"] = (p_sweep, (13, 300, 40, 250, 7, 0.02), False)
CASES["sweep_dense_k5"] = (p_sweep, (17, 800, 40, 150, 5, 0.05), False)
CASES["misc_nine_k3"] = (p_misc_passes, ("nine", 3), False)
CASES["misc_four_k5"] = (p_misc_passes, ("four", 5), False)
CASES["drivers_eight"] = (p_drivers, ("eight",), False)
CASES["drivers_nine"] = (p_drivers, ("nine",), False)
CASES["cluster_eight_k3"] = (p_cluster_fixture, ("eight", 3, ["dfrA17"]), False)
CASES["planted_s20250909"] = (p_planted, (20250909, 1500, 40, 1000, 5), True)
CASES["planted_small"] = (p_planted, (5, 300, 40, 1000, 5), False)
