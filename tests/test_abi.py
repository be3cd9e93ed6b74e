"""CPU-side checks of the drop-in boundary: libamg.so loads, exports every symbol that
include/amg.h declares, fails loudly without a GPU, and the host-side token encoding
reproduces the reference's ordering."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "amg.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(amg_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    so = os.path.join(ROOT, "amira_amd", "libamg.so")
    assert os.path.exists(so), "build libamg.so first: python -c 'import __graft_entry__ as g; g.build()'"
    lib = ctypes.CDLL(so)
    names = declared_symbols()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/amg.h but not exported"


def test_binding_covers_the_header():
    from amira_amd import _ffi
    assert sorted(_ffi.SYMBOLS) == declared_symbols()


def test_no_cpu_fallback():
    """Without a HIP device the product must raise, not compute on the host."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from amira_amd import Engine, _ffi
    with pytest.raises(_ffi.AmgError) as ei:
        Engine(0)
    assert "no HIP device" in str(ei.value)


def test_product_does_not_import_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "amira_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert "amira_oracle" not in src and "token_oracle" not in src, f


def test_token_order_equals_signed_hash_order():
    """tokens.py: integer order of tokens == order of the reference's signed gene hashes,
    flip == strand reversal (construct_gene.py:91-93, construct_gene_mer.py:15-39)."""
    from amira_amd.tokens import Vocabulary
    from amira_oracle import Gene, GeneMer
    names = [f"gene{i}" for i in range(40)] + ["blaTEM-1", "group_77", "a b"]
    v = Vocabulary([n.replace(" ", "_") for n in names])
    genes = [s + n for n in names for s in "+-"]
    by_tok = sorted(genes, key=v.token)
    by_hash = sorted(genes, key=lambda g: Gene(g).__hash__())
    assert by_tok == by_hash
    for g in genes:
        t = v.token(g)
        assert v.signed_hash(t) == Gene(g).__hash__()
        assert v.gene(v.flip(t)) == Gene(g).reverse_gene().as_string()
        assert v.gene(t) == Gene(g).as_string()
    # canonical choice on tokens == canonical choice on hashes
    import random
    rng = random.Random(5)
    for _ in range(300):
        k = rng.choice([1, 3, 5, 7])
        mer = [rng.choice(genes) for _ in range(k)]
        toks = [v.token(g) for g in mer]
        rc = [v.flip(t) for t in reversed(toks)]
        if toks == rc:
            continue
        gm = GeneMer([Gene(g) for g in mer])
        assert (1 if toks < rc else -1) == gm.get_geneMerDirection()
        canon = toks if toks < rc else rc
        assert [v.gene(t) for t in canon] == [x.as_string() for x in gm.get_canonical_geneMer()]


# methods / functions of the reference that the pipeline never reaches (last list; the
# multi-process merge the pipeline never uses; plots): everything else must exist under the same name
UNREACHED = {
    "GeneMerGraph": {"create_adjacency_matrix", "find_paths", "all_paths_for_subgraph", "get_anchors_of_interest",
                     "extract_elements", "find_paths_between_nodes", "insert_valid_paths", "get_gene_to_node_mapping",
                     "get_new_genes_from_alignment", "all_sublist_combinations",
                     "mp_get_all_paths_between_junctions_in_component", "find_potential_paths", "reverse_complement",
                     "merge_dict", "new_get_minhashes_for_paths", "find", "union", "cluster_paths", "assess_connectivity",
                     "merge_read_clusters", "new_merge_clusters", "make_intersection_matrix",
                     "get_node_with_highest_subthreshold_connections", "filter_nodes_by_intersection", "trim_fringe_nodes"},
    "graph_utils": {"plot_node_coverages"},
    "path_finding_utils": {"orient_nodes_on_read", "get_start_stop_indices", "get_unique_anchor_suffixes",
                           "filter_anchor_suffixes"},
}


def test_python_api_surface_covers_the_reference():
    """every method of the reference's GeneMerGraph / Node / Edge / Gene / GeneMer / Read and every function of its
    graph_utils / path_finding_utils (names indexed in tests/golden/api_surface.json by gen_api_surface.py) exists in
    the product under the same name, except the ones the pipeline never reaches"""
    import json
    import os
    import amira_amd
    import amira_amd.graph_utils as gu
    import amira_amd.path_finding_utils as pf
    surface = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "api_surface.json")))
    for cls, names in surface["classes"].items():
        have = set(dir(getattr(amira_amd, cls)))
        missing = [n for n in names if n not in have and n not in UNREACHED.get(cls, set())]
        assert not missing, (cls, missing)
    for mod, obj in (("graph_utils", gu), ("path_finding_utils", pf)):
        missing = [n for n in surface["functions"][mod] if not hasattr(obj, n) and n not in UNREACHED[mod]]
        assert not missing, (mod, missing)
