"""Canonical dumps of a gene-mer graph, shared by the golden generator and the tests.

Works on anything exposing the reference's accessor API (the reference itself, the
oracle, the HIP-backed product).  Order is preserved wherever the reference's order
is observable (dict insertion order of nodes / edges / reads, per-node lists).
"""
import hashlib
import json
import lzma
import os

HERE = os.path.dirname(os.path.abspath(__file__))


def load_fixture(name):
    """tests/golden/data/<name>.json.xz -> object (copies of the reference's own test data)."""
    with lzma.open(os.path.join(HERE, "data", name + ".json.xz"), "rt") as fh:
        return json.load(fh)


def digest(obj):
    return hashlib.sha256(json.dumps(obj, separators=(",", ":")).encode()).hexdigest()


def _edge_desc(g, e):
    return [
        g.get_gene_mer_label(e.get_sourceNode()),
        g.get_gene_mer_label(e.get_targetNode()),
        e.get_sourceNodeDirection(),
        e.get_targetNodeDirection(),
        e.get_edge_coverage(),
    ]


def dump_nodes(g, with_hash=False):
    out = []
    for h, n in g.get_nodes().items():
        row = [
            g.get_gene_mer_label(n),
            n.get_node_coverage(),
            n.get_component(),
            n.get_geneMer().get_geneMerDirection(),
            list(n.get_list_of_reads()),
            [_edge_desc(g, g.get_edges()[eh]) for eh in n.get_forward_edge_hashes()],
            [_edge_desc(g, g.get_edges()[eh]) for eh in n.get_backward_edge_hashes()],
        ]
        if with_hash:
            row.append(str(h))
        out.append(row)
    return out


def dump_edges(g, with_hash=False):
    out = []
    for h, e in g.get_edges().items():
        row = _edge_desc(g, e)
        if with_hash:
            row.append(str(h))
        out.append(row)
    return out


def dump_read_nodes(g):
    labels = {h: g.get_gene_mer_label(n) for h, n in g.get_nodes().items()}
    rn, rd, rp = g.get_readNodes(), g.get_readNodeDirections(), g.get_readNodePositions()
    out = []
    for rid in rn:
        out.append(
            [
                rid,
                # a hash that is no longer in the graph keeps its place as "?" (never
                # happens on the paths we dump; None marks a masked node)
                [None if h is None else labels.get(h, "?") for h in rn[rid]],
                list(rd[rid]),
                [None if p is None else list(p) for p in rp[rid]],
            ]
        )
    return out


def dump_graph(g, with_hash=False):
    return {
        "nodes": dump_nodes(g, with_hash),
        "edges": dump_edges(g, with_hash),
        "read_nodes": dump_read_nodes(g),
        "short_reads": [[r, list(v)] for r, v in g.get_short_read_annotations().items()],
        "to_correct": sorted(g.get_reads_to_correct()),
    }


def summarise(d, sample=20):
    """Digest form of dump_graph(): what is committed in goldens.json."""
    return {
        "n_nodes": len(d["nodes"]),
        "n_edges": len(d["edges"]),
        "n_reads_with_nodes": len(d["read_nodes"]),
        "n_short": len(d["short_reads"]),
        "n_to_correct": len(d["to_correct"]),
        "sum_node_cov": sum(r[1] for r in d["nodes"]),
        "sum_edge_cov": sum(r[4] for r in d["edges"]),
        "n_components": len({r[2] for r in d["nodes"]}),
        "nodes_digest": digest(d["nodes"]),
        "edges_digest": digest(d["edges"]),
        "read_nodes_digest": digest(d["read_nodes"]),
        "short_digest": digest(d["short_reads"]),
        "to_correct_digest": digest(d["to_correct"]),
        "first_nodes": [r[:4] + [len(r[4]), len(r[5]), len(r[6])] for r in d["nodes"][:sample]],
        "first_edges": d["edges"][:sample],
    }


def dump_corrected(genes, positions):
    return {
        "genes": [[r, list(v)] for r, v in genes.items()],
        "positions": [[r, [list(p) for p in v]] for r, v in positions.items()],
    }


def summarise_corrected(d):
    return {
        "n_reads": len(d["genes"]),
        "n_genes": sum(len(v) for _, v in d["genes"]),
        "genes_digest": digest(d["genes"]),
        "positions_digest": digest(d["positions"]),
    }


def canon_clusters(clustered, path_reads):
    """Order-independent form of assign_reads_to_genes() output."""
    c = []
    for comp in sorted(clustered, key=lambda x: (x is None, x)):
        for gene in sorted(clustered[comp]):
            for allele in sorted(clustered[comp][gene]):
                c.append([comp, gene, allele, sorted(clustered[comp][gene][allele])])
    p = sorted([list(k), sorted(v)] for k, v in path_reads.items())
    return {"clusters": c, "path_reads": p}


def anon_clusters(clustered, path_reads, genes):
    """Allele-numbering-independent form: the reference numbers alleles in an order that
    leaks set-of-str iteration order (PYTHONHASHSEED), the partition itself does not."""
    import re

    c = []
    for comp in sorted(clustered, key=lambda x: (x is None, x)):
        for gene in sorted(clustered[comp]):
            c.append([comp, gene, sorted(sorted(v) for v in clustered[comp][gene].values())])
    pat = re.compile(r"^([+-](?:%s))_\d+$" % "|".join(re.escape(g) for g in genes))
    p = sorted(
        [[pat.sub(r"\1", x) for x in k], sorted(v)] for k, v in path_reads.items()
    )
    return {"clusters": c, "path_reads": p}
