"""Test helpers: bring oracle graphs and engine arrays to one comparable form."""
import numpy as np


def oracle_arrays(g, vocab, read_ids, read_off, k):
    """Array view of an oracle (or reference-API) graph in the engine's conventions."""
    node_id = {}
    tokens, cov, first_dir, comp = [], [], [], []
    for h, n in g.get_nodes().items():
        node_id[h] = len(node_id)
        tokens.append([vocab.token(("+" if x.get_strand() == 1 else "-") + x.get_name())
                       for x in n.get_canonical_geneMer()])
        cov.append(n.get_node_coverage())
        first_dir.append(n.get_geneMer().get_geneMerDirection())
        comp.append(n.get_component())
    edge_id = {}
    src, tgt, sdir, tdir, ecov = [], [], [], [], []
    for h, e in g.get_edges().items():
        edge_id[h] = len(edge_id)
        src.append(node_id[e.get_sourceNode().__hash__()])
        tgt.append(node_id[e.get_targetNode().__hash__()])
        sdir.append(e.get_sourceNodeDirection())
        tdir.append(e.get_targetNodeDirection())
        ecov.append(e.get_edge_coverage())
    T = int(read_off[-1])
    tok_node = np.full(T, -1, np.int32)
    tok_dir = np.zeros(T, np.int8)
    rn, rd = g.get_readNodes(), g.get_readNodeDirections()
    for r, rid in enumerate(read_ids):
        if rid not in rn:
            continue
        a = int(read_off[r])
        for i, h in enumerate(rn[rid]):
            if h is None:
                tok_node[a + i] = -2
            else:
                tok_node[a + i] = node_id[h]
                tok_dir[a + i] = rd[rid][i]
    adj = []
    for h, n in g.get_nodes().items():
        adj.append([edge_id[x] for x in n.get_forward_edge_hashes()])
        adj.append([edge_id[x] for x in n.get_backward_edge_hashes()])
    ridx = {rid: i for i, rid in enumerate(read_ids)}
    node_reads = [[ridx[r] for r in n.get_list_of_reads()] for n in g.get_nodes().values()]
    return {
        "tokens": np.asarray(tokens, np.int32).reshape(len(tokens), k),
        "coverage": np.asarray(cov, np.uint32), "first_dir": np.asarray(first_dir, np.int8),
        "component": np.asarray(comp, np.int32),
        "src": np.asarray(src, np.int32), "tgt": np.asarray(tgt, np.int32),
        "sdir": np.asarray(sdir, np.int8), "tdir": np.asarray(tdir, np.int8),
        "ecov": np.asarray(ecov, np.uint32),
        "tok_node": tok_node, "tok_dir": tok_dir, "adj": adj, "node_reads": node_reads,
        "short": [r for r in read_ids if r in g.get_short_read_annotations()],
        "to_correct": sorted(g.get_reads_to_correct()),
    }


def csr_lists(off, vals, alive=None):
    out = []
    for i in range(len(off) - 1):
        row = vals[off[i]:off[i + 1]]
        if alive is not None:
            row = row[alive[row] != 0]
        out.append(row.tolist())
    return out


def compare_engine_to_oracle(eng, want, live_only=False):
    """Assert that the engine's current graph equals the oracle arrays `want`.
    With live_only, dead nodes / edges of the engine are dropped and ids renumbered
    (the oracle deletes them from its dicts)."""
    c = eng.counts()
    nodes, edges = eng.nodes(), eng.edges()
    tok_node, tok_dir = eng.read_nodes()
    off, adj = eng.node_adj()
    if not live_only:
        assert c["n_nodes"] == len(want["coverage"]) and c["n_edges"] == len(want["src"])
        nmap = np.arange(c["n_nodes"])
        emap = np.arange(c["n_edges"])
        nkeep = np.ones(c["n_nodes"], bool)
        ekeep = np.ones(c["n_edges"], bool)
    else:
        nkeep, ekeep = nodes["alive"] != 0, edges["alive"] != 0
        nmap = np.cumsum(nkeep) - 1
        emap = np.cumsum(ekeep) - 1
        assert int(nkeep.sum()) == len(want["coverage"]), (int(nkeep.sum()), len(want["coverage"]))
        assert int(ekeep.sum()) == len(want["src"]), (int(ekeep.sum()), len(want["src"]))
    assert np.array_equal(nodes["tokens"][nkeep], want["tokens"])
    assert np.array_equal(nodes["coverage"][nkeep], want["coverage"])
    assert np.array_equal(nodes["first_dir"][nkeep], want["first_dir"])
    assert np.array_equal(nodes["component"][nkeep], want["component"])
    assert np.array_equal(nmap[edges["src"][ekeep]], want["src"])
    assert np.array_equal(nmap[edges["tgt"][ekeep]], want["tgt"])
    assert np.array_equal(edges["sdir"][ekeep], want["sdir"])
    assert np.array_equal(edges["tdir"][ekeep], want["tdir"])
    assert np.array_equal(edges["coverage"][ekeep], want["ecov"])
    got_node = tok_node.copy()
    m = tok_node >= 0
    got_node[m] = nmap[tok_node[m]]
    assert np.array_equal(got_node, want["tok_node"])
    assert np.array_equal(np.where(tok_node >= 0, tok_dir, 0), want["tok_dir"])
    rows = csr_lists(off, adj, edges["alive"] if live_only else None)
    got_adj = []
    for n in range(c["n_nodes"]):
        if nkeep[n]:
            got_adj.append([int(emap[e]) for e in rows[2 * n]])
            got_adj.append([int(emap[e]) for e in rows[2 * n + 1]])
    assert got_adj == want["adj"]
    roff, ridx = eng.node_reads()
    got_reads = [r for n, r in enumerate(csr_lists(roff, ridx)) if nkeep[n]]
    assert got_reads == want["node_reads"]


def compare_engine_to_sweep(eng, orc, what=""):
    """Array-level equality of the engine's current state with the C sweep oracle
    (tests/token_oracle.Sweep) — vectorised, for the full benchmark sizes.  Both sides keep removed
    nodes / edges in place with alive = 0, so ids compare directly."""
    ce, co = eng.counts(), orc.counts()
    for key in ("n_reads", "n_tokens", "n_windows", "n_short_reads", "n_nodes", "n_edges", "n_components",
                "n_live_nodes", "n_live_edges", "n_reads_to_correct"):
        assert ce[key] == co[key], (what, key, ce[key], co[key])
    ne, no = eng.nodes(), orc.nodes()
    for key in ("tokens", "coverage", "first_dir", "component", "alive"):
        assert np.array_equal(ne[key], no[key]), (what, "node", key)
    ee, eo = eng.edges(), orc.edges()
    for key in ("src", "tgt", "sdir", "tdir", "coverage", "alive"):
        assert np.array_equal(ee[key], eo[key]), (what, "edge", key)
    tn_e, td_e = eng.read_nodes()
    tn_o, td_o = orc.read_nodes()
    assert np.array_equal(tn_e, tn_o), (what, "tok_node")
    assert np.array_equal(np.where(tn_e >= 0, td_e, 0), np.where(tn_o >= 0, td_o, 0)), (what, "tok_dir")
    off_e, adj_e = eng.node_adj()
    off_o, adj_o = orc.node_adj()
    assert np.array_equal(off_e, off_o) and np.array_equal(adj_e, adj_o), (what, "adjacency")
    assert np.array_equal(eng.reads_to_correct() != 0, orc.reads_to_correct() != 0), (what, "reads to correct")


def compare_corrected(eng, orc, with_pos, what=""):
    """correct_reads on both sides; corrected CSR, origin, changed flags and positions equal"""
    nr_e, nt_e = eng.correct_reads()
    nr_o, nt_o = orc.correct_reads()
    assert (nr_e, nt_e) == (nr_o, nt_o), (what, nr_e, nt_e, nr_o, nt_o)
    a, b = eng.corrected(nr_e, nt_e, with_pos), orc.corrected(nr_o, nt_o, with_pos)
    keys = ["tokens", "read_offsets", "orig_read", "changed"] + (["gene_start", "gene_end"] if with_pos else [])
    for key in keys:
        assert np.array_equal(a[key], b[key]), (what, key)
    return a


def live_arrays(obj, with_adj=True):
    """The graph restricted to live nodes / edges, ids renumbered densely in id order (what the
    reference's dicts hold after removals) — vectorised; obj = Engine or token_oracle.Sweep."""
    n, e = obj.nodes(), obj.edges()
    nk, ek = n["alive"] != 0, e["alive"] != 0
    nmap, emap = np.cumsum(nk) - 1, np.cumsum(ek) - 1
    tok_node, tok_dir = obj.read_nodes()
    mapped = tok_node.copy()
    m = tok_node >= 0
    mapped[m] = nmap[tok_node[m]]
    out = {"tokens": n["tokens"][nk], "coverage": n["coverage"][nk], "first_dir": n["first_dir"][nk],
           "src": nmap[e["src"][ek]], "tgt": nmap[e["tgt"][ek]], "sdir": e["sdir"][ek], "tdir": e["tdir"][ek],
           "ecov": e["coverage"][ek], "tok_node": mapped, "tok_dir": np.where(m, tok_dir, 0),
           "to_correct": obj.reads_to_correct() != 0}
    if with_adj:
        off, adj = obj.node_adj()
        keep = ek[adj]
        csum = np.concatenate([[0], np.cumsum(keep)])
        row_cnt = csum[off[1:]] - csum[off[:-1]]
        rows = np.repeat(nk, 2)
        out["adj_off"] = np.concatenate([[0], np.cumsum(row_cnt[rows])])
        out["adj"] = emap[adj[keep]]
    return out
