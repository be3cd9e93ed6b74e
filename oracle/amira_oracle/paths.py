"""Oracle path-finding helpers.  TEST INFRASTRUCTURE ONLY (oracle/README.md).

Restates amira/path_finding_utils.py:4-310 (reference v0.11.0).  The reference
delegates substring search to the third-party package suffix-tree==0.1.2
(pyproject.toml:29, not vendored, not installable here).  Only two calls are used:
``Tree(dict_of_sequences)`` and ``Tree.find_all(seq) -> [(id, path)]`` where
``str(path)`` is the space-joined suffix from the match start plus the end marker
"$" (call sites path_finding_utils.py:85,90,107,244,290,300).  ``Tree`` below
restates that published behaviour by brute force; the reference's own tests
(tests/test_path_finding.py) pin the call sites.
"""


class Tree:
    """Generalised suffix tree stand-in: exact sub-list search over stored sequences."""

    def __init__(self, data=None):
        self._seqs = {k: list(v) for k, v in (data or {}).items()}

    def find_all(self, query):
        q = list(query)
        n = len(q)
        hits = []
        for key, seq in self._seqs.items():
            for i in range(len(seq) - n + 1):
                if seq[i : i + n] == q:
                    hits.append((key, seq[i:]))
        return hits


def _ints(suffix):
    # the reference parses str(path).split(" "), drops "$" and int()s the rest; a leading
    # "None" (masked node) makes it skip the hit (path_finding_utils.py:91-94,108-111)
    if suffix[0] is None:
        return None
    return [int(x) for x in suffix]


def is_sublist(long_list, sub_list):
    # :4-8
    assert isinstance(long_list, list) and isinstance(sub_list, list)
    n = len(sub_list)
    return any(sub_list == long_list[i : i + n] for i in range(len(long_list) - n + 1))


def find_sublist_indices(main, sub):
    # :67-76
    n = len(sub)
    return [(i, i + n - 1) for i in range(len(main) - n + 1) if main[i : i + n] == sub]


def _cluster(adjacent, compatible):
    # :11-64 — longest first; join iff compatible with exactly one existing cluster
    clusters = {}
    for p in sorted(list(adjacent), key=len, reverse=True):
        lp = list(p)
        fits = [c for c in clusters if (not lp) or compatible(lp, list(c))]
        if len(fits) == 0:
            clusters[p] = {p}
        if len(fits) == 1:
            clusters[fits[0]].add(p)
    out = {}
    for c, members in clusters.items():
        ms = list(members)
        out[min(ms, key=len)] = {"longest": max(ms, key=len), "all": ms}
    return out


def cluster_downstream_adjacent_paths(adjacent):
    return _cluster(adjacent, lambda p, c: p == c[: len(p)])


def cluster_upstream_adjacent_paths(adjacent):
    return _cluster(adjacent, lambda p, c: p == c[-len(p) :])


def construct_suffix_tree(read_nodes):
    # :79-85 — mutates the dict it is given
    extra = {}
    for r, nodes in read_nodes.items():
        if len(set(nodes)) != 1:
            extra[r + "_reverse"] = list(reversed(nodes))
    read_nodes.update(extra)
    return Tree(read_nodes)


def get_suffixes_from_initial_tree(tree, a1):
    # :88-100 — longest suffix per id
    out = {}
    for rid, suffix in tree.find_all([a1]):
        nodes = _ints(suffix)
        if nodes is None:
            continue
        if rid not in out or len(nodes) > len(out[rid]):
            out[rid] = nodes
    return out


def get_blocks_from_subtree(sub_tree, a2, anchors):
    # :103-123
    block_reads, dup = {}, {}
    for rid, suffix in sub_tree.find_all([a2]):
        nodes = _ints(suffix)
        if nodes is None:
            continue
        assert nodes[0] in anchors and nodes[-1] in anchors
        dup[tuple(min(nodes, list(reversed(nodes))))] = False
        if "_reverse" not in rid:
            if rid not in block_reads or len(nodes) > len(block_reads[rid]):
                block_reads[rid] = list(reversed(nodes))
    return block_reads, dup


def get_all_context_options(nodes, start, end):
    # :126-141
    up, down = nodes[:start], nodes[end + 1 :]
    ups = {tuple(up[-i:]) for i in range(1, len(up) + 1)}
    downs = {tuple(down[:i]) for i in range(1, len(down) + 1)}
    ups.add(())
    downs.add(())
    return ups, downs


def get_canonical_representation(block):
    return sorted([block, list(reversed(block))])[0]


def get_full_path_contexts(positions, contexts, reads, rid, block_reads):
    # :144-162 — reversed blocks OVERWRITE the entry (reference behaviour)
    start, end = positions[0]
    ups, downs = get_all_context_options(reads[rid], start, end)
    canon = get_canonical_representation(block_reads[rid])
    key = tuple(canon)
    if key not in contexts:
        contexts[key] = {"upstream": set(), "downstream": set()}
    if canon == block_reads[rid]:
        contexts[key]["upstream"].update(ups)
        contexts[key]["downstream"].update(downs)
    else:
        contexts[key] = {
            "upstream": {tuple(reversed(d)) for d in downs},
            "downstream": {tuple(reversed(u)) for u in ups},
        }


def generate_contexts(block_reads, dup, reads):
    # :204-215
    contexts = {}
    for rid, block in block_reads.items():
        where = find_sublist_indices(reads[rid], block)
        assert len(where) > 0
        if len(where) > 1:
            dup[tuple(get_canonical_representation(block))] = True
        if len(where) == 1:
            get_full_path_contexts(where, contexts, reads, rid, block_reads)
    return contexts


def update_full_blocks(full_paths, tree, threshold, full_blocks, c):
    # :241-247
    for f in full_paths:
        support = {rid.replace("_reverse", "") for rid, _ in tree.find_all(f)}
        if len(support) > 0:
            full_blocks[tuple(f)] = support


def generate_full_paths(contexts, dup, full_blocks, tree, threshold):
    # :227-238
    for c in contexts:
        if dup[c] is False:
            ups = cluster_upstream_adjacent_paths(contexts[c]["upstream"])
            downs = cluster_downstream_adjacent_paths(contexts[c]["downstream"])
            update_full_blocks(
                [u + c + d for u in ups for d in downs], tree, threshold, full_blocks, c
            )


def process_anchors(sub_tree, anchors, a1, full_blocks, reads, tree, threshold):
    # :196-201
    for a2 in anchors:
        if a1 != a2:
            block_reads, dup = get_blocks_from_subtree(sub_tree, a2, anchors)
            contexts = generate_contexts(block_reads, dup, reads)
            generate_full_paths(contexts, dup, full_blocks, tree, threshold)


def filter_blocks(full_blocks):
    # :250-259
    kept = {}
    for p in sorted(list(full_blocks.keys()), key=len, reverse=True):
        fwd, rev = list(p), list(reversed(p))
        if not any(is_sublist(list(f), fwd) or is_sublist(list(f), rev) for f in kept):
            kept[p] = full_blocks[p]
    return kept


def get_reads_supporting_path(path, gene_tree):
    # :288-293
    return {rid.replace("_reverse", "") for rid, _ in gene_tree.find_all(list(path))}


def process_combinations_for_i(args):
    # :296-310
    i, threshold, gene, lst, gene_call_subset = args
    gene_tree = Tree(gene_call_subset)
    out = {}
    want = lst.count(f"+{gene}") + lst.count(f"-{gene}")
    for s in range(len(lst) - i + 1):
        comb = tuple(lst[s : s + i])
        if comb.count(f"+{gene}") + comb.count(f"-{gene}") == want:
            support = get_reads_supporting_path(comb, gene_tree)
            if len(support) >= threshold:
                out[comb] = len(support)
    return out
