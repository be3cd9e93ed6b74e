"""Read-path clustering helpers — drop-in for amira/path_finding_utils.py (reference v0.11.0).

Same function names and argument meaning as the reference.  Sub-list search goes through
``Tree`` below: the reference uses the PyPI package suffix-tree==0.1.2 only for
``Tree(dict)`` / ``Tree.find_all(seq)``; when that package is importable it is used,
otherwise an exact scan with the same results (ids + suffix from the match start).
"""
try:  # pragma: no cover - the package is not installed in the build image
    from suffix_tree import Tree as _ExternalTree
except Exception:  # noqa: BLE001
    _ExternalTree = None


class _Suffix:
    """What Tree.find_all hands back per hit: str() is the space-joined suffix + ' $'."""

    __slots__ = ("items",)

    def __init__(self, items):
        self.items = items

    def __str__(self):
        return " ".join([str(x) for x in self.items] + ["$"])


class _ReversedSuffixes:
    """sequence j = parent[rows[j]][first[j]:] back to front, made when asked for; suffix(j, i) = that sequence from
    its item i on, as ONE slice of the parent's list"""

    __slots__ = ("_parent", "_rows", "_first")

    def __init__(self, parent, rows, first):
        self._parent, self._rows, self._first = parent, rows, first

    def __len__(self):
        return len(self._rows)

    def suffix(self, j, i):
        seq, p = self._parent[self._rows[j]], self._first[j]
        return seq[len(seq) - 1 - i:p - 1:-1] if p else seq[len(seq) - 1 - i::-1]

    def __getitem__(self, j):
        return self.suffix(j, 0)


class _ScanTree:
    """Exact sub-list search over a {key: sequence} mapping with the call shape of suffix_tree.Tree
    (`Tree(dict)`, `find_all(seq) -> [(key, suffix from the match start)]`, hits in key order then
    position).  Items are interned to small integers and all sequences laid end to end in one numpy
    array, so a query is a handful of vectorised comparisons instead of a Python loop over every
    read (the read-path clustering of a million-read graph issues hundreds of queries over tens of
    thousands of node lists)."""

    def __init__(self, data=None):
        import numpy as np
        from itertools import chain
        self._np = np
        self._keys = list((data or {}).keys())
        self._seqs = [list(seq) for seq in (data or {}).values()]
        starts = np.zeros(len(self._seqs) + 1, dtype=np.int64)
        np.cumsum([len(seq) for seq in self._seqs], out=starts[1:])
        self._starts = starts
        try:
            # items that are machine integers already (node ids) need no interning: one C-level pass
            self._flat = np.fromiter(chain.from_iterable(self._seqs), dtype=np.int64, count=int(starts[-1]))
            self._code = None
        except (TypeError, ValueError, OverflowError):
            code = self._code = {}   # 256-bit node hashes, None for masked nodes, strings, ...
            flat = []
            for seq in self._seqs:
                flat.extend(code.setdefault(x, len(code)) for x in seq)
            self._flat = np.asarray(flat, dtype=np.int64)

    @classmethod
    def from_codes(cls, keys, seqs, code_arrays, code_of):
        """sequences whose integer codes are known already (device node ids of the per-window array):
        `seqs` are the item lists themselves (what suffixes are cut from), `code_arrays` their codes as
        numpy arrays, `code_of(item)` the code of a query item (None: no sequence holds it)"""
        import numpy as np
        self = cls.__new__(cls)
        self._np = np
        self._keys, self._seqs = list(keys), list(seqs)
        starts = np.zeros(len(code_arrays) + 1, dtype=np.int64)
        np.cumsum([len(a) for a in code_arrays], out=starts[1:])
        self._starts = starts
        self._flat = (np.concatenate(code_arrays).astype(np.int64) if len(code_arrays) else np.zeros(0, np.int64))
        self._code = code_of
        return self

    @classmethod
    def from_flat(cls, keys, seqs, flat, starts, code_of):
        """as from_codes, with the codes already laid end to end (`flat`, `starts`) and `seqs` anything that
        answers seqs[row] with the row's item list (it may build the list when asked)"""
        import numpy as np
        self = cls.__new__(cls)
        self._np = np
        self._keys, self._seqs = list(keys), seqs
        self._starts = np.asarray(starts, dtype=np.int64)
        self._flat = np.asarray(flat)   # any integer type (the device's 32-bit ids: half the bytes per scan)
        self._code = code_of
        return self

    def _codes_of(self, query):
        np = self._np
        if self._code is None:
            if not all(isinstance(x, (int, np.integer)) and not isinstance(x, bool) for x in query):
                return None
            return query
        lookup = self._code if callable(self._code) else self._code.get
        codes = [lookup(x) for x in query]
        return None if None in codes else codes

    def reversed_suffix_tree(self, item):
        """get_suffixes_from_initial_tree(self, item) followed by Tree({key: reversed(suffix)}) (the two
        steps of get_full_paths, path_finding_utils.py:88-100 + construct_graph.py:2732-2735) without
        re-interning a single item: per key the LONGEST suffix that starts at an occurrence of `item`
        (= its first occurrence), reversed; codes are cut out of this tree's own code array."""
        np = self._np
        codes = self._codes_of([item])
        sub = type(self).__new__(type(self))
        sub._np, sub._code = np, self._code
        if codes is None or len(self._flat) == 0:
            sub._keys, sub._seqs = [], []
            sub._starts, sub._flat = np.zeros(1, np.int64), np.zeros(0, self._flat.dtype)
            return sub
        at = np.flatnonzero(self._flat == codes[0])
        row = np.searchsorted(self._starts, at, side="right") - 1
        first = np.concatenate([[True], row[1:] != row[:-1]]) if len(row) else np.zeros(0, bool)
        at, row = at[first], row[first]
        ends = self._starts[row + 1]
        sub._keys = [self._keys[r] for r in row.tolist()]
        # the item lists of the sub-tree are cut out of this tree's lists when a hit asks for them
        sub._seqs = _ReversedSuffixes(self._seqs, row.tolist(), (at - self._starts[row]).tolist())
        lens = ends - at
        starts = np.zeros(len(at) + 1, dtype=np.int64)
        np.cumsum(lens, out=starts[1:])
        sub._starts = starts
        # codes of all suffixes back to front in one gather
        within = np.arange(int(starts[-1]), dtype=np.int64) - np.repeat(starts[:-1], lens)
        sub._flat = self._flat[np.repeat(ends - 1, lens) - within]
        return sub

    def _hits(self, query):
        """(positions in the code array, rows) of the occurrences of `query`, in key order then position"""
        np = self._np
        m = len(query)
        if m == 0 or len(self._flat) == 0:
            return None
        codes = self._codes_of(query)
        if codes is None:   # an item no sequence holds
            return None
        flat, starts = self._flat, self._starts
        at = np.flatnonzero(flat == codes[0])
        if len(at) == 0:
            return None
        row = np.searchsorted(starts, at, side="right") - 1
        ok = at + m <= starts[row + 1]          # the match must end inside its own sequence
        at, row = at[ok], row[ok]
        for j in range(1, m):
            if len(at) == 0:
                return None
            keep = flat[at + j] == codes[j]
            at, row = at[keep], row[keep]
        return at, row

    def find_all(self, query):
        hits = self._hits(list(query))
        if hits is None:
            return []
        at, row = hits
        begin = (at - self._starts[row]).tolist()
        keys, seqs = self._keys, self._seqs
        if isinstance(seqs, _ReversedSuffixes):
            cut = seqs.suffix
            return [(keys[r], _Suffix(cut(r, i))) for r, i in zip(row.tolist(), begin)]
        return [(keys[r], _Suffix(seqs[r][i:])) for r, i in zip(row.tolist(), begin)]

    def find_all_coded(self, query):
        """find_all for callers that see the same suffix over and over (thousands of reads run through one block):
        (keys, codes, items_of) — per hit its key and the integer codes of its suffix as bytes (equal codes <=> equal
        items), and items_of(j), which cuts the item list of hit j only when asked"""
        hits = self._hits(list(query))
        if hits is None:
            return [], [], None
        at, row = hits
        starts, flat = self._starts, self._flat
        rows = row.tolist()
        begin = (at - starts[row]).tolist()
        codes = [flat[a:e].tobytes() for a, e in zip(at.tolist(), starts[row + 1].tolist())]
        keys, seqs = self._keys, self._seqs
        if isinstance(seqs, _ReversedSuffixes):
            cut = seqs.suffix
            items_of = lambda j: cut(rows[j], begin[j])   # noqa: E731
        else:
            items_of = lambda j: seqs[rows[j]][begin[j]:]   # noqa: E731
        return [keys[r] for r in rows], codes, items_of


Tree = _ExternalTree or _ScanTree


def _suffix_ints(path):
    if isinstance(path, _Suffix):  # our own tree: no need to print and re-parse the suffix
        return None if path.items[0] is None else path.items   # (a list of its own: find_all cuts one per hit)
    parts = str(path).split(" ")
    if parts[0] == "None":
        return None
    return [int(n) for n in parts if n != "$"]


def is_sublist(long_list, sub_list):
    assert isinstance(long_list, list) and isinstance(sub_list, list)
    m = len(sub_list)
    for i in range(len(long_list) - m + 1):
        if long_list[i:i + m] == sub_list:
            return True
    return False


def find_sublist_indices(main_list, sublist):
    m = len(sublist)
    if m == 0:
        return [(i, i - 1) for i in range(len(main_list) + 1)]
    # candidates = where the first item occurs (list.index scans at C speed), then one slice compare each
    found, first, i, last = [], sublist[0], -1, len(main_list) - m
    try:
        while True:
            i = main_list.index(first, i + 1)
            if i > last:
                break
            if main_list[i:i + m] == sublist:
                found.append((i, i + m - 1))
    except ValueError:
        pass
    return found


def _greedy_clusters(adjacent_paths, fits):
    ordered = sorted([p for p in adjacent_paths], key=len, reverse=True)
    groups = {}
    for p in ordered:
        as_list = list(p)
        matches = [rep for rep in groups if (not as_list) or fits(as_list, list(rep))]
        if not matches:
            groups[p] = {p}
        if len(matches) == 1:
            groups[matches[0]].add(p)
    result = {}
    for rep in groups:
        members = list(groups[rep])
        result[min(members, key=len)] = {"longest": max(members, key=len), "all": members}
    return result


def cluster_downstream_adjacent_paths(adjacent_paths):
    """prefix-compatible clustering, longest first (path_finding_utils.py:11-36)."""
    return _greedy_clusters(adjacent_paths, lambda p, c: p == c[:len(p)])


def cluster_upstream_adjacent_paths(adjacent_paths):
    """suffix-compatible clustering, longest first (:39-64)."""
    return _greedy_clusters(adjacent_paths, lambda p, c: p == c[-len(p):])


def construct_suffix_tree(read_nodes):
    """adds '<read>_reverse' entries for reads with >1 distinct node — mutates its argument (:79-85)."""
    flipped = {r + "_reverse": list(reversed(nodes)) for r, nodes in read_nodes.items()
               if len(set(nodes)) != 1}
    read_nodes.update(flipped)
    return Tree(read_nodes)


def get_suffixes_from_initial_tree(tree, a1):
    best = {}
    for read_id, path in tree.find_all([a1]):
        nodes = _suffix_ints(path)
        if nodes is None:
            continue
        if read_id not in best or len(nodes) > len(best[read_id]):
            best[read_id] = nodes
    return best


def get_blocks_from_subtree(sub_tree, a2, nodeAnchors):
    block_reads, block_duplicates = {}, {}
    coded = getattr(sub_tree, "find_all_coded", None)
    if coded is not None:
        # the same block comes from thousands of reads: its list, its reversal and its canonical tuple are made for
        # the first of them and SHARED by the rest (nobody writes to a block list) — which assignment happens for
        # which read, and in which order the dicts are filled, stays as in the loop below
        keys, codes, items_of = coded([a2])
        seen = {}   # codes of a suffix -> its block (the suffix back to front), None: the suffix starts with None
        for j, read_id in enumerate(keys):
            cb = codes[j]
            if cb in seen:
                block = seen[cb]
            else:
                nodes = items_of(j)
                if nodes[0] is None:
                    block = seen[cb] = None
                else:
                    assert nodes[0] in nodeAnchors and nodes[-1] in nodeAnchors
                    block = seen[cb] = nodes[::-1]
                    block_duplicates[tuple(sorted([nodes, block])[0])] = False
            if block is None:
                continue
            if "_reverse" not in read_id:
                if read_id not in block_reads or len(block) > len(block_reads[read_id]):
                    block_reads[read_id] = block
        return block_reads, block_duplicates
    for read_id, path in sub_tree.find_all([a2]):
        nodes = _suffix_ints(path)
        if nodes is None:
            continue
        assert nodes[0] in nodeAnchors and nodes[-1] in nodeAnchors
        block_duplicates[tuple(sorted([nodes, nodes[::-1]])[0])] = False
        if "_reverse" not in read_id:
            if read_id not in block_reads or len(nodes) > len(block_reads[read_id]):
                block_reads[read_id] = nodes[::-1]
    return block_reads, block_duplicates


def get_all_context_options(nodes_on_reads, start, end):
    up, down = tuple(nodes_on_reads[:start]), tuple(nodes_on_reads[end + 1:])
    # (slices of a tuple are tuples; the elements go in one by one in the reference's order — suffixes by growing
    # length, then the empty tuple — so that the sets iterate as the reference's do: a union with {()} would
    # rebuild the table)
    up_options = {up[-i:] for i in range(1, len(up) + 1)}
    up_options.add(())
    down_options = {down[:i] for i in range(1, len(down) + 1)}
    down_options.add(())
    return up_options, down_options


def get_canonical_representation(block_read):
    return sorted([block_read, list(reversed(block_read))])[0]


def update_duplicates(block_duplicates, canonical_tuple, positions_of_path):
    if len(positions_of_path) > 1:
        block_duplicates[canonical_tuple] = True
    return block_duplicates


def get_full_path_contexts(positions_of_path, contexts, reads, read_id, block_reads):
    start, end = positions_of_path[0]
    up_options, down_options = get_all_context_options(reads[read_id], start, end)
    canonical = get_canonical_representation(block_reads[read_id])
    key = tuple(canonical)
    contexts.setdefault(key, {"upstream": set(), "downstream": set()})
    if canonical == block_reads[read_id]:
        contexts[key]["upstream"].update(up_options)
        contexts[key]["downstream"].update(down_options)
    else:  # the reference REPLACES the entry here
        contexts[key] = {"upstream": {tuple(reversed(d)) for d in down_options},
                         "downstream": {tuple(reversed(u)) for u in up_options}}


def generate_contexts(block_reads, block_duplicates, reads):
    """contexts of every block (path_finding_utils.py:150-215), the reference's read-by-read loop with two
    savings that leave every set operation it performs on the surviving entries exactly as it is:
      * a read whose block is in the NON-canonical orientation replaces the block's entry by sets made from
        that read alone, so whatever earlier reads contributed is gone — only the LAST such read of a block
        and the canonical reads after it are looked at;
      * the O(L^2) option sets of a read depend on its upstream / downstream node lists only, and the reads
        through one block share a few dozen of those: the sets are made once per distinct list and handed to
        `set.update` again (same contents, same size, same iteration order as a freshly made one — `update`
        resizes by the size of its argument even when every element is present already, so the calls
        themselves cannot be skipped without changing how the context sets iterate, which decides ties in the
        clustering that follows)."""
    contexts = {}
    todo = {}  # key -> [(read_id, where, is_canonical), ...] in read order
    per_block = {}   # id(block list) -> [key, is_canonical, its todo list]: the reads through one block share ONE list
    for read_id, block in block_reads.items():   # (get_blocks_from_subtree), so canonical form and key are made once
        on_read = reads[read_id]
        where = find_sublist_indices(on_read, block)
        assert len(where) > 0
        rec = per_block.get(id(block))
        if rec is None:
            canonical = get_canonical_representation(block)
            rec = per_block[id(block)] = [tuple(canonical), canonical == block, None]
        key = rec[0]
        update_duplicates(block_duplicates, key, where)
        if len(where) == 1:
            if rec[2] is None:
                contexts.setdefault(key, {"upstream": set(), "downstream": set()})   # dict order = first read
                rec[2] = todo.setdefault(key, [])
            rec[2].append((read_id, where, rec[1]))
    up_cache, down_cache = {}, {}
    for key, entries in todo.items():
        first = 0
        for i, (_, _, is_canonical) in enumerate(entries):
            if not is_canonical:
                first = i
        for read_id, where, is_canonical in entries[first:]:
            start, end = where[0]
            on_read = reads[read_id]
            up, down = tuple(on_read[:start]), tuple(on_read[end + 1:])
            up_options = up_cache.get(up)
            if up_options is None:
                up_options = up_cache[up] = {up[-i:] for i in range(1, len(up) + 1)}
                up_options.add(())
            down_options = down_cache.get(down)
            if down_options is None:
                down_options = down_cache[down] = {down[:i] for i in range(1, len(down) + 1)}
                down_options.add(())
            if is_canonical:
                contexts[key]["upstream"].update(up_options)
                contexts[key]["downstream"].update(down_options)
            else:  # the reference REPLACES the entry here
                contexts[key] = {"upstream": {tuple(reversed(d)) for d in down_options},
                                 "downstream": {tuple(reversed(u)) for u in up_options}}
    return contexts


def build_full_paths(upstream_clusters, downstream_clusters, c):
    return [u + c + d for u in upstream_clusters for d in downstream_clusters]


def update_full_blocks(full_paths, tree, threshold, full_blocks, c):
    for f in full_paths:
        support = {read_id.replace("_reverse", "") for read_id, _ in tree.find_all(f)}
        if support:
            full_blocks[tuple(f)] = support


def generate_full_paths(contexts, block_duplicates, full_blocks, tree, threshold):
    for c in contexts:
        if block_duplicates[c] is False:
            ups = cluster_upstream_adjacent_paths(contexts[c]["upstream"])
            downs = cluster_downstream_adjacent_paths(contexts[c]["downstream"])
            update_full_blocks(build_full_paths(ups, downs, c), tree, threshold, full_blocks, c)


def process_anchors(sub_tree, nodeAnchors, a1, full_blocks, reads, tree, threshold):
    for a2 in nodeAnchors:
        if a1 != a2:
            block_reads, block_duplicates = get_blocks_from_subtree(sub_tree, a2, nodeAnchors)
            contexts = generate_contexts(block_reads, block_duplicates, reads)
            generate_full_paths(contexts, block_duplicates, full_blocks, tree, threshold)


def filter_blocks(full_blocks):
    kept = {}
    for p in sorted(list(full_blocks.keys()), key=len, reverse=True):
        fwd = list(p)
        rev = fwd[::-1]
        if not any(is_sublist(list(f), fwd) or is_sublist(list(f), rev) for f in kept):
            kept[p] = full_blocks[p]
    return kept


def get_reads_supporting_path(path, gene_tree):
    return {read_id.replace("_reverse", "") for read_id, _ in gene_tree.find_all(list(path))}


def process_combinations_for_i(args):
    """every window of length i that holds all copies of the gene, with its read support (:296-310)."""
    i, threshold, geneOfInterest, lst, gene_call_subset = args
    gene_tree = Tree(gene_call_subset)
    plus, minus = f"+{geneOfInterest}", f"-{geneOfInterest}"
    wanted = lst.count(plus) + lst.count(minus)
    found = {}
    for start in range(len(lst) - i + 1):
        comb = tuple(lst[start:start + i])
        if comb.count(plus) + comb.count(minus) == wanted:
            n = len(get_reads_supporting_path(comb, gene_tree))
            if n >= threshold:
                found[comb] = n
    return found
