"""filter_paths_between_bubble_starts (construct_graph.py:2125-2146) finds the paths that hold path i through the
occurrence lists of their items; the reference asks a suffix tree (Tree(dict).find_all) for them.  Same survivors, same
order, on random path sets with nested, mirrored and repeated pieces."""
import numpy as np
import pytest

from amira_amd.bubble_popping import BubblePopping
from amira_amd.path_finding_utils import Tree


class _Stub(BubblePopping):
    _host_edits = True

    def calculate_path_coverage(self, path):
        return len(path)


def _by_suffix_tree(unique_paths):
    unique_paths = sorted(list(unique_paths), key=len)
    tree = Tree({i: p for i, p in enumerate(unique_paths)})
    kept, targets = [], set()
    for i in range(len(unique_paths)):
        if i in targets:
            continue
        p = unique_paths[i]
        p_list = list(p)
        res = [pid for pid, _ in tree.find_all(p_list)]
        rv_res = [pid for pid, _ in tree.find_all(list(reversed(p_list)))]
        for j in res + rv_res:
            if i != j:
                targets.add(j)
        if len(p) > 2:
            kept.append((p, len(p)))
    return kept


def _random_paths(rng, n_nodes, n_paths):
    paths = set()
    base = [tuple((int(h), int(d)) for h, d in zip(rng.integers(0, n_nodes, L), rng.choice([-1, 1], L)))
            for L in rng.integers(1, 9, n_paths)]
    for p in base:
        paths.add(p)
        if len(p) > 2 and rng.random() < 0.5:       # a piece of it
            a = int(rng.integers(0, len(p) - 1))
            paths.add(p[a:a + int(rng.integers(1, len(p) - a + 1))])
        if rng.random() < 0.3:                      # read the other way along, directions kept
            paths.add(p[::-1])
        if rng.random() < 0.3:                      # a longer one around it
            q = base[int(rng.integers(0, len(base)))]
            paths.add(q[:2] + p + q[-2:])
    return list(paths)


@pytest.mark.parametrize("seed", range(40))
def test_same_survivors_as_the_suffix_tree(seed):
    rng = np.random.default_rng(seed)
    paths = _random_paths(rng, int(rng.integers(3, 30)), int(rng.integers(1, 60)))
    rng.shuffle(paths)
    assert _Stub().filter_paths_between_bubble_starts(list(paths)) == _by_suffix_tree(list(paths))


def test_no_paths_and_short_paths():
    assert _Stub().filter_paths_between_bubble_starts([]) == []
    one = [((1, 1),), ((1, 1), (2, -1))]
    assert _Stub().filter_paths_between_bubble_starts(one) == _by_suffix_tree(one) == []
