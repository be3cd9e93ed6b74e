"""The equal-length shortcut of the position carry-over kernel (k_corr_nw_fast): its decision
rule, restated here in Python, must only ever claim "pure diagonal" for inputs on which the
reference's Needleman-Wunsch (oracle restatement of construct_graph.py:1433-1480, same scores,
borders and tie order) really returns the pure diagonal.  Exhaustive over short lists on small
alphabets, plus random longer ones with tandem arrays."""
import itertools
import random
import sys
import os

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))


def shortcut_says_diagonal(x, y):
    """the rule of amg_passes.hip (k_corr_nw_fast, 'shortcut')"""
    n = len(x)
    if n != len(y):
        return False
    mism = [i for i in range(n) if x[i] != y[i]]
    if len(mism) <= 1:
        return True
    if len(mism) > 2:
        return False
    b = mism[1]
    tie_a = all(x[i] == y[i + 1] for i in range(b))
    tie_b = all(x[i + 1] == y[i] for i in range(b))
    return not tie_a and not tie_b


def reference_is_diagonal(nw, x, y):
    return nw(None, x, y) == [(a, b) for a, b in zip(x, y)]


def test_shortcut_never_contradicts_the_reference_alignment():
    from amira_oracle.graph import GeneMerGraph
    nw = GeneMerGraph.needleman_wunsch
    claimed = 0
    for n in range(1, 7):
        for alpha in (2, 3):
            for x in itertools.product(range(alpha), repeat=n):
                for y in itertools.product(range(alpha), repeat=n):
                    if shortcut_says_diagonal(x, y):
                        claimed += 1
                        assert reference_is_diagonal(nw, list(x), list(y)), (x, y)
    rng = random.Random(7)
    for _ in range(4000):
        n = rng.randint(7, 16)
        y = [rng.randint(0, 3) for _ in range(n)]
        if rng.random() < 0.6:   # tandem array
            at, ln = rng.randrange(n), rng.randint(2, 6)
            y[at:at + ln] = [y[at]] * min(ln, n - at)
        x = list(y)
        if rng.random() < 0.5:   # shift a stretch by one
            x = x[1:] + [rng.randint(0, 3)]
        for _ in range(rng.randint(0, 2)):
            x[rng.randrange(n)] = rng.randint(0, 4)
        if shortcut_says_diagonal(x, y):
            claimed += 1
            assert reference_is_diagonal(nw, x, y), (x, y)
    assert claimed > 1000
