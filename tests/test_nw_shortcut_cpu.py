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
    m = len(mism)
    if m <= 1:
        return True
    if m == 2:
        b = mism[1]
        tie_a = all(x[i] == y[i + 1] for i in range(b))
        tie_b = all(x[i + 1] == y[i] for i in range(b))
        return not tie_a and not tie_b
    if m > 4:
        return False
    # m = 3, 4: only the alignments with one gap in each list can reach N - m; the best of them must stay below
    D = [1 if x[i] == y[i] else 0 for i in range(n)]
    best = -10 ** 9
    for S in ([1 if i + 1 < n and x[i] == y[i + 1] else 0 for i in range(n)],
              [1 if i + 1 < n and x[i + 1] == y[i] else 0 for i in range(n)]):
        g = -10 ** 9
        for l in range(n):
            pref_d, pref_s, suf_d = sum(D[:l]), sum(S[:l]), sum(D[l + 1:])
            g = max(g, pref_d - pref_s + (1 if l == 0 else 0))
            best = max(best, g + pref_s + suf_d - 2)
    return best < n - m


def reference_is_diagonal(nw, x, y):
    return nw(None, x, y) == [(a, b) for a, b in zip(x, y)]


def test_shortcut_never_contradicts_the_reference_alignment():
    from amira_oracle.graph import GeneMerGraph
    nw = GeneMerGraph.needleman_wunsch
    claimed = 0
    for n in range(1, 8):
        for alpha in (2, 3):
            if n == 7 and alpha == 3:
                continue
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
        for _ in range(rng.randint(0, 4)):
            x[rng.randrange(n)] = rng.randint(0, 4)
        if shortcut_says_diagonal(x, y):
            claimed += 1
            assert reference_is_diagonal(nw, x, y), (x, y)
    assert claimed > 1000
