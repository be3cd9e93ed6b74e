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


# ---------------------------------------------------------------------------------------------------------------------
# The offset-diagonal certificate (k_corr_nw_fast, second shortcut): a corrected list x that is the original list y with
# an end trimmed off and a few genes replaced.  Rule: every gene of x occurs in y at most ONCE, and where it does, at
# i + s for ONE offset s in [0, M - N]; at least two genes match.  Then every optimal alignment runs along diagonal s
# from the first to the last match (gaps elsewhere cost 2 per detour and nothing off the diagonal scores), the ties that
# remain (where the s leading / M - N - s trailing gaps sit among the unmatched genes at either end) never move a
# matched column, and the positions the reference carries over are: matched x[q] -> original position number
# s + (matches before q); unmatched -> (None, None).
def certificate_positions(x, y):
    """None when the rule does not apply, else per gene of x: index into the original positions, or None"""
    n, m_ = len(x), len(y)
    if n > m_ or n == 0:
        return None
    s, matched = None, []
    for i, g in enumerate(x):
        at = [j for j in range(m_) if y[j] == g]
        if not at:
            matched.append(False)
            continue
        if len(at) != 1:
            return None
        if s is None:
            s = at[0] - i
        elif at[0] - i != s:
            return None
        matched.append(True)
    if s is None or not (0 <= s <= m_ - n) or sum(matched) < 2:
        return None
    out, before = [], 0
    for q in range(n):
        out.append(s + before if matched[q] else None)
        before += 1 if matched[q] else 0
    return out


def reference_positions(nw, x, y):
    """process_read_correction's loop over the alignment (construct_graph.py:1314-1325) with position NUMBERS for
    positions: per gene of x the number of the original position it takes over, or None"""
    out, cur = [], 0
    for a, b in nw(None, x, y):
        if a != "*":
            if b != a:
                out.append(None)
            else:
                out.append(cur)
                cur += 1
        else:
            cur += 1
    return out


def test_offset_diagonal_certificate_never_contradicts_the_reference():
    from amira_oracle.graph import GeneMerGraph
    nw = GeneMerGraph.needleman_wunsch
    claimed = 0
    # exhaustive: every pair of lists with len(x) <= len(y) <= 4 over four symbols, 5 over three, 6 over two (x may also
    # hold two symbols y never has).  (Run once with len(y) <= 6 over four symbols: 229 M pairs, 8 minutes, and with
    # <= 5 over four / 6 over three: 24 M pairs — no contradiction; the CPU suite keeps the two million smallest.)
    for m_ in range(1, 7):
        alpha = 4 if m_ <= 4 else (3 if m_ == 5 else 2)
        for n in range(1, m_ + 1):
            for y in itertools.product(range(alpha), repeat=m_):
                for x in itertools.product(range(alpha + 2), repeat=n):
                    want = certificate_positions(x, y)
                    if want is not None:
                        claimed += 1
                        assert reference_positions(nw, list(x), list(y)) == want, (x, y)
    # random longer ones shaped like the re-threaded reads: a slice of y with replaced genes, sometimes a repeated gene
    rng = random.Random(11)
    for _ in range(6000):
        m_ = rng.randint(6, 40)
        y = rng.sample(range(1000), m_)
        if rng.random() < 0.2:
            y[rng.randrange(m_)] = y[rng.randrange(m_)]
        a = rng.randint(0, 4)
        b = m_ - rng.randint(0, 4)
        x = y[a:b]
        for _ in range(rng.randint(0, 5)):
            if x:
                x[rng.randrange(len(x))] = rng.choice([rng.randrange(1000, 1100), rng.choice(y)])
        want = certificate_positions(x, y)
        if want is not None:
            claimed += 1
            assert reference_positions(nw, x, y) == want, (x, y)
    assert claimed > 4000
