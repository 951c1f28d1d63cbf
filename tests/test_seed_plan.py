"""The seed plan of k_search as data (host only): the recipe lists must name exactly the depth-k nodes
the reference's recursion reaches (index.hpp:182-248: every variant of the first k consumed symbols with
at most m substitutions), and under two-sided seeding the two strands' shares must cover every class of
sites (a, o, b substitutions in X, O, R) exactly once."""
from importlib import import_module
from itertools import combinations, product
from math import comb

import numpy as np
import pytest

api = import_module("guidescan-cli_amd.api")


def decode(w):
    """recipe word -> (frozenset of (step, digit), lower bound, rotated copy step or None)"""
    w = int(w)
    n, lo, rot, rs = w & 7, (w >> 3) & 7, (w >> 6) & 1, (w >> 7) & 31
    subs = []
    for i in range(7):
        f = (w >> (12 + 7 * i)) & 127
        if i < n:
            assert f & 3 != 3
            subs.append((f >> 2, f & 3))
        else:
            assert f == 3   # unused fields point at the table's "no substitution" entry
    assert len({s for s, _ in subs}) == n   # one substitution per step
    return frozenset(subs), lo, (rs if rot else None)


def all_variants(steps, m):
    out = set()
    for j in range(min(m, len(steps)) + 1):
        for pos in combinations(steps, j):
            for digs in product(range(3), repeat=j):
                out.add(frozenset(zip(pos, digs)))
    return out


@pytest.mark.parametrize("k,m", [(8, 0), (8, 2), (10, 3), (12, 4), (14, 3)])
def test_one_sided_list_is_every_variant_once(k, m):
    full, a, b = api.seed_recipes(k, 20, 3, m, 4)
    assert len(a) == 0 and len(b) == 0
    got = [decode(w) for w in full]
    sets = [g[0] for g in got]
    assert len(set(sets)) == len(sets) == sum(comb(k, j) * 3 ** j for j in range(m + 1))
    assert set(sets) == all_variants(range(k), m)
    for s, lo, rot in got:
        assert lo == 0
        if rot is not None:   # the copy of the last substituted step (or of step k-2 for the second-last symbol)
            assert rot <= k - 2 and rot in {st for st, _ in s}


GEOM = [  # k, L, P, m, |X|, thresholds, deep
    (14, 20, 3, 3, 9, (2, 2, 1, 1), False),
    (14, 20, 3, 3, 8, (2, 2, 1, 1), True),
    (13, 20, 3, 4, 10, (3, 2, 2, 1, 1), False),
    (12, 18, 3, 5, 7, (3, 3, 2, 2, 1, 1), True),
    (11, 16, 3, 2, 8, (2, 1, 1), False),
    (12, 20, 0, 3, 8, (2, 2, 1, 1), False),
    (14, 20, 3, 6, 8, (4, 3, 3, 2, 2, 1, 1), True),
    (14, 23, 4, 3, 13, (2, 2, 1, 1), False),   # X reaches the table's two-symbol extension (step k-2)
    (14, 23, 4, 4, 13, (3, 2, 2, 1, 1), False),
    (12, 20, 3, 3, 11, (2, 2, 1, 1), False),
]


@pytest.mark.parametrize("k,L,P,m,nx,astar,deep", GEOM)
def test_two_sided_shares_cover_every_site_class_once(k, L, P, m, nx, astar, deep):
    """this strand enumerates (a, o) with a < a*(o) over its k-mer (X then O) and finds every b; the other
    strand enumerates (o, b) over its k-mer (R then O, counted from the guide's end) with the lower bound
    a >= a*(o): each (a, o, b) with a + o + b <= m is then found by exactly one side"""
    astar8 = (list(astar) + [15] * 8)[:8]
    full, ra, rb = api.seed_recipes(k, L, P, m, nx, astar8, deep)
    n_o, n_r = k - nx, L - k
    # this strand's share: exactly the variants of the k-mer with a < a*(o)
    a_sets = [decode(w)[0] for w in ra]
    assert len(set(a_sets)) == len(a_sets)
    want = {s for s in all_variants(range(k), m)
            if sum(1 for st, _ in s if st < nx) < astar8[min(7, sum(1 for st, _ in s if st >= nx))]}
    assert set(a_sets) == want
    # the other strand's: substitutions over y = 0 .. n_r + n_o - 1 (guide symbol L-1-y): R first, then O
    b_dec = [decode(w) for w in rb]
    b_sets = [d[0] for d in b_dec]
    assert len(set(b_sets)) == len(b_sets)
    cover = {}
    for s, lo, rot in b_dec:
        b = sum(1 for y, _ in s if y < n_r)
        o = len(s) - b
        assert all(y < n_r + n_o for y, _ in s)
        assert lo == min(7, astar8[o]) and astar8[o] + o + b <= m
        cover[(o, b)] = cover.get((o, b), 0) + 1
        if deep:
            assert rot is None   # a deep table's line is one index: no copies
    for (o, b), cnt in cover.items():
        assert cnt == comb(n_o, o) * comb(n_r, b) * 3 ** (o + b)
    for a in range(0, min(nx, m) + 1):
        for o in range(0, min(n_o, m - a) + 1):
            for b in range(0, min(n_r, m - a - o) + 1):
                mine = a < astar8[o]
                theirs = (o, b) in cover and a >= astar8[o]
                assert mine != theirs, (a, o, b)


def test_cost_model_thresholds_at_the_baseline_geometry():
    """hg38-sized tables (k = 14), 20-mers + NGG: the thresholds the measurements in DESIGN.md section 5.1 were taken
    with - strand tables on both sides, and PAM-pair + deep tables"""
    assert api.choose_thresholds(3, 9, 5, 6, 4.0)[:4] == [2, 2, 1, 1]
    assert api.choose_thresholds(6, 9, 5, 6, 4.0)[:6] == [4, 3, 3, 2, 2, 1]
    assert api.choose_thresholds(3, 8, 6, 6, 1.0, 0.4, 1.6)[:4] == [2, 2, 1, 1]
    assert api.choose_thresholds(4, 8, 6, 6, 1.0, 0.4, 1.6)[:5] == [3, 2, 2, 1, 1]
    assert api.choose_thresholds(6, 8, 6, 6, 1.0, 0.4, 1.6)[:7] == [3, 3, 3, 2, 2, 1, 1]   # profiles/r02_sweep_astar_deep.txt
    for m in range(1, 8):
        t = api.choose_thresholds(m, 8, 6, 6, 1.0, 0.4, 1.6)
        assert all(t[i] >= t[i + 1] for i in range(7))   # non-increasing in o
