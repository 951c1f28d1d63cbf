"""The 64-byte descriptor an item of the two seeding launches starts from (gs_seed.hip: k_describe), field by field against
a restatement in numpy/Python of what each field means for the reference's search (index.hpp:182-248: the query is
consumed symbol by symbol, a k-mer of the consumed symbols indexes the interval table; process.hpp:51-63: the PAM list).
Host only (gs_debug_guide_descriptor runs the very function the kernel runs)."""
import random
from importlib import import_module

import pytest

api = import_module("guidescan-cli_amd.api")


def restate(q_syms, pams, L, P, k, x_len, codes, n_pt):
    """q_syms[t] = 2-bit code of the t-th consumed symbol; pams = lists of P codes (0-3, 4 = N) in consumption order"""
    sx, nYb = x_len, L - x_len
    d = {}
    q = sum(s << (2 * t) for t, s in enumerate(q_syms))
    d["q_lo"], d["q_hi"] = q & 0xFFFFFFFF, q >> 32
    # this strand's table index: the first k consumed symbols, the first one most significant
    d["pidx0"] = sum(q_syms[t] << (2 * (k - 1 - t)) for t in range(k))
    # the other strand's deep table: the complemented LAST nYb guide symbols, the last one most significant
    d["pidxg"] = sum((3 - q_syms[L - 1 - y]) << (2 * (nYb - 1 - y)) for y in range(nYb))
    # what the other side's verification compares: the complemented first |X| symbols, last first
    d["qrem_b"] = sum((3 - q_syms[sx - 1 - j]) << (2 * j) for j in range(sx))
    # a deep-table entry's four 16-bit pair masks (context offsets 0, 2, 4, 6): the bit of the guide's own pair in each
    z = w = npairs = 0
    for j in range(4):
        o = 2 * j
        if o + 1 < sx:
            v = (3 - q_syms[sx - 1 - o]) | ((3 - q_syms[sx - 2 - o]) << 2)
            if j < 2:
                z |= 1 << (16 * j + v)
            else:
                w |= 1 << (16 * (j - 2) + v)
            npairs += 1
    d["bsel_z"], d["bsel_w"] = z, w
    # the nearest six symbols after the table depth as one-hot nibbles (PAM-pair table filters)
    gA = L - k
    d["qhot"] = sum(1 << (4 * j + q_syms[k + j]) for j in range(min(6, gA)))
    pslots = bits = 0
    for pj, pw in enumerate(pams):
        c0, c1, cn = pw[P - 2], pw[P - 1], pw[0]
        bslot = 1 if (n_pt > 1 and (c0 | (c1 << 2)) == codes[1]) else 0
        pslots |= 1 << bslot
        bits |= bslot << (8 + pj)
        bits |= (15 if cn == 4 else 1 << (3 - cn)) << (12 + 4 * pj)
    d["meta"] = len(pams) | (pslots << 3) | (npairs << 5) | bits
    xa, rb = min(sx, 8), min(L - k, 8)
    d["key_a"] = d["pidx0"] >> (2 * (k - xa))
    d["key_b"] = d["pidxg"] >> (2 * (nYb - rb)) if rb <= nYb else 0
    for j in range(4):
        d[f"pam{j}"] = sum(c << (3 * u) for u, c in enumerate(pams[j])) if j < len(pams) else 0
    return q, d


@pytest.mark.parametrize("L,P,k,x_len", [(20, 3, 14, 8), (20, 3, 13, 9), (23, 3, 14, 11), (20, 3, 12, 10), (21, 3, 14, 9)])
def test_every_field_of_the_descriptor(L, P, k, x_len):
    rng = random.Random(1000 * L + k)
    for trial in range(200):
        q_syms = [rng.randrange(4) for _ in range(L)]
        n_pt = rng.choice([1, 2])
        pair = [rng.randrange(4), rng.randrange(4)]
        pair2 = [rng.randrange(4), rng.randrange(4)]
        codes = (pair[0] | (pair[1] << 2), (pair2[0] | (pair2[1] << 2)) if n_pt > 1 else 0xFFFFFFFF)
        pams = []
        for _ in range(rng.randrange(1, 5)):   # every pattern ends (as consumed) in a pair that has a table slot
            pr = pair if (n_pt == 1 or rng.random() < 0.5) else pair2
            pams.append([rng.choice([0, 1, 2, 3, 4])] + [rng.randrange(4) for _ in range(P - 3)] + pr)
        q, exp = restate(q_syms, pams, L, P, k, x_len, codes, n_pt)
        got = api.guide_descriptor(q, [exp[f"pam{j}"] for j in range(len(pams))], L, P, k, x_len, codes, n_pt)
        for key, v in exp.items():
            assert got[key] == v, (trial, key, hex(got[key]), hex(v))
        assert got["guide"] == 0
        # the scheduling keys name the table pieces the launches share: the items of one key read the same piece
        assert got["key_a"] < 65536 and got["key_b"] < 65536


def test_a_guide_that_is_not_valid_has_no_patterns():
    got = api.guide_descriptor(0x123456789, [0b100_010_010], 20, 3, 14, 8, valid=False)
    assert got["meta"] & 7 == 0
