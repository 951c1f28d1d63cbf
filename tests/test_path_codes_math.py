"""The 2-bit-field arithmetic k_search uses for a hit's path codes (gs_search.hip: path_codes16, rev_fields16),
restated in numpy and checked against the per-symbol rule it replaced (index.hpp:230-247 as the walk writes it:
0 where text and query agree, else the text base's place among the three other bases, A<C<G<T, counted from 1).
Device code is exercised by the -m gpu parity tests; this guards the formula itself."""
import numpy as np

M32 = np.uint64(0xFFFFFFFF)


def path_codes16(T, Q):
    T = T.astype(np.uint64)
    Q = Q.astype(np.uint64)
    x = T ^ Q
    ne = (x | (x >> np.uint64(1))) & np.uint64(0x55555555)
    nq = (~T & M32) & Q
    lt = ((nq >> np.uint64(1)) | (((~x & M32) >> np.uint64(1)) & nq)) & np.uint64(0x55555555)
    return ((T + lt) & (ne * np.uint64(3))) & M32


def rev_fields16(x):
    x = x.astype(np.uint64)
    r = np.zeros_like(x)
    for b in range(32):
        r |= ((x >> np.uint64(b)) & np.uint64(1)) << np.uint64(31 - b)
    return (((r >> np.uint64(1)) & np.uint64(0x55555555)) | ((r & np.uint64(0x55555555)) << np.uint64(1))) & M32


def loop_codes(T, Q, g):
    out = np.zeros(T.shape, np.uint64)
    for v in range(g):
        tb = (T >> np.uint64(2 * v)) & np.uint64(3)
        qc = (Q >> np.uint64(2 * v)) & np.uint64(3)
        code = np.where(tb == qc, 0, 1 + tb - (tb > qc)).astype(np.uint64)
        out |= code << np.uint64(2 * v)
    return out


def test_codes_equal_the_per_symbol_rule():
    rng = np.random.default_rng(1)
    for g in (1, 5, 6, 8, 13, 16):
        mask = np.uint64((1 << (2 * g)) - 1)
        T = rng.integers(0, 1 << 32, 20000, dtype=np.uint64) & mask
        Q = rng.integers(0, 1 << 32, 20000, dtype=np.uint64) & mask
        Q[:2000] = T[:2000] ^ (np.uint64(1) << rng.integers(0, 2 * g, 2000).astype(np.uint64))  # near-equal words
        assert np.array_equal(path_codes16(T, Q), loop_codes(T, Q, g)), g


def test_codes_land_on_the_path_bits_of_both_sides():
    """this strand's side: word symbol v is guide symbol k+v, code at path bit 50 - 2 (k+v) (fields reversed, then
    one shift); the other strand's side: word symbol j is guide symbol g-1-j, complemented, code at 50 - 2 (g-1-j)"""
    rng = np.random.default_rng(2)
    for k, g in ((14, 6), (10, 10), (8, 12), (4, 16), (16, 4)):
        mask = np.uint64((1 << (2 * g)) - 1)
        w = rng.integers(0, 1 << 32, 5000, dtype=np.uint64) & mask
        q = rng.integers(0, 1 << 32, 5000, dtype=np.uint64) & mask
        codes = loop_codes(w, q, g)
        want = np.zeros(w.shape, np.uint64)
        for v in range(g):
            want |= ((codes >> np.uint64(2 * v)) & np.uint64(3)) << np.uint64(50 - 2 * (k + v))
        got = (rev_fields16(path_codes16(w, q)) << np.uint64(32)) >> np.uint64(12 + 2 * k)
        assert np.array_equal(got, want), (k, g)
    for g in (6, 8, 9, 16):
        mask = np.uint64((1 << (2 * g)) - 1)
        w = rng.integers(0, 1 << 32, 5000, dtype=np.uint64) & mask           # the other strand's context word
        gq = rng.integers(0, 1 << 32, 5000, dtype=np.uint64) & mask          # guide symbols t = 0 .. g-1
        qrem_b = np.zeros(w.shape, np.uint64)                                # complemented, last first
        for j in range(g):
            qrem_b |= (np.uint64(3) - ((gq >> np.uint64(2 * (g - 1 - j))) & np.uint64(3))) << np.uint64(2 * j)
        want = np.zeros(w.shape, np.uint64)
        for j in range(g):
            t = g - 1 - j
            qc = (gq >> np.uint64(2 * t)) & np.uint64(3)
            tb = np.uint64(3) - ((w >> np.uint64(2 * j)) & np.uint64(3))
            code = np.where(tb == qc, 0, 1 + tb - (tb > qc)).astype(np.uint64)
            want |= code << np.uint64(50 - 2 * t)
        got = path_codes16((~w) & mask, (~qrem_b) & mask) << np.uint64(52 - 2 * g)
        assert np.array_equal(got, want), g
