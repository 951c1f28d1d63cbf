"""The rule of gs_suffix.hip restated on numpy, step for step with its kernels, against a plain sort of the suffixes: prefix
doubling in which a suffix whose group has one member is never touched again - only the rows of groups that still have company
are gathered, sorted by (rank[s], rank[s + h]), written back to the same rows and regrouped.  What the restatement pins on
the CPU: the first word of the pair keeps every group in its own stretch of rows (the sorted suffixes go back to `pos` in
order), a new group's rank is the row of its first member (a max-scan over heads that carry their row), and a suffix out of
play has its row as its rank.  (The product's arrays are compared with the first builder's and proved against the text on the
GPU: tests/test_gpu_suffix_array.py; the order itself is what csa_wt presumes, sdsl/include/sdsl/csa_wt.hpp:333-346.)"""
import numpy as np
import pytest


def suffix_array_discarding(text: bytes):
    """text without its sentinel; returns (suffix array of text + sentinel, rows in play per round)"""
    t = np.frombuffer(text + b"\0", dtype=np.uint8)
    n = t.shape[0]
    present = np.zeros(256, bool)
    present[t] = True
    dense = np.cumsum(present) - 1                       # k_sx_histogram + the host's dense alphabet
    sigma = int(present.sum())
    bits = 1
    while (1 << bits) < sigma:
        bits += 1
    k0 = 64 // bits
    nbits = 1
    while (1 << nbits) < n:
        nbits += 1
    keys = np.zeros(n, dtype=object)                     # k_sx_init_keys (python ints: 64-bit words without overflow games)
    for i in range(n):
        key = 0
        for j in range(k0):
            key = (key << bits) | (int(dense[t[i + j]]) if i + j < n else 0)
        keys[i] = key
    sa = np.array(sorted(range(n), key=lambda i: keys[i]), dtype=np.int64)   # the first radix sort (stable)
    sk = keys[sa]
    head = np.array([i if (i == 0 or sk[i] != sk[i - 1]) else 0 for i in range(n)], dtype=np.int64)   # k_sx_heads
    grp = np.maximum.accumulate(head)                    # inclusive max-scan
    rank = np.zeros(n, dtype=np.int64)
    rank[sa] = grp                                       # k_sx_scatter_rank

    def company(g):                                      # k_sx_company
        m = g.shape[0]
        f = np.zeros(m, bool)
        f[1:] |= g[1:] == g[:-1]
        f[:-1] |= g[:-1] == g[1:]
        return f

    pos = np.nonzero(company(grp))[0]                    # scan + k_sx_compact
    in_play = [int(pos.shape[0])]
    h = k0
    while pos.shape[0]:
        assert h < 2 * n, "doubling did not converge"
        s = sa[pos]                                      # k_sx_pair_keys
        r1 = rank[s]
        assert (np.diff(r1) >= 0).all()                  # the rows in play are already in the order of their groups
        r2 = np.where(s + h < n, rank[np.minimum(s + h, n - 1)], 0)
        order = np.lexsort((r2, r1))                     # the radix sort of the 64-bit pairs (stable)
        s2, k1, k2 = s[order], r1[order], r2[order]
        hd = np.zeros(pos.shape[0], dtype=np.int64)      # k_sx_heads with the rows as values
        hd[0] = pos[0]
        diff = (k1[1:] != k1[:-1]) | (k2[1:] != k2[:-1])
        hd[1:][diff] = pos[1:][diff]
        ngrp = np.maximum.accumulate(hd)
        sa[pos] = s2                                     # k_sx_apply
        rank[s2] = ngrp
        pos = pos[company(ngrp)]                         # k_sx_company + scan + k_sx_compact
        in_play.append(int(pos.shape[0]))
        h *= 2
    return sa, in_play


def plain(text: bytes):
    t = text + b"\0"
    return np.array(sorted(range(len(t)), key=lambda i: t[i:]), dtype=np.int64)


def texts():
    rng = np.random.default_rng(12)
    acgt = b"ACGT"
    rnd = lambda m: bytes(acgt[i] for i in rng.integers(0, 4, m))
    yield "random", rnd(3000)
    yield "n_runs", rnd(500) + b"N" * 700 + rnd(300) + b"N" * 129 + rnd(40) + b"NNN"
    yield "tandem", rnd(13) * 150
    yield "families", b"".join(rnd(int(rng.integers(0, 9))) + (lambda u: u[:70] + rnd(1) + u[71:])(b"ACGTTGCAAGGCTTAACCGGTTAGCATCGATCGGATCCATGGTACCGAGCTCGAATTCACTGGCCGTCGTTTTACAACGTCGTG" * 2) for _ in range(12))
    yield "other_bytes", bytes(rng.choice(list(b"ACGTNRYKMSWBDHV"), 1500).tolist())
    yield "one_symbol", b"A" * 600
    yield "few", b"ACGTTGCANNACGTACGTAC"
    yield "single", b"G"
    yield "empty", b""


@pytest.mark.parametrize("name,text", list(texts()), ids=[n for n, _ in texts()])
def test_discarding_doubling_makes_the_suffix_array(name, text):
    sa, in_play = suffix_array_discarding(text)
    assert np.array_equal(sa, plain(text)), name
    assert in_play[-1] == 0 and all(a >= b for a, b in zip(in_play, in_play[1:]))   # rows only ever leave
    if name == "n_runs":      # a run of 700 N: its rows stay in play for log2(700 / k0) more rounds, everything else leaves at once
        assert len(in_play) >= 6 and in_play[0] < len(text)
    if name == "random":
        assert len(in_play) <= 2
