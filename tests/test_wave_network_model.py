"""The schedule of k_to_wsort's network (gs_tileorder.hip: to_wsort), restated on numpy and run on the CPU: 64 lanes x 8
registers, place 8 * lane + u; the lane's eight ordered up in even lanes and down in odd ones, then for runs of
16, 32, ... 512 places the crossing steps at lane distances run/16 ... 1 (a lane keeps the larger word when it is the
upper lane of an ascending pair or the lower of a descending one) and the three local steps at register distances 4, 2, 1.
This is a model of the schedule, not of the kernel: it says that the 45 steps order ANY 512 words ascending by place -
which the zero-one principle turns into a finite check on 0/1 inputs - so that a change of the schedule in the kernel has
something to be compared with.  (The kernel itself is compared with the oracle on the GPU: tests/test_gpu_parity.py.)"""
import numpy as np
import pytest

LANES, REGS = 64, 8


def local_ce(k, i, j, desc):
    """registers i < j of every lane: the smaller to i where the lane's direction is up"""
    sw = (k[:, i] > k[:, j]) != desc
    a, b = k[:, i].copy(), k[:, j].copy()
    k[:, i] = np.where(sw, b, a)
    k[:, j] = np.where(sw, a, b)


def cross(k, d, asc):
    lane = np.arange(LANES)
    partner = k[lane ^ d]
    keep_max = ((lane & d) == 0) != asc
    take = (partner < k) != keep_max[:, None]
    return np.where(take, partner, k)


def wsort(words):
    k = words.reshape(LANES, REGS).copy()
    lane = np.arange(LANES)
    desc = (lane & 1) != 0
    for i, j in ((0, 1), (2, 3), (4, 5), (6, 7), (0, 2), (1, 3), (4, 6), (5, 7), (1, 2), (5, 6),
                 (0, 4), (1, 5), (2, 6), (3, 7), (2, 4), (3, 5), (1, 2), (3, 4), (5, 6)):   # Batcher, 19 exchanges
        local_ce(k, i, j, desc)
    run_bit = 2
    while run_bit <= 64:                      # runs of 16 .. 512 places: lanes whose bit `run_bit` is clear go up
        asc = (lane & run_bit) == 0 if run_bit < 64 else np.ones(LANES, dtype=bool)
        d = run_bit // 2
        while d >= 1:
            k = cross(k, d, asc)
            d //= 2
        for dist in (4, 2, 1):
            for i in range(REGS):
                if i & dist == 0:
                    local_ce(k, i, i | dist, ~asc)
        run_bit *= 2
    return k.reshape(-1)


@pytest.mark.parametrize("seed", range(8))
def test_random_words_come_out_in_order(seed):
    rng = np.random.default_rng(seed)
    w = rng.integers(0, 2**63, LANES * REGS, dtype=np.int64).astype(np.uint64)
    w[rng.integers(0, w.size, 40)] = w[0]          # equal words among them
    n = int(rng.integers(1, w.size + 1))
    w[n:] = np.uint64(0xFFFFFFFFFFFFFFFF)          # padding places hold words of all ones
    out = wsort(w)
    assert np.array_equal(out, np.sort(w))


def test_zero_one_inputs():
    """a comparison network orders every input if it orders every 0/1 input; 2^512 of them are out of reach, so: every
    threshold position of every cyclic shift of a sorted 0/1 row, of its reversal, and 2,000 random ones"""
    rng = np.random.default_rng(1)
    n = LANES * REGS
    rows = []
    for ones in range(0, n + 1, 7):
        base = np.zeros(n, dtype=np.uint64)
        base[n - ones:] = 1
        for sh in (0, 1, 63, 64, 65, 255, 256, 257):
            rows.append(np.roll(base, sh))
            rows.append(np.roll(base[::-1], sh))
    rows += [(rng.random(n) < p).astype(np.uint64) for p in rng.random(2000)]
    for r in rows:
        assert np.array_equal(wsort(r), np.sort(r))
