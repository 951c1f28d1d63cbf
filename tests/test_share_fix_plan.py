"""k_share_fix's plan, restated on numpy (gs_order.hip): after a launch that shared verification passes an item's records lie in its
slots, in the owner's arena chunks and in chunks of their own that the helping waves filled - each helper chunk full but the last of
its episode.  The ordering kernels expect "slots, then chunks 0, 1, .. in order, all full but the last" (process.hpp:100-115 is what
they finally reproduce): the fix moves the records that lie beyond place T = the item's total into the holes before it - mover r
(the r-th record beyond T, segment by segment) goes to hole r (the r-th free place before T).  This file pins that plan: every
record is kept exactly once, and afterwards segment s holds exactly the places of [start(s), start(s) + room(s)) that lie before T.
(The kernel's bytes are checked against the plain form's on the GPU: tests/test_gpu_parity.py, the sharing modes.)"""
import numpy as np
import pytest
from hypothesis import given, settings, strategies as st

CHUNK = 1024


def plan(fill, cap):
    """fill[s]: records in segment s (0 = the slots, s >= 1 = chunk s - 1 of the item's directory).  Returns (T, keep[s], moves)
    with moves = [(src segment, src index, dst segment, dst index)] exactly as the kernel's threads compute them."""
    fill = np.asarray(fill, dtype=np.int64)
    nseg = fill.shape[0]
    room = np.full(nseg, CHUNK, dtype=np.int64)
    room[0] = cap
    start = np.concatenate([[0], cap + CHUNK * np.arange(nseg - 1)])
    T = int(fill.sum())
    inside = np.clip(T - start, 0, room)                 # places of the segment before T
    hole = np.maximum(inside - fill, 0)
    mov = np.maximum(fill - inside, 0)
    assert hole.sum() == mov.sum()                       # (V = T: what the kernel checks before it moves anything)
    hpre = np.concatenate([[0], np.cumsum(hole)])        # exclusive prefixes, one more entry at the end
    mpre = np.concatenate([[0], np.cumsum(mov)])
    moves = []
    for r in range(int(mov.sum())):
        sm = int(np.searchsorted(mpre, r, side="right") - 1)      # the last segment whose prefix is <= r ...
        while mov[sm] == 0:                                       # ... (segments without movers share a prefix with their successor:
            sm += 1                                               #      the kernel's search lands on the last of them, which has movers)
        in_m = int(min(max(T - start[sm], 0), fill[sm]))
        sh = int(np.searchsorted(hpre, r, side="right") - 1)
        while hole[sh] == 0:
            sh += 1
        moves.append((sm, in_m + r - int(mpre[sm]), sh, int(fill[sh]) + r - int(hpre[sh])))
    return T, inside, moves


def apply(fill, cap, moves):
    nseg = len(fill)
    seg = [np.full(cap if s == 0 else CHUNK, -1, dtype=np.int64) for s in range(nseg)]
    rid = 0
    for s in range(nseg):
        seg[s][:fill[s]] = np.arange(rid, rid + fill[s])
        rid += fill[s]
    for sm, im, sh, ih in moves:
        assert seg[sm][im] >= 0 and seg[sh][ih] == -1, (sm, im, sh, ih)     # a record moves into a free place
        seg[sh][ih] = seg[sm][im]
        seg[sm][im] = -1
    return seg, rid


def layout(cap, own, helper_fills):
    """an item as k_search leaves it: `own` records of the owner (slots, then full chunks, then a partial one), then the helpers'
    chunks with whatever they hold"""
    fill = [min(own, cap)]
    rest = max(own - cap, 0)
    while rest > 0:
        fill.append(min(rest, CHUNK))
        rest -= min(rest, CHUNK)
    return fill + list(helper_fills)


@settings(max_examples=300, deadline=None)
@given(cap=st.sampled_from([64, 256, 1024]), own=st.integers(0, 5000),
       helpers=st.lists(st.integers(1, CHUNK), min_size=0, max_size=12))
def test_every_record_once_and_a_dense_layout(cap, own, helpers):
    fill = layout(cap, own, helpers)
    T, inside, moves = plan(fill, cap)
    seg, n = apply(fill, cap, moves)
    assert n == T
    kept = np.concatenate([seg[s][:inside[s]] for s in range(len(fill))])
    assert (kept >= 0).all() and np.array_equal(np.sort(kept), np.arange(T))      # every record exactly once, no gap before T
    for s in range(len(fill)):
        assert (seg[s][inside[s]:] == -1).all()                                      # nothing left beyond it
    full_chunks = max(T - cap, 0) // CHUNK
    assert all(inside[1 + j] == CHUNK for j in range(full_chunks))                   # "all full but the last"


def test_nothing_moves_when_nothing_was_shared_out_of_place():
    for cap, own in ((64, 10), (64, 64), (64, 64 + 3 * CHUNK), (64, 64 + 3 * CHUNK + 17)):
        T, inside, moves = plan(layout(cap, own, []), cap)
        assert T == own and moves == []


def test_a_hand_made_case():
    # slots of 64 holding 64, one owner chunk with 100, two helper chunks with 1,024 and 5: T = 1,193 -> chunk 0 is filled up from
    # the tail (924 places), chunk 1 keeps 105 of its 1,024, chunk 2 is emptied
    fill = [64, 100, 1024, 5]
    T, inside, moves = plan(fill, 64)
    assert T == 1193 and list(inside) == [64, 1024, 105, 0]
    assert len(moves) == 924 and moves[0] == (2, 105, 1, 100) and moves[-1][0] == 3
    seg, _ = apply(fill, 64, moves)
    assert (seg[3] == -1).all() and (seg[1] >= 0).all()
