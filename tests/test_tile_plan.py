"""The tile ordering's bucket plan as data (host only; gs_tileorder.hip: to_buckets).  An item of more than 4,096 match
records is dealt into buckets by splitters taken from a sample; a bucket's size is then a gamma variate of the shape
"sample words per splitter", and these must hold whatever the item's size: the sample fits the 8,192 words one workgroup
orders; a bucket's slot is 1.3-1.8 x the aim, and what a bucket receives beyond it goes to a spill list while the bucket
moves to a slot of its real size (k_to_respill) - the spill list (an eighth of the dealt records) and the room kept for
moved buckets (an eighth of the planned slots) are several times what the model expects; a moved bucket fits the 8,192
records one workgroup orders; and most buckets fit the 512 records one wave orders.  (Until round 5 the slots were 2.9-6.2 x
the aim so that no bucket ever outgrew one: 3.7 x the records in bucket space.)"""
from importlib import import_module

import numpy as np
import pytest
from scipy.stats import gamma

api = import_module("guidescan-cli_amd.api")

SIZES = sorted(set([1, 511, 512, 513, 4096, 4097, 5000, 51_200, 51_201, 102_400, 102_401, 163_840, 163_841, 294_912, 294_913,
                    524_288, 1_048_576] + [int(x) for x in np.geomspace(4097, 1_048_576, 220)]))


def test_small_items_are_one_tile():
    for c in (0, 1, 512, 4096):
        assert api.tile_plan(c)["buckets"] == 0


@pytest.mark.parametrize("c", SIZES)
def test_bucket_plan_invariants(c):
    p = api.tile_plan(c)
    if c <= 4096:
        assert p["buckets"] == 0
        return
    nb, slot, per = p["buckets"], p["slot"], p["per"]
    assert 2 <= nb <= p["max_buckets"]
    assert 4 <= per <= 32 and per * nb <= 8192 and per * nb <= c        # one sample word per stretch of the item
    aim = c / nb
    # items to 3 x 10^5 records aim below one wave's 512; beyond, at 1,024 (the workgroup kernels take tiles of up to 8,192)
    assert aim <= (512 if c <= 294_912 else 1024) and slot % 128 == 0 and slot <= 8192
    # a bucket's size: aim x Gamma(per) / per.  Records beyond the slot (expected share of the item's records), buckets
    # that move (a moved bucket takes its size in units of 128 records: about the slot + 1), and the largest moved bucket
    a = slot / aim * per
    p_over = gamma.sf(a, per)
    spilled = (per * gamma.sf(a, per + 1) - a * gamma.sf(a, per)) / per
    assert spilled <= 0.125 / 4, (c, nb, slot, per, spilled)                     # the spill list: an eighth of the records
    assert p_over * (slot / 128 + 1) / (slot / 128) <= 1 / 8 / 1.15, (c, p_over)  # room for moved buckets: an eighth of the slots
    assert gamma.sf(8192 / aim * per, per) < 1e-10, (c, nb, slot, per)           # a moved bucket is one workgroup tile
    # buckets beyond one wave's 512 records are served, by the slower kernels: few of them where the sample allows
    if per >= 16 and c <= 294_912:
        assert gamma.sf(512 / aim * per, per) < 0.07


def test_items_beyond_the_plan_are_left_to_the_device_wide_ordering():
    p = api.tile_plan(1_048_577)
    assert p["buckets"] > p["max_buckets"]     # k_to_plan / k_to_fill leave the GUIDE out of the tiles (gs_enumerate.hip orders it alone)


def test_bucket_space_stays_within_a_small_multiple_of_the_records():
    for c in (10_000, 46_000, 100_000, 150_000, 219_515, 438_204, 1_000_000):
        p = api.tile_plan(c)
        assert p["buckets"] * p["slot"] * (1 + 1 / 8) <= 2.2 * c
    p = api.tile_plan(46_000)     # the repeat-rich batch's average item: slots + the room for moved buckets
    assert p["buckets"] * p["slot"] * (1 + 1 / 8) <= 1.5 * 46_000
