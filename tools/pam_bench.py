#!/usr/bin/env python3
"""k_search / step time for a PAM other than NGG, also at the 5' end (--start: Cas12a's TTTN):
    python tools/pam_bench.py [--workload hg38] [--pam TTTN] [--start] [--batch 1000000] [--mismatches 3] [--L 20]
Guides are sampled on-target (PAM + protospacer read off the genome, both strands), so every guide must
report its own site at distance 0; the NGG batch of bench.py is timed next to it on the same index."""
import argparse
import sys
import time
from importlib import import_module
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402

api = import_module("guidescan-cli_amd.api")
synth = import_module("guidescan-cli_amd.synth")


def sample_sites(text, n, pam, L, start, seed):
    """on-target guides for a PAM before (--start) or behind the protospacer; returns seqs [n, L], pams [n, P]"""
    rng = np.random.Generator(np.random.PCG64(seed))
    P = len(pam)
    comp = synth._COMP
    acgt = np.zeros(256, dtype=bool)
    acgt[list(b"ACGT")] = True
    out = np.empty((n, L), dtype=np.uint8)
    got = 0
    W = L + P
    ar = np.arange(W)[None, :]
    pat = np.frombuffer(pam, np.uint8)
    while got < n:
        m = int(min(max(65536, (n - got) * 64), 4_000_000))
        c = rng.integers(0, text.shape[0] - W, size=m)
        minus = rng.random(m) < 0.5
        win = text[c[:, None] + ar]
        win[minus] = comp[win[minus][:, ::-1]]
        pam_part = win[:, :P] if start else win[:, L:]
        guide = win[:, P:] if start else win[:, :L]
        ok = acgt[win].all(axis=1)
        for j in range(P):
            if pat[j] != ord("N"):
                ok &= pam_part[:, j] == pat[j]
        sel = np.nonzero(ok)[0][: n - got]
        out[got:got + sel.size] = guide[sel]
        got += sel.size
    return out, np.tile(pat, (n, 1))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="hg38")
    ap.add_argument("--pam", default="TTTN")
    ap.add_argument("--start", action="store_true")
    ap.add_argument("--batch", type=int, default=1_000_000)
    ap.add_argument("--mismatches", type=int, default=3)
    ap.add_argument("--L", type=int, default=20)
    a = ap.parse_args()
    lname, _, probs = bench.WORKLOADS[a.workload]
    lens = [synth.CHR1_LENGTH] if lname == "CHR1" else getattr(synth, lname)
    text, names, lengths = bench.make_workload_genome(synth, a.workload, lens, probs)
    g = api.GenomeIndex.build(text, device=0)
    print(f"index {g.device_bytes / 1e9:.1f} GB", flush=True)
    runs = [("NGG", False, *synth.sample_guides(text, a.batch, seed=5)[:2]),
            (a.pam, a.start, *sample_sites(text, a.batch, a.pam.encode(), a.L, a.start, 6))]
    for label, start, seqs, pams in runs:
        d_s = torch.from_numpy(np.ascontiguousarray(seqs)).cuda()
        d_p = torch.from_numpy(np.ascontiguousarray(pams)).cuda()
        for rep in range(3):
            t0 = time.perf_counter()
            d_off, d_hits, st = g.enumerate_device(d_s.data_ptr(), a.batch, seqs.shape[1], d_p.data_ptr(), pams.shape[1],
                                                   mismatches=a.mismatches, start=start)
            wall = (time.perf_counter() - t0) * 1e3
            c = g.last_counters()
            print(f"{label}{' --start' if start else ''} L={seqs.shape[1]}: run {rep}: k_search {st['ms_search']:.1f} ms, call {wall:.1f} ms, "
                  f"{a.batch / wall * 1e3:.3g} guides/s, hits {st['n_hits']}, two-sided items {c['items_two_sided']}, one-sided {c['items_one_sided']}, "
                  f"through PAM-pair tables {c['items_pair_tables']} of {2 * a.batch}, index now {g.device_bytes / 1e9:.1f} GB", flush=True)
        del d_s, d_p


if __name__ == "__main__":
    main()
