#!/usr/bin/env python3
"""Experiment: does the order of the guides inside a repeat-rich batch matter to k_search?  One wave takes one (guide,
strand) item; the heaviest items of the hg38rep batch hold 2 x 10^5 records.  The same 20,000 guides three ways: as
sampled, heaviest first (by the hit counts of the first run), lightest first.
Usage (GPU box, repo root): python tools/rep_lpt.py"""
import ctypes as C
import sys
import time
from importlib import import_module
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def main():
    import torch
    bench = import_module("bench")
    api = import_module("guidescan-cli_amd.api")
    synth = import_module("guidescan-cli_amd.synth")
    lens_name, _, probs = bench.WORKLOADS["hg38rep"]
    text, names, lengths = bench.make_workload_genome(synth, "hg38rep", getattr(synth, lens_name), probs)
    g = api.GenomeIndex.build(text, device=0)
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    try:
        n = 20000
        seqs, pams, _, _ = synth.sample_guides(text, n, seed=1000)

        def run(order, tag):
            d_s, d_p = torch.from_numpy(seqs[order]).cuda(), torch.from_numpy(pams[order]).cuda()
            best = None
            for _ in range(4):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                d_off, _, st = g.enumerate_device(d_s.data_ptr(), n, 20, d_p.data_ptr(), 3, mismatches=3)
                torch.cuda.synchronize()
                dt = 1e3 * (time.perf_counter() - t0)
                best = (dt, st["ms_search"]) if best is None or dt < best[0] else best
            off = np.empty(n + 1, dtype=np.int64)
            assert hip.hipMemcpy(off.ctypes.data, d_off, 8 * (n + 1), 2) == 0
            print(f"{tag}: step {best[0]:.1f} ms, k_search {best[1]:.1f} ms, {int(off[-1])} hits, largest guide {int(np.diff(off).max())}", flush=True)
            return np.diff(off)

        ident = np.arange(n)
        cnt = run(ident, "as sampled")
        heavy = np.argsort(-cnt, kind="stable")
        run(heavy, "heaviest first")
        run(heavy[::-1].copy(), "lightest first")
    finally:
        g.close()


if __name__ == "__main__":
    main()
