#!/usr/bin/env python3
"""Does the ORDER of a batch's guides matter to k_search?  The table probes of an item fall into regions its guide's
symbols name: this strand's seeds without a substitution in X (the first consumed symbols) read one 32-KB piece of
the PAM-pair table, the other strand's seeds without one in R (the last guide symbols) one 256-KB piece of the deep
table.  Guides that share those symbols and run at the same time share the lines in the L2s.  The same 1 M guides,
in the order drawn and sorted by either key: k_search time and hits (the hit set must not depend on the order).
Usage (GPU box, repo root): python tools/sorted_batch.py [workload] [batch] [m]"""
import sys
import zlib
from importlib import import_module
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def keys(seqs, lo, hi):
    code = np.zeros(256, np.uint64)
    for i, c in enumerate(b"ACGT"):
        code[c] = i
    k = np.zeros(seqs.shape[0], np.uint64)
    for t in range(lo, hi):
        k = (k << np.uint64(2)) | code[seqs[:, t]]
    return k


def main():
    import torch
    bench = import_module("bench")
    api = import_module("guidescan-cli_amd.api")
    synth = import_module("guidescan-cli_amd.synth")
    workload = sys.argv[1] if len(sys.argv) > 1 else "hg38"
    lens_name, batch, probs = bench.WORKLOADS[workload]
    if len(sys.argv) > 2:
        batch = int(sys.argv[2])
    m = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    lengths = [synth.CHR1_LENGTH] if lens_name == "CHR1" else getattr(synth, lens_name)
    text, names, lengths = bench.make_workload_genome(synth, workload, lengths, probs)
    g = api.GenomeIndex.build(text, device=0)
    try:
        seqs, pams, _, _ = synth.sample_guides(text, batch, seed=1000)
        orders = {
            "as drawn": np.arange(batch),
            "by X (guide[0:8])": np.argsort(keys(seqs, 0, 8), kind="stable"),
            "by R (guide[14:20])": np.argsort(keys(seqs, 14, 20), kind="stable"),
            "by R then X": np.lexsort((keys(seqs, 0, 8), keys(seqs, 14, 20))),
            "by guide[15:20] then X": np.lexsort((keys(seqs, 0, 8), keys(seqs, 15, 20))),
            "whole guide": np.argsort(keys(seqs, 0, 20), kind="stable"),
        }
        ref = None
        for name, o in orders.items():
            s = torch.from_numpy(np.ascontiguousarray(seqs[o])).cuda()
            p = torch.from_numpy(np.ascontiguousarray(pams[o])).cuda()
            ms = []
            for _ in range(4):
                d_off, d_hits, st = g.enumerate_device(s.data_ptr(), batch, seqs.shape[1], p.data_ptr(), pams.shape[1], mismatches=m)
                ms.append(round(st["ms_search"], 2))
            n_hits = int(st["n_hits"])
            # the hit set per guide must be the one the drawn order gives: compare per-guide counts through the permutation
            off = api.copy_offsets(d_off, batch) if hasattr(api, "copy_offsets") else None
            print(f"{name:28s} k_search {ms} ms, {n_hits} hits", flush=True)
            ref = n_hits if ref is None else ref
            assert n_hits == ref, "the order of a batch changed its hits"
    finally:
        g.close()


if __name__ == "__main__":
    main()
