#!/usr/bin/env python3
"""Sweep of the two-sided thresholds astar(o) on one resident index: times
k_search for each candidate vector (GS_ASTAR) next to the cost model's own choice.
    python tools/sweep_astar.py [workload=hg38] [m=6] [batch=20000] 'a0,a1,..' ['a0,a1,..' ...]"""
import json
import os
import sys
from importlib import import_module
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def main():
    import torch
    import bench
    api = import_module("guidescan-cli_amd.api")
    synth = import_module("guidescan-cli_amd.synth")
    workload, m, batch = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    lens_name, _, probs = bench.WORKLOADS[workload]
    lengths = [synth.CHR1_LENGTH] if lens_name == "CHR1" else getattr(synth, lens_name)
    text, names, lengths = bench.make_workload_genome(synth, workload, lengths, probs)
    gidx = api.GenomeIndex.build(text, device=0)
    seqs, pams, _, _ = synth.sample_guides(text, batch, seed=1000)
    d_s, d_p = torch.from_numpy(seqs).cuda(), torch.from_numpy(pams).cuda()
    for cand in [None] + sys.argv[4:]:
        gidx.set_option("GS_ASTAR", cand)
        best = None
        for _ in range(3):
            _, _, st = gidx.enumerate_device(d_s.data_ptr(), batch, 20, d_p.data_ptr(), 3, mismatches=m)
            best = st["ms_search"] if best is None else min(best, st["ms_search"])
        print(json.dumps({"m": m, "astar": cand or "model", "k_search_ms": round(best, 2), "ms_total": round(st["ms_total"], 2),
                          "hits": st["n_hits"]}), flush=True)
    gidx.close()


if __name__ == "__main__":
    main()
