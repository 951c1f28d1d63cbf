#!/usr/bin/env python3
"""CFD + specificity (gs_score_device) of one enumerated batch, timed by itself: k_score_hits + k_score_sum on the hits
gs_enumerate_device left in HBM.  Usage (GPU box, repo root): python tools/score_bench.py [workload=hg38rep] [guides=20000] [m=3] [reps=4]"""
import json
import sys
import time
import zlib
from importlib import import_module
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def main():
    import numpy as np
    import torch
    bench = import_module("bench")
    api = import_module("guidescan-cli_amd.api")
    synth = import_module("guidescan-cli_amd.synth")
    workload = sys.argv[1] if len(sys.argv) > 1 else "hg38rep"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
    m = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    reps = int(sys.argv[4]) if len(sys.argv) > 4 else 4
    lens_name, _, probs = bench.WORKLOADS[workload]
    lengths = [synth.CHR1_LENGTH] if lens_name == "CHR1" else getattr(synth, lens_name)
    text, names, lengths = bench.make_workload_genome(synth, workload, lengths, probs)
    g = api.GenomeIndex.build(text, device=0)
    try:
        seqs, pams, _, _ = synth.sample_guides(text, n, seed=1000)
        d_s, d_p = torch.from_numpy(seqs).cuda(), torch.from_numpy(pams).cuda()
        gs = api.make_genome_structure(names, lengths)
        d_off, d_hits, st = g.enumerate_device(d_s.data_ptr(), n, 20, d_p.data_ptr(), 3, mismatches=m)
        cfd = torch.empty(st["n_hits"] + 1, dtype=torch.float32, device="cuda")
        spec = torch.empty(n, dtype=torch.float32, device="cuda")
        ts = []
        for _ in range(reps):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            g.score_device(gs, d_s.data_ptr(), n, 20, 3, d_off, d_hits, cfd.data_ptr(), spec.data_ptr())
            torch.cuda.synchronize()
            ts.append(1e3 * (time.perf_counter() - t0))
        sp = spec.cpu().numpy()
        print(json.dumps({"workload": workload, "guides": n, "m": m, "hits": st["n_hits"], "score_ms": [round(t, 2) for t in ts],
                          "specificity_crc32": f"{zlib.crc32(sp.tobytes()):08x}",
                          "cfd_crc32": f"{zlib.crc32(cfd[:st['n_hits']].cpu().numpy().tobytes()):08x}"}), flush=True)
    finally:
        g.close()


if __name__ == "__main__":
    main()
