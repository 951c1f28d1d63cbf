#!/usr/bin/env python3
"""The repeat-rich row of bench.py's detail.extra_rows by itself, step by step: 20,000 fresh guides per step on the
hg38rep-sized index (every step's hit lists, buckets and tiles differ in size).  GS_DEBUG=1 shows the workspace buffers
that grow.  Usage (GPU box, repo root): python tools/rep_steps.py [steps]"""
import sys
import time
from importlib import import_module
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def main():
    import torch
    bench = import_module("bench")
    api = import_module("guidescan-cli_amd.api")
    synth = import_module("guidescan-cli_amd.synth")
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    lens_name, _, probs = bench.WORKLOADS["hg38rep"]
    lengths = getattr(synth, lens_name)
    text, names, lengths = bench.make_workload_genome(synth, "hg38rep", lengths, probs)
    g = api.GenomeIndex.build(text, device=0)
    try:
        seqs, pams, _, _ = synth.sample_guides(text, 20000 * steps, seed=1000)
        d_s, d_p = torch.from_numpy(seqs).cuda(), torch.from_numpy(pams).cuda()
        for i in range(steps):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            _, _, st = g.enumerate_device(d_s[i * 20000:].data_ptr(), 20000, 20, d_p[i * 20000:].data_ptr(), 3, mismatches=3)
            torch.cuda.synchronize()
            print(f"step {i}: {1e3 * (time.perf_counter() - t0):.1f} ms, k_search {st['ms_search']:.1f} ms, {st['n_hits']} hits, "
                  f"free {torch.cuda.mem_get_info()[0] / 2**30:.1f} GiB", flush=True)
    finally:
        g.close()


if __name__ == "__main__":
    main()
