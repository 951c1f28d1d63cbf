#!/usr/bin/env python3
"""BASELINE config 4 in small: the genome-wide NGG candidate set, generated on the GPU chromosome
by chromosome (gs_kmers_generate) and enumerated in batches straight from HBM
(gs_enumerate_device), optionally scored (gs_score_device).  The candidates never visit the host.

    python tools/genomewide_enumerate.py [--workload chr1|hg38|saccer3] [--mismatches 3]
        [--batch 1000000] [--max-guides 4000000] [--score]

Multi-GPU: launch with torch.distributed.run; rank r takes the batches b with b % world == r of
every chromosome (independent units, no data-path collective), totals are summed at the end."""
import argparse
import json
import os
import sys
import time
from importlib import import_module
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="chr1")
    ap.add_argument("--mismatches", type=int, default=3)
    ap.add_argument("--batch", type=int, default=1_000_000)
    ap.add_argument("--max-guides", type=int, default=4_000_000, help="per rank; 0 = the whole candidate set")
    ap.add_argument("--score", action="store_true")
    a = ap.parse_args()
    import torch
    api = import_module("guidescan-cli_amd.api")
    synth = import_module("guidescan-cli_amd.synth")
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    lengths = {"chr1": [synth.CHR1_LENGTH], "hg38": synth.GRCH38_LENGTHS, "saccer3": synth.SACCER3_LENGTHS}[a.workload]
    text, names, lengths = synth.make_genome(lengths, seed=1)
    gidx = api.GenomeIndex.build(text, device=local)
    gs = api.make_genome_structure(names, lengths)
    n_guides = n_hits = n_cand = 0
    t_gen = t_enum = t_score = 0.0
    off = 0
    spec_sum = 0.0
    for name, ln in zip(names, lengths):
        if a.max_guides and n_guides >= a.max_guides:
            break
        d_chr = torch.from_numpy(text[off:off + ln]).cuda()
        off += ln
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        km = api.generate_kmers(None, "NGG", 20, device=local, chrm_device_ptr=d_chr.data_ptr(), chrm_len=ln)
        t_gen += time.perf_counter() - t0
        n_cand += km.n
        for b, lo in enumerate(range(0, km.n, a.batch)):
            if b % world != rank:
                continue
            if a.max_guides and n_guides >= a.max_guides:
                break
            n = min(a.batch, km.n - lo)
            t0 = time.perf_counter()
            d_off, d_hits, st = gidx.enumerate_device(km.seqs_ptr + lo * 20, n, 20, km.pams_ptr + lo * 3, 3,
                                                      mismatches=a.mismatches)
            t_enum += time.perf_counter() - t0
            n_guides += n
            n_hits += st["n_hits"]
            if a.score:
                d_spec = torch.empty(n, dtype=torch.float32, device="cuda")
                t0 = time.perf_counter()
                gidx.score_device(gs, km.seqs_ptr + lo * 20, n, 20, 3, d_off, d_hits, None, d_spec.data_ptr())
                t_score += time.perf_counter() - t0
                spec_sum += float(d_spec.sum().item())
        km.close()
        del d_chr
    tot = torch.tensor([n_guides, n_hits], dtype=torch.float64, device="cuda")
    tmax = torch.tensor([t_enum], dtype=torch.float64, device="cuda")
    if dist is not None:
        dist.all_reduce(tot)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    if rank == 0:
        g, h = float(tot[0]), float(tot[1])
        print(json.dumps({"workload": a.workload, "mismatches": a.mismatches, "n_gpus": world,
                          "candidates_scanned": n_cand, "guides_enumerated": int(g), "hits": int(h),
                          "kmers_generate_s": t_gen, "enumerate_s": float(tmax[0]),
                          "guides_per_s": g / float(tmax[0]) if float(tmax[0]) > 0 else None,
                          "score_s": t_score if a.score else None,
                          "mean_specificity": spec_sum / n_guides if a.score and n_guides else None}))
    gidx.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
