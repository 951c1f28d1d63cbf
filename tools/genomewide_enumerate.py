#!/usr/bin/env python3
"""BASELINE config 4 in small: the genome-wide NGG candidate set, generated on the GPU chromosome
by chromosome (gs_kmers_generate) and enumerated in batches straight from HBM
(gs_enumerate_device), optionally scored (gs_score_device).  The candidates never visit the host.

    python tools/genomewide_enumerate.py [--workload chr1|hg38|saccer3] [--mismatches 3]
        [--batch 1000000] [--max-guides 4000000] [--score]

Multi-GPU: launch with torch.distributed.run; rank r takes the batches b with b % world == r of
every chromosome (independent units, no data-path collective), totals are summed at the end.
GS_TOOLS_STUB=1 rehearses exactly that plumbing on gloo without a GPU (tools/_stub.py; not a measurement)."""
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
    ap.add_argument("--sorted", action="store_true",
                    help="collect every candidate on the device first and enumerate them in lexicographic order: guides "
                         "that run at the same time then read neighbouring table lines (single GPU)")
    a = ap.parse_args()
    import torch
    api = import_module("guidescan-cli_amd.api")
    synth = import_module("guidescan-cli_amd.synth")
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    stub = None
    if os.environ.get("GS_TOOLS_STUB") == "1":
        import _stub as stub
        if a.sorted:
            raise SystemExit("--sorted is a single-GPU mode: nothing to rehearse")
    dev = "cpu" if stub else "cuda"
    sync = (lambda: None) if stub else torch.cuda.synchronize
    if not stub:
        torch.cuda.set_device(local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if stub:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    lengths = {"chr1": [synth.CHR1_LENGTH], "hg38": synth.GRCH38_LENGTHS, "saccer3": synth.SACCER3_LENGTHS}[a.workload]
    text, names, lengths = synth.make_genome(lengths, seed=1)
    gidx = stub.StubIndex(text) if stub else api.GenomeIndex.build(text, device=local)
    gs = api.make_genome_structure(names, lengths)
    n_guides = n_hits = n_cand = 0
    t_gen = t_enum = t_score = 0.0
    off = 0
    spec_sum = 0.0
    if a.sorted:
        import ctypes as C
        hip = C.CDLL("libamdhip64.so")
        hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        parts_s, parts_p = [], []
        for name, ln in zip(names, lengths):
            d_chr = torch.from_numpy(text[off:off + ln]).cuda()
            off += ln
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            km = api.generate_kmers(None, "NGG", 20, device=local, chrm_device_ptr=d_chr.data_ptr(), chrm_len=ln)
            t_gen += time.perf_counter() - t0
            s = torch.empty((km.n, 20), dtype=torch.uint8, device="cuda")
            p = torch.empty((km.n, 3), dtype=torch.uint8, device="cuda")
            assert hip.hipMemcpy(s.data_ptr(), km.seqs_ptr, 20 * km.n, 3) == 0
            assert hip.hipMemcpy(p.data_ptr(), km.pams_ptr, 3 * km.n, 3) == 0
            parts_s.append(s)
            parts_p.append(p)
            n_cand += km.n
            km.close()
            del d_chr
        seqs = torch.cat(parts_s)
        pams = torch.cat(parts_p)
        del parts_s, parts_p
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        code = torch.zeros(256, dtype=torch.int64, device="cuda")
        for i, ch in enumerate(b"ACGT"):
            code[ch] = i
        keys = torch.zeros(seqs.shape[0], dtype=torch.int64, device="cuda")
        step = 1 << 24
        for lo in range(0, seqs.shape[0], step):
            c = code[seqs[lo:lo + step].long()]
            k = torch.zeros(c.shape[0], dtype=torch.int64, device="cuda")
            for i in range(20):
                k |= c[:, i] << (2 * (19 - i))
            keys[lo:lo + step] = k
            del c, k
        order = torch.argsort(keys)
        del keys
        seqs_s = torch.empty_like(seqs)
        pams_s = torch.empty_like(pams)
        for lo in range(0, order.shape[0], step):
            seqs_s[lo:lo + step] = seqs[order[lo:lo + step]]
            pams_s[lo:lo + step] = pams[order[lo:lo + step]]
        del seqs, pams, order
        torch.cuda.synchronize()
        t_sort = time.perf_counter() - t0
        total = seqs_s.shape[0] if not a.max_guides else min(a.max_guides, seqs_s.shape[0])
        for lo in range(0, total, a.batch):
            n = min(a.batch, total - lo)
            t0 = time.perf_counter()
            d_off, d_hits, st = gidx.enumerate_device(seqs_s.data_ptr() + lo * 20, n, 20, pams_s.data_ptr() + lo * 3, 3,
                                                      mismatches=a.mismatches)
            t_enum += time.perf_counter() - t0
            n_guides += n
            n_hits += st["n_hits"]
        print(json.dumps({"workload": a.workload, "mismatches": a.mismatches, "order": "lexicographic", "candidates_scanned": n_cand,
                          "guides_enumerated": n_guides, "hits": n_hits, "kmers_generate_s": t_gen, "sort_s": t_sort,
                          "enumerate_s": t_enum, "guides_per_s": n_guides / t_enum}))
        gidx.close()
        return
    for name, ln in zip(names, lengths):
        if a.max_guides and n_guides >= a.max_guides:
            break
        d_chr = torch.from_numpy(text[off:off + ln]).to(dev)
        sync()
        t0 = time.perf_counter()
        if stub:
            km = stub.generate_kmers(text[off:off + ln])
        else:
            km = api.generate_kmers(None, "NGG", 20, device=local, chrm_device_ptr=d_chr.data_ptr(), chrm_len=ln)
        off += ln
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
                d_spec = torch.empty(n, dtype=torch.float32, device=dev)
                t0 = time.perf_counter()
                gidx.score_device(gs, km.seqs_ptr + lo * 20, n, 20, 3, d_off, d_hits, None, d_spec.data_ptr())
                t_score += time.perf_counter() - t0
                spec_sum += float(d_spec.sum().item())
        km.close()
        del d_chr
    tot = torch.tensor([n_guides, n_hits, spec_sum], dtype=torch.float64, device=dev)
    tmax = torch.tensor([t_enum], dtype=torch.float64, device=dev)
    if dist is not None:
        dist.all_reduce(tot)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    if rank == 0:
        g, h = float(tot[0]), float(tot[1])
        print(json.dumps({"workload": a.workload, "mismatches": a.mismatches, "n_gpus": world, **({"stub": True} if stub else {}),
                          "candidates_scanned": n_cand, "guides_enumerated": int(g), "hits": int(h),
                          "kmers_generate_s": t_gen, "enumerate_s": float(tmax[0]),
                          "guides_per_s": g / float(tmax[0]) if float(tmax[0]) > 0 else None,
                          "score_s": t_score if a.score else None,
                          "mean_specificity": float(tot[2]) / g if a.score and g else None}))
    gidx.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
