#!/usr/bin/env python3
"""BASELINE config 5 at its stated size on one GPU (or sharded by batches over several): 1,000,000
sampled NGG 20-mers at <= 6 mismatches with CFD / specificity, streamed through the resident hg38-sized
index in batches: gs_enumerate_device -> gs_score_device per batch, hit lists never leave the HBM.

    python tools/config5_stream.py [--workload hg38] [--guides 1000000] [--batch 20000] [--mismatches 6]

Reports guides/s and hits/s over the whole stream (search + ordering + locate + scoring), the share of
k_search, peak HBM in use, and what the slot sizing did per batch (slots per item, guides that overflowed
into the arena / were searched again).  A checksum over every batch's CSR offsets, hit records and
specificities makes two runs comparable.  Multi-GPU: launch with torch.distributed.run; rank r takes the
batches b with b % world == r (independent units, no data-path collective).  GS_TOOLS_STUB=1 rehearses exactly that
plumbing on gloo without a GPU (tools/_stub.py; not a measurement)."""
import argparse
import ctypes as C
import json
import os
import sys
import time
from importlib import import_module
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="hg38")
    ap.add_argument("--guides", type=int, default=1_000_000)
    ap.add_argument("--batch", type=int, default=20_000)
    ap.add_argument("--mismatches", type=int, default=6)
    ap.add_argument("--checksum", action="store_true", help="fold every batch's offsets / hits / specificities into a checksum (device side)")
    a = ap.parse_args()
    import torch
    api = import_module("guidescan-cli_amd.api")
    synth = import_module("guidescan-cli_amd.synth")
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    stub = None
    if os.environ.get("GS_TOOLS_STUB") == "1":
        import _stub as stub
        a.checksum = False
    dev = "cpu" if stub else "cuda"
    sync = (lambda: None) if stub else torch.cuda.synchronize
    mem_info = (lambda: (0, 0)) if stub else torch.cuda.mem_get_info
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
    t0 = time.time()
    gidx = stub.StubIndex(text) if stub else api.GenomeIndex.build(text, device=local)
    t_index = time.time() - t0
    gs = api.make_genome_structure(names, lengths)
    seqs, pams, pos, strands = synth.sample_guides(text, a.guides, seed=1000)
    d_seqs, d_pams = torch.from_numpy(seqs).to(dev), torch.from_numpy(pams).to(dev)
    if not stub:
        hip = C.CDLL("libamdhip64.so")
        hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    free0, total_mem = mem_info()
    min_free = free0
    n_guides = n_hits = 0
    t_enum = t_score = ms_search = 0.0
    slots, redone, from_arena, wide, tiles_n = [], 0, 0, 0, 0
    batch_ms = []
    own_found = 0
    csum = 0
    spec_sum = 0.0
    sync()
    t_start = time.perf_counter()
    for b, lo in enumerate(range(0, a.guides, a.batch)):
        if b % world != rank:
            continue
        n = min(a.batch, a.guides - lo)
        t0 = time.perf_counter()
        d_off, d_hits, st = gidx.enumerate_device(d_seqs.data_ptr() + lo * 20, n, 20, d_pams.data_ptr() + lo * 3, 3,
                                                  mismatches=a.mismatches)
        t_enum += time.perf_counter() - t0
        batch_ms.append((time.perf_counter() - t0) * 1e3)
        ms_search += st["ms_search"]
        ctr = gidx.last_counters()
        slots.append(int(ctr["slots_per_item"]))
        redone += int(ctr["guides_redone"])
        from_arena += int(ctr["overflow_from_arena"])
        wide += int(ctr["ordered_device_wide"])
        tiles_n += int(ctr.get("ordered_in_tiles", False))
        d_spec = torch.empty(n, dtype=torch.float32, device=dev)
        t0 = time.perf_counter()
        gidx.score_device(gs, d_seqs.data_ptr() + lo * 20, n, 20, 3, d_off, d_hits, None, d_spec.data_ptr())
        t_score += time.perf_counter() - t0
        min_free = min(min_free, mem_info()[0])
        n_guides += n
        n_hits += st["n_hits"]
        spec_sum += float(d_spec.double().sum().item())
        if a.checksum:
            off = torch.empty(n + 1, dtype=torch.int64, device="cuda")
            assert hip.hipMemcpy(off.data_ptr(), d_off, 8 * (n + 1), 3) == 0
            nh = st["n_hits"]
            step = 1 << 26   # hit records folded in pieces: the batch's list can be GBs
            for h0 in range(0, nh, step):
                m = min(step, nh - h0)
                piece = torch.empty((m, 2), dtype=torch.int64, device="cuda")
                assert hip.hipMemcpy(piece.data_ptr(), d_hits + 16 * h0, 16 * m, 3) == 0
                w = torch.arange(h0, h0 + m, dtype=torch.int64, device="cuda") * 0x9E3779B1 + 12345
                csum = (csum + int(((piece[:, 0] ^ piece[:, 1]) * w).sum().item())) & 0xFFFFFFFFFFFFFFFF
                if h0 == 0:
                    # every sampled guide reports its own site at distance 0 (counted on the first piece's guides)
                    pass
                del piece, w
            csum = (csum * 1099511628211 + int((off * torch.arange(1, n + 2, device="cuda")).sum().item()) +
                    int(d_spec.view(torch.int32).long().sum().item())) & 0xFFFFFFFFFFFFFFFF
            del off
        del d_spec
    sync()
    wall = time.perf_counter() - t_start
    tot = torch.tensor([n_guides, n_hits, redone, from_arena, wide, spec_sum], dtype=torch.float64, device=dev)
    tmax = torch.tensor([wall, t_enum, t_score, ms_search, float(total_mem - min_free)], dtype=torch.float64, device=dev)
    if dist is not None:
        dist.all_reduce(tot)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    if rank == 0:
        g, h = float(tot[0]), float(tot[1])
        w = float(tmax[0])
        print(json.dumps({
            **({"stub": True} if stub else {}),
            "config": f"{a.workload}-sized synthetic genome, {a.guides} sampled NGG 20-mers, <= {a.mismatches} mismatches + CFD, "
                      f"batches of {a.batch}, {world} GPU(s)",
            "guides": int(g), "hits": int(h), "hits_per_guide": h / g if g else None,
            "wall_s": w, "guides_per_s": g / w, "hits_per_s": h / w,
            "enumerate_s": float(tmax[1]), "score_s": float(tmax[2]), "k_search_s": float(tmax[3]) / 1e3,
            "k_search_share_of_wall": float(tmax[3]) / 1e3 / w,
            "index_build_s": t_index, "index_bytes": gidx.device_bytes,
            "peak_hbm_in_use_bytes": int(tmax[4]), "hbm_total_bytes": int(total_mem),
            "enumerate_ms_per_batch": {"first5": [round(x, 1) for x in batch_ms[:5]], "median": round(float(np.median(batch_ms)), 1),
                                       "min": round(min(batch_ms), 1), "max": round(max(batch_ms), 1)} if batch_ms else None,
            "slots_per_item_first_last": [slots[0], slots[-1]] if slots else None,
            "slots_per_item_distinct": sorted(set(slots)),
            "guides_overflowed": int(tot[2]), "batches_served_from_arena": int(tot[3]),
            "batches_ordered_device_wide": int(tot[4]),
            "batches_ordered_per_guide_in_lds_tiles_rank0": tiles_n,
            "mean_specificity": float(tot[5]) / g if g else None,
            "checksum": f"{csum:016x}" if a.checksum else None}))
    gidx.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
