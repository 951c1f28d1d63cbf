#!/usr/bin/env python3
"""bench.py -- guides/sec of off-target enumeration on MI355X (BASELINE.json metric).

A "step" is one pass of the hot path (prepare -> search on both strand indexes ->
canonical order -> locate) over one batch of synthetic NGG 20-mer guides that is
already resident in HBM; the CSR hit lists are left in HBM.  Index construction is
outside the timed region (the reference's own timer also starts after index load,
src/guidescan.cxx:239).

    python bench.py --gpus N --steps K --warmup W [--workload chr1|hg38|saccer3]

N>1: launched by torch.distributed.run, one rank per GPU; the index is replicated
into every GPU's HBM and each rank enumerates its own shard of the guide batch
(no data-path collective; weak scaling: per-GPU batch fixed).
"""
import argparse
import json
import os
import sys
import subprocess
import time
from importlib import import_module
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
for p in (str(ROOT), str(ROOT / "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)

WORKLOADS = {
    # name: (chromosome lengths, guides per step per GPU, base probabilities)
    "saccer3": ("SACCER3_LENGTHS", 1000, (0.31, 0.19, 0.19, 0.31)),
    "chr1": ("CHR1", 100_000, (0.29, 0.21, 0.21, 0.29)),
    "hg38": ("GRCH38_LENGTHS", 1_000_000, (0.29, 0.21, 0.21, 0.29)),
    # the same sizes with 45 % of the bases overwritten by repeat families (synth.plant_repeats:
    # ~1e6 SINE-like 300-mers at 10-15 %, ~1.6e5 LINE-like copies at 5 %, tandem arrays, 100-kb
    # segmental duplications at 1 %, both strands); guides are sampled uniformly, so ~45 % of them
    # lie inside repeats and have 10^3..10^5 hits: small batches
    "chr1rep": ("CHR1", 20_000, (0.29, 0.21, 0.21, 0.29)),
    "hg38rep": ("GRCH38_LENGTHS", 20_000, (0.29, 0.21, 0.21, 0.29)),
    # ... with an Alu-like SINE family: 1.2e6 copies at 2-15 % from the unit (the young ones nearly identical): a guide
    # drawn from it has several 10^5 sites within three mismatches
    "hg38alu": ("GRCH38_LENGTHS", 20_000, (0.29, 0.21, 0.21, 0.29)),
}


def make_workload_genome(synth, workload, lengths, probs, out=None):
    if workload.endswith("alu"):
        return synth.make_repeat_genome(lengths, seed=1, probs=probs, out=out, sine_div=(0.02, 0.15), sine_copies=1_200_000)
    if workload.endswith("rep"):
        return synth.make_repeat_genome(lengths, seed=1, probs=probs, out=out)
    return synth.make_genome(lengths, seed=1, probs=probs, out=out)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default=os.environ.get("GS_BENCH_WORKLOAD", "hg38"))
    ap.add_argument("--mismatches", type=int, default=3)
    ap.add_argument("--batch", type=int, default=0, help="guides per step per GPU (0 = workload default)")
    ap.add_argument("--cpu-sample", type=int, default=-1, help="guides timed on the CPU oracle (0 = skip)")
    ap.add_argument("--cpu-kind", choices=["auto", "port", "reference"], default="auto",
                    help="cpu_baseline: the reference itself (oracle/_ref/gs_ref_enumerate, compiled "
                         "from the reference's sources; its SDSL index files are written first, about "
                         "75 s at hg38 size) or the oracle port; auto = reference when oracle/_ref "
                         "was built, else port")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="weak: --batch guides per GPU per step (the default); strong: --batch guides per step in all, "
                         "split over the ranks in contiguous shards (configs 4/5: a fixed guide set over 8 GPUs)")
    ap.add_argument("--deal", type=int, default=0,
                    help="with --scaling strong: the step's guides are handed out in chunks of this many from one shared counter "
                         "(the rendezvous store's atomic add; guidescan-cli_amd/parallel.py) instead of one contiguous shard per "
                         "rank - a rank that draws repeat-dense chunks draws fewer (src/guidescan.cxx:226-231 deals round-robin for "
                         "that reason); 0 = contiguous shards")
    ap.add_argument("--stream", type=int, default=0,
                    help="a step goes through its guides in sub-batches of this many (config 5: 1 M guides at <= 6 "
                         "mismatches as 50 batches of 20 k); 0 = one call per step")
    ap.add_argument("--score", action="store_true", help="CFD + specificity of every hit inside the timed step (config 5)")
    ap.add_argument("--cpu-threads", default="32,64,128,256",
                    help="thread counts the reference CPU baseline is timed at (the best is reported)")
    ap.add_argument("--hwpopcnt", action="store_true", help="also time the reference built with hardware POPCNT")
    ap.add_argument("--extra-rows", choices=["auto", "on", "off", "e2e"], default="auto",
                    help="after the headline: the repeat-rich genome at <= 3 mismatches and <= 6 mismatches + CFD on this "
                         "genome, 20 k guides each (auto: with the default hg38 workload on one GPU; e2e: the end-to-end row alone)")
    ap.add_argument("--e2e-guides", type=int, default=1_000_000,
                    help="guides of the end-to-end row among the extra rows (the built `guidescan enumerate`, kmers CSV -> CSV; 0 = skip)")
    ap.add_argument("--n-gaps", type=int, default=0,
                    help="overwrite this many random stretches of the genome with N runs (a scaffold-level assembly: "
                         "three literal-N windows per run and strand for k_search to look through)")
    ap.add_argument("--verify", action="store_true",
                    help="after timing, check full-size properties of the last batch (on-target found, order)")
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ:
        # `python bench.py --gpus N` by itself: this process never touches the GPU; it starts the
        # N ranks as children (one process per GPU under torch.distributed.run) and relays rank
        # 0's JSON line
        raise SystemExit(launch_ranks(args))
    if os.environ.get("GS_BENCH_STUB"):
        raise SystemExit(stub_main(args))

    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the enumerate path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or "RANK" in os.environ:
        # one process per GPU under torch.distributed.run; backend "nccl" is RCCL on ROCm.  Used
        # for rendezvous, the barriers around the timed region and the MAX-reduce only.
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    api = import_module("guidescan-cli_amd.api")
    synth = import_module("guidescan-cli_amd.synth")

    lens_name, batch, probs = WORKLOADS[args.workload]
    lengths = [synth.CHR1_LENGTH] if lens_name == "CHR1" else getattr(synth, lens_name)
    if args.batch:
        batch = args.batch
    m = args.mismatches
    t0 = time.time()
    text, names, lengths = shared_genome(synth, args.workload, lengths, probs, dist, local_rank)
    if args.n_gaps:
        text = np.array(text, copy=True)
        rng = np.random.Generator(np.random.PCG64(4321))
        for at in np.sort(rng.integers(1_000_000, text.shape[0] - 1_000_000, size=args.n_gaps)):
            text[at:at + int(rng.integers(100, 50_000))] = ord("N")
    t_gen = time.time() - t0
    t0 = time.time()
    gidx = api.GenomeIndex.build(text, device=local_rank)
    torch.cuda.synchronize()
    t_index = time.time() - t0
    # HBM left on this rank's GPU next to the index (RCCL's buffers exist by now: the rendezvous and the
    # barriers of shared_genome ran before the build), again after the timed steps (PAM-pair tables + workspace)
    free_after_index = int(torch.cuda.mem_get_info()[0])

    nb = args.steps + args.warmup
    parallel = import_module("guidescan-cli_amd.parallel")
    batch_all = batch * world   # guides per step over all ranks
    if args.scaling == "strong":
        # one seeded guide set per step for the whole job, rank r takes the contiguous shard r of every step
        # (src/guidescan.cxx:226-251 deals guides to threads; here to GPUs, each with the whole index)
        batch_all = batch
        all_seqs, all_pams, _, _ = synth.sample_guides(text, batch_all * nb, seed=1000)
        if args.deal:
            seqs, pams = all_seqs, all_pams   # every rank holds the step's guides and enumerates the chunks it draws
        else:
            b = parallel.shard_bounds(batch_all, world)
            lo, hi = b[rank], b[rank + 1]
            batch = hi - lo
            pick = np.concatenate([np.arange(i * batch_all + lo, i * batch_all + hi) for i in range(nb)])
            seqs, pams = np.ascontiguousarray(all_seqs[pick]), np.ascontiguousarray(all_pams[pick])
        del all_seqs, all_pams
    else:
        # every step and every rank gets its own guides (seeded): shard r of the global batch
        seqs, pams, _, _ = synth.sample_guides(text, batch * nb, seed=1000 + rank)
    d_seqs = torch.from_numpy(seqs).cuda()
    d_pams = torch.from_numpy(pams).cuda()
    L, P = seqs.shape[1], pams.shape[1]
    gs_struct = api.make_genome_structure(names, lengths) if args.score else None
    score_buf = {}

    def one_call(s, p, n):
        d_off, d_hits, st = gidx.enumerate_device(s.data_ptr(), n, L, p.data_ptr(), P, mismatches=m)
        if args.score:
            # CFD of every hit + specificity per guide, in HBM (printer.hpp:98-170): part of the step for config 5
            if score_buf.get("n", 0) < st["n_hits"] or score_buf.get("g", 0) < n:
                score_buf.update(n=int(st["n_hits"] * 1.25) + 1, g=n)
                score_buf["cfd"] = torch.empty(score_buf["n"], dtype=torch.float32, device="cuda")
                score_buf["spec"] = torch.empty(n, dtype=torch.float32, device="cuda")
            gidx.score_device(gs_struct, s.data_ptr(), n, L, P, d_off, d_hits, score_buf["cfd"].data_ptr(),
                              score_buf["spec"].data_ptr())
        return d_off, d_hits, st

    busy = {"s": 0.0, "chunks": 0}

    def step(i):
        s = d_seqs[i * batch:(i + 1) * batch]
        p = d_pams[i * batch:(i + 1) * batch]
        if args.scaling == "strong" and args.deal:
            # chunks of the step's guides from the shared counter; what the step returns is the sum over this rank's chunks
            acc = {"st": None, "last": (None, None)}

            def one(c, lo, hi):
                d_off, d_hits, st = one_call(s[lo:hi], p[lo:hi], hi - lo)
                acc["st"] = st if acc["st"] is None else {k: acc["st"][k] + st[k] for k in st}
                acc["last"] = (d_off, d_hits)
            mine, sec = parallel.deal_chunks(one, batch, args.deal, f"step{i}", dist)
            busy["s"] += sec
            busy["chunks"] += len(mine)
            zero = {"n_ext": 0, "n_matches": 0, "n_hits": 0, "ms_search": 0.0, "ms_total": 0.0}
            return acc["last"][0], acc["last"][1], acc["st"] or zero
        if not args.stream or args.stream >= batch:
            return one_call(s, p, batch)
        tot = None
        for j in range(0, batch, args.stream):   # the step's guides as a stream of sub-batches
            n = min(args.stream, batch - j)
            d_off, d_hits, st = one_call(s[j:j + n], p[j:j + n], n)
            tot = st if tot is None else {k: tot[k] + st[k] for k in st}
        return d_off, d_hits, tot

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    # N_ext is a property of the input under the REFERENCE's traversal
    # rules.  The timed path skips the input-independent top of the tree through the prefix
    # table, so it is measured here, untimed, with the reference-order walk on a sample (sized so
    # that the walk stays around a second: it is ~10^5 extensions per guide at m = 3, 3.5e7 at m = 6).
    ns = min(batch, args.stream or batch, {0: 20000, 1: 20000, 2: 20000, 3: 20000, 4: 4000}.get(m, 0))
    if ns:
        _, _, st_ref = gidx.enumerate_device(d_seqs.data_ptr(), ns, L, d_pams.data_ptr(), P, mismatches=m,
                                             faithful=True)
        n_ext_ref_per_guide = st_ref["n_ext"] / ns
    else:
        # m >= 5: one item of the walk is 10^7 extensions on ONE wavefront (16 s at m = 6 whatever the
        # sample); SURVEY App. C's closed form, within 7 % of the measured walk at three genome sizes
        from math import comb
        n_rows = float(text.shape[0])
        n_ext_ref_per_guide = 2.0 * sum(min(1.0, n_rows / 4.0 ** d) * sum(comb(d, k) * 3 ** k for k in range(min(m, d) + 1))
                                        for d in range(L + 1))
    # The bytes THIS algorithm asks the memory system for: one untimed pass of the first batch
    # through the counting instantiation of k_search (same code, plus a tally of the distinct
    # 64-byte lines every load instruction requests; include/guidescan_amd.h GS_FLAG_COUNT_REQUESTS)
    n_cnt = min(batch, args.stream) if args.stream else batch   # guides of one launch
    launches_per_step = (batch + n_cnt - 1) // n_cnt if n_cnt else 1
    _, _, st_cnt = gidx.enumerate_device(d_seqs.data_ptr(), n_cnt, L, d_pams.data_ptr(), P, mismatches=m,
                                         count_requests=True)
    req = gidx.last_counters()
    # the same pass once more tallying 128-byte blocks: what the memory system serves as ONE random request (a 128-byte
    # block read by one instruction costs what a 64-byte one does: tools/gather_bench, profiles/r04_gather_calibration_groups.txt)
    gidx.set_option("GS_COUNT_SHIFT", "7")
    try:
        gidx.enumerate_device(d_seqs.data_ptr(), n_cnt, L, d_pams.data_ptr(), P, mismatches=m, count_requests=True)
        req128 = gidx.last_counters()
    finally:
        gidx.set_option("GS_COUNT_SHIFT", None)
    n_req128 = sum(req128[k] for k in ("table_lines", "ctx16_lines", "ctx_words", "sa_isa_gathers", "occ_lines"))

    for i in range(args.warmup):
        step(i)
    fence()
    busy["s"] = 0.0
    busy["chunks"] = 0
    t0 = time.perf_counter()
    n_ext = n_hits = 0
    ms_search = ms_total = 0.0
    per_step_ms = []
    for i in range(args.warmup, nb):
        _, _, st = step(i)
        per_step_ms.append(round(st["ms_search"], 1))
        n_ext += st["n_ext"]
        n_hits += st["n_hits"]
        ms_search += st["ms_search"]
        ms_total += st["ms_total"]
    torch.cuda.synchronize()
    own_elapsed = time.perf_counter() - t0   # this rank's own steps, before it waits for the others
    fence()
    elapsed = time.perf_counter() - t0
    # every rank's own time per step and what the slowest adds to the job ((max - mean) / max): the MAX below is `value`'s clock
    rank_s = parallel.gather_floats(own_elapsed, dist)
    rank_chunks = parallel.gather_floats(float(busy["chunks"]), dist)
    if dist is not None and args.scaling == "strong":   # hits are per rank in a split job: the job's total
        th = torch.tensor([float(n_hits)], dtype=torch.float64, device="cuda")
        dist.all_reduce(th, op=dist.ReduceOp.SUM)
        n_hits_job = float(th.item())
    else:
        n_hits_job = float(n_hits) * world
    last_ctr = gidx.last_counters()   # of the last timed call (the flags of the counting pass are in `req`)
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    mem = {"rank": rank, "index_build_s": round(t_index, 1), "free_after_index_bytes": free_after_index,
           "index_bytes_after_steps": int(gidx.device_bytes), "free_after_steps_bytes": int(torch.cuda.mem_get_info()[0]),
           "hbm_total_bytes": int(torch.cuda.mem_get_info()[1])}
    per_rank = [mem]
    if dist is not None:
        try:
            gathered = [None] * world
            dist.all_gather_object(gathered, mem)
            per_rank = gathered
        except Exception as e:  # a side figure: never lose the bench line to it
            print(f"[bench] per-rank memory report not gathered: {e!r}", file=sys.stderr)
    K = args.steps
    guides_total = batch_all * K
    value = guides_total / elapsed

    # Roofline of the dominant kernel (k_search), HBM bound.  Algorithmic bytes of the algorithm that
    # runs: every load k_search issues is a random access that moves one 64-byte line (Occ block,
    # table line of four entries, eight 16-bit context words, one context word, one SA/ISA entry), so
    # B = 64 B x distinct lines requested (counted per load instruction by the counting pass above)
    # + 16 B per match record written.  achieved = B per launch / the launch's HIP-event duration
    # (measured inside the library on the launch stream), against the 8 TB/s HBM3E peak.
    # `traffic` is what the memory side actually moved (FETCH_SIZE + WRITE_SIZE from separate
    # rocprofv3 --pmc passes of this command, profiles/traffic.json) - only when that file was
    # recorded with these kernel sources; more than B means re-reads, less means cache hits.
    lines = {k: req[k] for k in ("table_lines", "ctx16_lines", "ctx_words", "sa_isa_gathers", "occ_lines", "recipe_lines")}
    # the recipe lists (a few KB, the same for every item) stay in the L2: their lines are reported, not priced
    n_lines = sum(v for k, v in lines.items() if k != "recipe_lines")
    alg_bytes_per_launch = 64.0 * n_lines + 16.0 * st_cnt["n_matches"]
    search_s = (ms_search / (K * launches_per_step)) / 1e3   # one launch
    achieved = alg_bytes_per_launch / search_s / 1e9 if search_s > 0 else 0.0
    traffic, traffic_src, issue = recorded_traffic(args.workload, n_cnt, m)
    if issue:
        # what binds the kernel when the bytes do not: VALU wave-instructions of the same dispatch (SQ_INSTS_VALU
        # of a separate --pmc pass) against the SIMDs' issue slots - 4 cycles per wave64 instruction, 1024 SIMDs, 2.4 GHz
        issue = dict(issue, valu_frac_of_issue_cycles=issue["valu_wave_instructions"] * 4.0 /
                     (1024 * 2.4e9 * issue["duration_ms"] * 1e-3), source=traffic_src.replace("fetch_size", "sq_wave_cycles").split(" + ")[0])
    ref_bytes = 128.0 * n_ext_ref_per_guide * n_cnt
    out = {
        "metric": f"guides/sec off-target enum, <={m} mismatches",
        "value": value,
        "unit": "guides/s",
        # the same timed region in hits: what a repeat-rich genome is priced in (guides/s falls with the hits per guide)
        "hits_per_s": n_hits_job / elapsed,
        "per_rank_ms_per_step": [round(x / args.steps * 1e3, 3) for x in rank_s],
        "rank_imbalance": parallel.imbalance(rank_s),
        "dealt_in_chunks_of": args.deal if (args.scaling == "strong" and args.deal) else None,
        "chunks_per_rank": [int(x) for x in rank_chunks] if (args.scaling == "strong" and args.deal) else None,
        "n_gpus": world,
        "steps": K,
        "warmup": args.warmup,
        "ms_per_step": elapsed / K * 1e3,
        "higher_is_better": True,
        "scaling": args.scaling,
        "vs_baseline": None,
        "dtype": "u32",
        "data": "synthetic",
        "config": {"workload": f"{args.workload}-sized synthetic genome ({sum(lengths)} bp, "
                               f"{len(lengths)} chr, fwd+rev index in HBM), "
                               + (f"{batch_all} NGG 20-mers per step split over {world} GPU(s)" if args.scaling == "strong"
                                  else f"{batch} NGG 20-mers per GPU per step")
                               + f", <={m} mismatches"
                               + (f", streamed in sub-batches of {args.stream}" if args.stream else "")
                               + (", CFD + specificity in the step" if args.score else ""),
                   "guides_per_step_per_gpu": batch, "guides_per_step": batch_all, "mismatches": m,
                   "parallelism": f"replicated index, guide batch sharded x{world}"},
        "roofline": {"bound": "hbm", "kernel": search_kernel_name(gidx), "achieved": achieved, "peak": HBM_PEAK_GBS,
                     # achieved / frac price the lines each load instruction asks for; on a repeat-rich batch
                     # neighbouring hits share lines that L2 serves, so the figure can pass the HBM peak -
                     # `traffic` (PMC) is then the measure of what HBM delivered
                     "frac_note": ("counted lines exceed what HBM delivered: many are served by L2 (see traffic)"
                                   if achieved > HBM_PEAK_GBS * 0.8 else None),
                     "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                     # PMC counters cannot be read from inside the process that is being timed: `traffic` is
                     # the recorded figure of a separate rocprofv3 --pmc pass of this command with these
                     # kernel sources (hash checked), or null
                     "traffic_measured_in_this_run": False,
                     "traffic_source": traffic_src, "instruction_issue": issue,
                     # `traffic` is FETCH_SIZE + WRITE_SIZE, and FETCH_SIZE tallies every read request as 64 bytes: the
                     # requests themselves, and the bytes they brought (a PAM-pair block is one request of 128)
                     **memory_side_fields(recorded_memory_side(args.workload, n_cnt, m), ms_search / (K * launches_per_step)),
                     "request_rate_calibration": REQUEST_CALIBRATION,
                     "alg_bytes_per_launch": alg_bytes_per_launch,
                     "avg_launch_ms": ms_search / (K * launches_per_step),
                     "guides_per_launch": n_cnt,
                     "requests_per_guide": {k: v / n_cnt for k, v in lines.items()},
                     "random_requests": {
                         # requests as the memory system counts them: distinct 128-byte aligned blocks per load
                         # instruction (a PAM-pair block of sixteen 8-byte entries is ONE, not two lines)
                         "per_guide": n_req128 / n_cnt, "unit": "128-byte aligned blocks per load instruction",
                         "achieved_per_s": n_req128 / search_s if search_s > 0 else None,
                         # (counted blocks incl. what the L2s serve; the calibration rates are for independent L2-MISSING blocks:
                         # no fraction of them is a roofline - round 5's run at k = 13 passed the HBM figure)
                         "calibration_per_s": [4.8e10, 5.7e10],
                         # the same from the memory side (blocks the L2 serves are not in it): HBM lines of the recorded
                         # PMC pass against the 64-byte lines per second a pure 128-byte-block gather reaches (9.6e10)
                         "hbm_side": ({"lines_per_s": traffic / 64.0 / search_s} if traffic and search_s > 0 else None),
                         "lines_64_per_guide": n_lines / n_cnt,
                         "by_kind_per_guide": {k: req128[k] / n_cnt for k in ("table_lines", "ctx16_lines", "ctx_words",
                                                                               "sa_isa_gathers", "occ_lines")},
                         "note": "independent random blocks that miss the L2 are served at ~4.8e10 per second from HBM and 5.7e10 "
                                 "from the Infinity Cache whether they are 16, 64 or 128 bytes (tools/gather_calib): a calibration "
                                 "point, not a roof - a kernel's own rate depends on the chains its waves wait in and on the mix"},
                     # SURVEY 8d's figure, kept for comparison: the bytes the REFERENCE'S traversal
                     # (128 B per extended node, N_ext from the reference-order walk on a sample) would
                     # need for this batch.  Not what this kernel does: table, context mask, context
                     # arrays and two-sided seeding skip that work.
                     "reference_traversal": {"alg_bytes_per_launch": ref_bytes,
                                             "n_ext_per_guide": n_ext_ref_per_guide,
                                             "n_ext_source": f"reference-order walk on {ns} guides" if ns
                                             else "closed form of SURVEY App. C",
                                             "bytes_vs_this_algorithm": ref_bytes / alg_bytes_per_launch
                                             if alg_bytes_per_launch else None}},
        "detail": {"executed_ext_per_guide": n_ext / (batch * K), "hits_per_guide": n_hits / (batch * K),
                   "prefix_table_k": os.environ.get("GS_PREFIX_K", "auto"),
                   "k_search_ms_per_step": per_step_ms,
                   "k_search_ms_min_median_max": spread(per_step_ms),
                   "clocks": gpu_clocks(),
                   "device_ms_total_per_step": ms_total / K, "index_build_s": t_index,
                   "genome_gen_s": t_gen, "index_bytes": gidx.device_bytes,
                   "items_two_sided": req["items_two_sided"], "items_one_sided": req["items_one_sided"],
                   "overflow_items_first_pass": req["overflow_items"], "guides_redone": last_ctr["guides_redone"],
                   "slots_per_item": req["slots_per_item"], "matches_max_per_item": req["matches_max_per_item"],
                   "ordered_device_wide": last_ctr["ordered_device_wide"],
                   "redo_ordered_device_wide": last_ctr["redo_ordered_device_wide"],
                   "overflow_from_arena": last_ctr["overflow_from_arena"],
                   "ordered_by_one_composite_sort": last_ctr["ordered_by_one_composite_sort"],
                   "runs_put_right_after_the_sort": last_ctr["runs_turned_round"],
                   "ordered_per_guide_in_lds_tiles": last_ctr["ordered_in_tiles"],
                   "tile_ordering_gave_up": last_ctr["tile_ordering_gave_up"], "per_rank_memory": per_rank},
    }

    if rank == 0:
        # the device steps either side of the path (untimed side figures, never part of `value`):
        # CFD/specificity of the last batch's hits and the candidate-guide scan of chromosome 1
        try:
            out["detail"]["next_rows"] = side_steps(torch, api, gidx, d_seqs, d_pams, batch, nb - 1, L, P, m,
                                                    text, names, lengths)
        except Exception as e:
            print(f"[bench] side steps failed: {e!r}", file=sys.stderr)
    if args.verify:
        out["verify"] = verify_last_batch(torch, gidx, d_seqs, d_pams, batch, nb - 1, L, P, m, text, seqs)
    if rank == 0 and world == 1 and args.cpu_sample != 0:
        kind = args.cpu_kind
        if kind == "auto":
            kind = "reference" if (ROOT / "oracle" / "_ref" / "gs_ref_enumerate").exists() else "port"
        if kind == "reference":
            try:
                out["cpu_baseline"] = cpu_baseline_reference(text, names, lengths, gidx, seqs, pams, m,
                                                             args.cpu_sample, args.cpu_threads, args.hwpopcnt)
                if "hwpopcnt" in out["cpu_baseline"]:   # BASELINE.md section 3's second reference row
                    out["cpu_baseline_hwpopcnt"] = out["cpu_baseline"].pop("hwpopcnt")
            except Exception as e:  # the baseline is a side figure: never lose the bench line to it
                if args.cpu_kind == "reference":
                    raise
                print(f"[bench] reference baseline failed ({e!r}); timing the oracle port instead",
                      file=sys.stderr)
                kind = "port"
        if kind == "port":
            out["cpu_baseline"] = cpu_baseline(text, gidx, seqs, pams, m, args.cpu_sample)
    want_rows = args.extra_rows in ("on", "e2e") or (args.extra_rows == "auto" and args.workload == "hg38" and m == 3 and world == 1
                                            and not args.batch and not args.stream and args.cpu_sample != 0)
    if rank == 0 and world == 1 and want_rows:
        # the rows the headline does not show, under the same clock: the deep budget on this genome and the
        # repeat-rich genome (24 k hits per guide: what real hg38's repeat half looks like) - never part of `value`
        try:
            out["detail"]["extra_rows"], gidx = extra_rows(torch, api, synth, gidx, text, names, lengths, probs, L, P,
                                                           e2e_guides=args.e2e_guides, only_e2e=args.extra_rows == "e2e")
        except Exception as e:
            print(f"[bench] extra rows failed: {e!r}", file=sys.stderr)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if gidx is not None:
        gidx.close()
    if dist is not None:
        dist.destroy_process_group()


def search_kernel_name(gidx):
    """what ran between the library's search events in the last call: the one-launch k_search, or - a batch whose every
    pattern has its PAM-pair + deep tables and that shares no item - k_describe + k_sched_* + k_seed_b + k_seed_a"""
    form = gidx.last_sharing()["form"]
    return {0: "k_search", 1: "k_search (heavy instantiation)", 2: "k_search (publishing) + k_search (helpers)",
            3: "k_seed_b + k_seed_a (gs_seed.hip; + k_describe, k_sched_scan, k_sched_scatter: 0.3 ms)"}.get(form, "k_search")


def spread(ms):
    v = sorted(float(x) for x in ms)
    return [v[0], v[len(v) // 2], v[-1]] if v else None


def gpu_clocks():
    """shader / memory / fabric clocks the driver reports at the end of the run (a box's clocks move the launch time by
    several per cent: profiles/r05_ab_headline_across_commits.txt), read from sysfs - pp_dpm_sclk / _mclk / _fclk, the
    level marked '*' - without starting a program (under rocprofv3 a child that execs is refused on this pool); None
    where the files are not to be had"""
    import glob
    try:
        out = {}
        for card in sorted(glob.glob("/sys/class/drm/card*/device")):
            for name in ("sclk", "mclk", "fclk"):
                f = os.path.join(card, f"pp_dpm_{name}")
                if not os.path.exists(f):
                    continue
                cur = [ln.split(":")[1].strip().rstrip("*").strip() for ln in open(f).read().splitlines() if ln.strip().endswith("*")]
                if cur:
                    out.setdefault(os.path.basename(os.path.dirname(card)), {})[name] = cur[0]
            busy = os.path.join(card, "gpu_busy_percent")   # (a host holds eight cards: the busy one is this run's)
            if os.path.exists(busy) and os.path.basename(os.path.dirname(card)) in out:
                try:
                    out[os.path.basename(os.path.dirname(card))]["busy_percent"] = int(open(busy).read().strip())
                except ValueError:
                    pass
        return out or None
    except Exception:
        return None


def genome_file(workload):
    return os.path.join("/dev/shm" if os.path.isdir("/dev/shm") else "/tmp",
                        f"gs_bench_{workload}_{os.environ.get('MASTER_PORT', os.getpid())}.u8")


def shared_genome(synth, workload, lengths, probs, dist, local_rank):
    """The synthetic genome text, generated ONCE per node: with several ranks, local rank 0 (or the
    launcher, see launch_ranks) fills a file in /dev/shm and every rank maps it, so eight ranks do
    not each spend the generation time and 3.1 GB of host memory."""
    path = os.environ.get("GS_BENCH_TEXT")
    total = int(sum(lengths))
    if path and os.path.exists(path) and os.path.getsize(path) == total:
        text = np.memmap(path, dtype=np.uint8, mode="r")
        names, lengths = [f"chr{i + 1}" for i in range(len(lengths))], [int(x) for x in lengths]
        return text, names, lengths
    if dist is None or dist.get_world_size() == 1:
        return make_workload_genome(synth, workload, lengths, probs)
    path = genome_file(workload)
    if local_rank == 0:
        mm = np.lib.format.open_memmap(path + ".npy", mode="w+", dtype=np.uint8, shape=(total,))
        make_workload_genome(synth, workload, lengths, probs, out=mm)
        mm.flush()
        del mm
    dist.barrier()
    text = np.load(path + ".npy", mmap_mode="r")
    dist.barrier()
    if local_rank == 0:
        os.unlink(path + ".npy")  # the mappings keep the pages until the ranks exit
    return text, [f"chr{i + 1}" for i in range(len(lengths))], [int(x) for x in lengths]


def launch_ranks(args, extra_env=None, module="torch.distributed.run"):
    """Parent of an N-GPU run started as plain `python bench.py --gpus N`: generate the genome once
    into /dev/shm, start N ranks with torch.distributed.run (one per GPU, rendezvous on 127.0.0.1)
    as a child process, relay their output, return the child's exit code.  Nothing here initialises
    the GPU (a process that has must not be replaced or forked into ranks)."""
    import socket
    import subprocess
    synth = import_module("guidescan-cli_amd.synth")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.update(extra_env or {})
    path = None
    if args.workload in WORKLOADS:
        lens_name, _, probs = WORKLOADS[args.workload]
        lengths = [synth.CHR1_LENGTH] if lens_name == "CHR1" else getattr(synth, lens_name)
        path = os.path.join("/dev/shm" if os.path.isdir("/dev/shm") else "/tmp",
                            f"gs_bench_{args.workload}_{port}.u8")
        mm = np.memmap(path, dtype=np.uint8, mode="w+", shape=(int(sum(lengths)),))
        make_workload_genome(synth, args.workload, lengths, probs, out=mm)
        mm.flush()
        del mm
        env["GS_BENCH_TEXT"] = path
    cmd = [sys.executable, "-m", module, "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), str(Path(__file__).resolve())] + sys.argv[1:]
    try:
        return subprocess.run(cmd, env=env).returncode
    finally:
        if path and os.path.exists(path):
            os.unlink(path)


def stub_main(args):
    """GS_BENCH_STUB=1: rehearsal of the multi-rank plumbing WITHOUT a GPU (tests/test_distributed_gloo.py):
    launcher -> torch.distributed.run -> gloo rendezvous on 127.0.0.1 -> genome shared through
    /dev/shm -> barriers around K counted steps -> MAX over ranks -> one JSON line from rank 0.  The
    step is a checksum of the rank's guide shard; the line is marked "stub" and is not a measurement."""
    import torch
    import torch.distributed as dist
    synth = import_module("guidescan-cli_amd.synth")
    parallel = import_module("guidescan-cli_amd.parallel")
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        return f"--gpus {args.gpus} but WORLD_SIZE={world}"
    dist.init_process_group("gloo")
    lens_name, batch, probs = WORKLOADS[args.workload]
    lengths = [synth.CHR1_LENGTH] if lens_name == "CHR1" else getattr(synth, lens_name)
    batch = args.batch or batch
    text, names, lengths = shared_genome(synth, args.workload, lengths, probs, dist, int(os.environ.get("LOCAL_RANK", "0")))
    batch_all = batch * world
    dealt = []
    if args.scaling == "strong":   # the same contiguous shards - or chunks from the shared counter - the GPU path takes
        batch_all = batch
        b = parallel.shard_bounds(batch_all, world)
        all_seqs, all_pams, _, _ = synth.sample_guides(text, batch_all, seed=1000)
        if args.deal:
            mine, _ = parallel.deal_chunks(lambda c, lo, hi: dealt.append((lo, hi)), batch_all, args.deal, "stub", dist)
            pick = np.concatenate([np.arange(lo, hi) for lo, hi in dealt]) if dealt else np.zeros(0, np.int64)
            seqs, pams = all_seqs[pick], all_pams[pick]
            batch = int(pick.shape[0])
        else:
            seqs, pams = all_seqs[b[rank]:b[rank + 1]], all_pams[b[rank]:b[rank + 1]]
            batch = b[rank + 1] - b[rank]
    else:
        seqs, pams, _, _ = synth.sample_guides(text, batch, seed=1000 + rank)
    acc = []
    elapsed = parallel.timed_steps(lambda i: acc.append(int(seqs.sum()) + i), args.steps, args.warmup, lambda: None, dist)
    sums = [None] * world
    dist.all_gather_object(sums, (int(text[::4097].astype(np.int64).sum()), len(acc), int(seqs.shape[0]),
                                  int(seqs.astype(np.int64).sum())))
    if rank == 0:
        print(json.dumps({"stub": True, "metric": "plumbing rehearsal (no GPU)", "n_gpus": world, "steps": args.steps,
                          "warmup": args.warmup, "value": batch_all * args.steps / max(elapsed, 1e-9),
                          "unit": "guides/s", "scaling": args.scaling, "guides_per_step": batch_all,
                          "guides_per_rank": [s[2] for s in sums], "guide_checksums": [s[3] for s in sums],
                          "text_checksums": [s[0] for s in sums],
                          "steps_run_per_rank": [s[1] for s in sums],
                          "shared_text": bool(os.environ.get("GS_BENCH_TEXT"))}), flush=True)
    dist.destroy_process_group()
    return 0


def kernel_stamp():
    """sha256 of the sources that decide k_search's memory traffic: the kernel itself (gs_search.hip up to the end of
    its wrappers, gs_kernels.h: its argument struct and launch geometry), the device helpers, the device layout of the
    index (struct gs_pairtab_dev, gs_strand_dev) and its builders"""
    import hashlib
    h = hashlib.sha256()
    src = (ROOT / "guidescan-cli_amd" / "csrc" / "gs_search.hip").read_text()
    h.update(src[:src.index("/* ---- prepare: ASCII")].encode())
    seed = (ROOT / "guidescan-cli_amd" / "csrc" / "gs_seed.hip").read_text()   # the two-launch form of the table-seeded search
    h.update(seed[:seed.index("/* ---- host: the descriptor pre-pass")].encode())
    h.update((ROOT / "guidescan-cli_amd" / "csrc" / "gs_kernels.h").read_bytes())
    com = (ROOT / "guidescan-cli_amd" / "csrc" / "gs_common.h").read_text()
    h.update(com[com.index("struct gs_pairtab_dev {"):com.index("struct gs_strand {")].encode())  # the device layout
    h.update((ROOT / "guidescan-cli_amd" / "csrc" / "gs_device.h").read_bytes())
    bld = (ROOT / "guidescan-cli_amd" / "csrc" / "gs_index.hip").read_text()
    h.update(bld[:bld.index("C-ABI: index lifecycle")].encode())  # the builder kernels, not the handle bookkeeping
    ptb = (ROOT / "guidescan-cli_amd" / "csrc" / "gs_pairtab.hip").read_text()
    h.update(ptb[:ptb.index("void gs_pairtab_free(")].encode())  # the PAM-pair / deep table kernels, not the memory policy
    return h.hexdigest()[:16]


def recorded_traffic(workload, batch, m):
    """HBM-side bytes of one k_search launch (FETCH_SIZE + WRITE_SIZE).  PMC counters cannot be
    read from inside this process: they come from separate `rocprofv3 --pmc` passes of this same
    command (tools/profile_round.sh), summarised in profiles/traffic.json by tools/make_traffic_json.py
    together with a hash of the kernel sources.  None when no pass was recorded for this exact
    workload WITH THESE SOURCES (a stale number is worse than none)."""
    f = ROOT / "profiles" / "traffic.json"
    if not f.exists():
        return None, None, None
    stamp = kernel_stamp()
    for rec in json.loads(f.read_text()):
        if (rec["workload"] == workload and rec["batch"] == batch and rec["mismatches"] == m
                and rec.get("kernel_sha") == stamp):
            return rec["fetch_bytes"] + rec["write_bytes"], rec["source"], rec.get("issue")
    return None, None, None


# What the random-request rate of the memory system IS (tools/gather_calib.hip, profiles/r05_gather_calib*): every pattern that misses the L2
# - 16-byte words, 64-byte blocks, 128-byte blocks of one load instruction - costs ONE L2 request and ONE read request to
# the fabric (TCC_REQ = TCC_EA0_RDREQ = 1 per block) and runs at 4.8e10 requests per second from a 40 GB table, 5.7e10
# from a table inside the 256 MB Infinity Cache, 1.8e11 and more from one inside the L2; confining a wave-instruction's 64
# addresses to one 4 KB / 64 KB / 2 MB page changes nothing (not translation).  It is the rate at which the chip turns
# L2 misses into DRAM row activations (8 stacks x 32 pseudo-channels, four activates per tFAW window) - a request for 128
# bytes of one row costs what a request for 16 does.
# It is a CALIBRATION of the fabric, not a roofline of a kernel: round 5 priced k_search against per_s_hbm and its k = 13
# run passed it (4.99e10/s: more of its requests were Infinity-Cache hits); the seeding launches run at 3.2e10/s, bound by
# instruction issue.  The bench line carries the rates and the kernel's requests per second side by side, no fraction.
REQUEST_CALIBRATION = {"per_s_hbm": 4.8e10, "per_s_infinity_cache": 5.7e10, "per_s_l2_resident_at_least": 1.76e11,
                   "what": "L2-miss read requests (one per aligned block of up to 128 bytes per load instruction): DRAM row "
                           "activations behind the fabric, not translation (tools/gather_calib.hip, profiles/r05_gather_calib.txt)"}


def recorded_memory_side(workload, batch, m):
    """the whole record of recorded_traffic's pass: fabric read requests and the bytes they brought (FETCH_SIZE tallies every
    read request as 64 bytes, a 128-byte block's too: tools/gather_calib), VALU issue, share of wave cycles spent waiting"""
    f = ROOT / "profiles" / "traffic.json"
    if not f.exists():
        return None
    stamp = kernel_stamp()
    for rec in json.loads(f.read_text()):
        if rec["workload"] == workload and rec["batch"] == batch and rec["mismatches"] == m and rec.get("kernel_sha") == stamp:
            return rec
    return None


def memory_side_fields(rec, launch_ms):
    """roofline fields from a recorded PMC pass, priced on the launch time measured in THIS run"""
    if not rec:
        return {"traffic_requests": None, "traffic_bytes_corrected": None, "frac_hbm": None}
    out = {"traffic_requests": rec.get("read_requests"), "traffic_bytes_corrected": None, "frac_hbm": None}
    rd = rec.get("read_bytes_corrected")
    if rd is None and rec.get("read_bytes_corrected_bounds"):
        rd = rec["read_bytes_corrected_bounds"][1]
        out["traffic_bytes_corrected_is_upper_bound"] = True
        out["traffic_bytes_corrected_bounds"] = [b + rec["write_bytes"] for b in rec["read_bytes_corrected_bounds"]]
    if rd is not None:
        out["traffic_bytes_corrected"] = rd + rec["write_bytes"]
        out["traffic_bytes_corrected_from"] = rec.get("read_bytes_corrected_from", "read requests x the block each brings") + " + WRITE_SIZE"
        if launch_ms:
            out["frac_hbm"] = out["traffic_bytes_corrected"] / (launch_ms * 1e-3) / 1e9 / HBM_PEAK_GBS
    if rec.get("read_requests") and launch_ms:
        out["requests_per_s"] = rec["read_requests"] / (launch_ms * 1e-3)
    if rec.get("issue"):
        i = rec["issue"]
        out["valu_frac_of_issue_cycles"] = i["valu_wave_instructions"] * 4.0 / (1024 * 2.4e9 * i["duration_ms"] * 1e-3)
        out["wait_any_share_of_wave_cycles"] = i.get("wait_any_share")
    out["pmc_source"] = rec["source"]
    return out


def side_steps(torch, api, gidx, d_seqs, d_pams, batch, i, L, P, m, text, names, lengths):
    """gs_score_device on the hits of one batch and gs_kmers_generate on the first chromosome, both
    with inputs and outputs resident in HBM; wall-clock around the (synchronous) calls."""
    s = d_seqs[i * batch:(i + 1) * batch]
    p = d_pams[i * batch:(i + 1) * batch]
    d_off, d_hits, st = gidx.enumerate_device(s.data_ptr(), batch, L, p.data_ptr(), P, mismatches=m)
    gs = api.make_genome_structure(names, lengths)
    d_cfd = torch.empty(max(1, st["n_hits"]), dtype=torch.float32, device="cuda")
    d_spec = torch.empty(batch, dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    best = None
    for _ in range(3):
        t0 = time.perf_counter()
        gidx.score_device(gs, s.data_ptr(), batch, L, P, d_off, d_hits, d_cfd.data_ptr(), d_spec.data_ptr())
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    spec = d_spec.cpu().numpy()
    out = {"score_ms": best * 1e3, "score_hits": int(st["n_hits"]),
           "score_hits_per_s": st["n_hits"] / best if best > 0 else None,
           "specificity_mean": float(spec.mean()), "specificity_min": float(spec.min())}
    del d_cfd, d_spec
    # the host-pointer entry point (gs_enumerate: guides copied in, CSR hit list copied out over
    # PCIe, host vectors allocated) on the same batch: SURVEY 8d's rate (ii); never `value`
    import ctypes as C
    h_seqs = np.ascontiguousarray(s.cpu().numpy())
    h_pams = np.ascontiguousarray(p.cpu().numpy())
    L_ = api.lib()
    best_h = None
    for _ in range(2):
        r = C.c_void_p()
        t0 = time.perf_counter()
        rc = L_.gs_enumerate(gidx._h, h_seqs.ctypes.data, batch, L, h_pams.ctypes.data, P, None, 0, m, 0, C.byref(r))
        dt = time.perf_counter() - t0
        if rc != 0:
            raise RuntimeError(L_.gs_status_string(rc).decode())
        L_.gs_result_free(r)
        best_h = dt if best_h is None else min(best_h, dt)
    out.update({"host_pointer_call_ms": best_h * 1e3, "host_pointer_guides_per_s": batch / best_h})
    n0 = int(lengths[0])
    d_chr = torch.from_numpy(text[:n0]).cuda()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    km = api.generate_kmers(None, "NGG", 20, device=torch.cuda.current_device(), chrm_device_ptr=d_chr.data_ptr(),
                            chrm_len=n0)
    dt = time.perf_counter() - t0
    out.update({"kmers_chr": names[0], "kmers_bp": n0, "kmers_found": km.n, "kmers_ms": dt * 1e3,
                "kmers_bp_per_s": n0 / dt})
    km.close()
    return out


def timed_row(torch, api, gidx, text, names, lengths, L, P, m, n_guides, steps, score, seed):
    """K steps of n_guides sampled guides at <= m mismatches on a resident index (two warm-up steps first: derived
    tables, workspace; every step has its own guides): guides/s and hits/s by the wall clock around the steps, k_search's share from the library's
    own events; with `score`, CFD + specificity of every hit inside the step (gs_score_device)."""
    synth = import_module("guidescan-cli_amd.synth")
    seqs, pams, _, _ = synth.sample_guides(text, n_guides * (steps + 2), seed=seed)
    d_s, d_p = torch.from_numpy(seqs).cuda(), torch.from_numpy(pams).cuda()
    gs = api.make_genome_structure(names, lengths) if score else None
    bufs = {}

    def step(i):
        s, p = d_s[i * n_guides:(i + 1) * n_guides], d_p[i * n_guides:(i + 1) * n_guides]
        d_off, d_hits, st = gidx.enumerate_device(s.data_ptr(), n_guides, L, p.data_ptr(), P, mismatches=m)
        t_sc = 0.0
        if score:
            if bufs.get("n", 0) < st["n_hits"]:
                bufs.update(n=int(st["n_hits"] * 1.25) + 1, cfd=None)
                bufs["cfd"] = torch.empty(bufs["n"], dtype=torch.float32, device="cuda")
                bufs["spec"] = torch.empty(n_guides, dtype=torch.float32, device="cuda")
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            gidx.score_device(gs, s.data_ptr(), n_guides, L, P, d_off, d_hits, bufs["cfd"].data_ptr(), bufs["spec"].data_ptr())
            torch.cuda.synchronize()
            t_sc = time.perf_counter() - t0
        return st, t_sc

    # two warm-up steps, timed and reported: the first builds the derived tables and sizes the slots, the second is the
    # first to order its hits in tiles and allocates that workspace (tens of GB on the repeat-rich genome: a second)
    warm = []
    for i in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        step(i)
        torch.cuda.synchronize()
        warm.append(round((time.perf_counter() - t0) * 1e3, 1))
    t0 = time.perf_counter()
    hits = 0
    ms_search = ms_enum = t_score = 0.0
    per_ms = []
    for i in range(2, steps + 2):
        st, t_sc = step(i)
        per_ms.append(round(st["ms_search"], 2))
        hits += st["n_hits"]
        ms_search += st["ms_search"]
        ms_enum += st["ms_total"]
        t_score += t_sc
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    ctr = gidx.last_counters()
    sh = gidx.last_sharing()
    el_enum = el - t_score
    # the lines THIS algorithm asks for on the last step's guides: one untimed pass through the counting instantiation
    # (the headline's definition of the algorithmic bytes, bench.py main)
    i = steps + 1
    _, _, st_c = gidx.enumerate_device(d_s[i * n_guides:(i + 1) * n_guides].data_ptr(), n_guides, L,
                                       d_p[i * n_guides:(i + 1) * n_guides].data_ptr(), P, mismatches=m, count_requests=True)
    rq = gidx.last_counters()
    lines = {k: rq[k] for k in ("table_lines", "ctx16_lines", "ctx_words", "sa_isa_gathers", "occ_lines")}
    alg_bytes = 64.0 * sum(lines.values()) + 16.0 * st_c["n_matches"]
    return {"guides_per_step": n_guides, "mismatches": m, "steps": steps, "guides_per_s": n_guides * steps / el,
            "hits_per_s": hits / el, "hits_per_guide": hits / (n_guides * steps), "ms_per_step": el / steps * 1e3,
            "enumerate_ms_per_step": el_enum / steps * 1e3, "k_search_ms_per_step": ms_search / steps,
            "score_ms_per_step": (t_score / steps * 1e3) if score else None,
            # what of the enumerate step (prepare .. locate, wall clock) is not the search kernel
            "non_search_share_of_enumerate": 1.0 - (ms_search / 1e3) / el_enum if el_enum > 0 else None,
            "ordered_per_guide_in_lds_tiles": ctr["ordered_in_tiles"], "tile_ordering_gave_up": ctr["tile_ordering_gave_up"],
            "guides_redone": ctr["guides_redone"], "matches_max_per_item": ctr["matches_max_per_item"], "warmup_steps_ms": warm,
            # heavy items handed to idle waves (k_search's heavy instantiation, picked by the handle after a batch with heavy passes)
            "heavy_instantiation": sh["queue_packages"] != 0, "shared_items": sh["shared_items"], "packages": sh["packages"],
            "guides_ordered_device_wide_alone": sh["guides_ordered_device_wide_alone"],
            "search_form": sh["form"], "k_search_ms_min_median_max": spread(per_ms),
            "alg_bytes_per_launch": alg_bytes, "requests_per_guide": {k: v / n_guides for k, v in lines.items()}}


def extra_rows(torch, api, synth, gidx, text, names, lengths, probs, L, P, e2e_guides=1_000_000, only_e2e=False):
    """(rows, index still open or None).  Row 1 on the resident index: 20,000 guides at <= 6 mismatches with CFD
    (config 5's depth).  Row 2: the same index is closed, the repeat-rich genome of the same size is generated and
    indexed, 20,000 guides at <= 3 mismatches (bench.py --workload hg38rep)."""
    rows = {}
    if not only_e2e:
        rows["hg38_20k_m6_cfd"] = timed_row(torch, api, gidx, text, names, lengths, L, P, 6, 20000, 3, True, 4242)
        rows["hg38_20k_m6_cfd"]["roofline"] = row_roofline("hg38", 20000, 6, rows["hg38_20k_m6_cfd"])
    prep = None
    if e2e_guides:
        try:
            prep = e2e_prepare(torch, api, synth, gidx, text, names, lengths, L, P, e2e_guides)
        except Exception as e:
            print(f"[bench] end-to-end row (in-process half) failed: {e!r}", file=sys.stderr)
    gidx.close()
    if prep is not None:
        # the CLI builds its own index in its own process: the bench's is closed, the repeat-rich one not yet built
        try:
            rows[f"e2e_cli_{e2e_guides // 1000}k_m3"] = e2e_cli_row(prep, text, names, lengths, "csv")
        except Exception as e:
            print(f"[bench] end-to-end row (CLI) failed: {e!r}", file=sys.stderr)
        prep = None
    # the repeat-rich genome, and the one whose SINE-like family is Alu-like (1.2e6 copies at 2-15 %)
    for wl, key, steps in (() if only_e2e else (("hg38rep", "hg38rep_20k_m3", 3), ("hg38alu", "hg38alu_20k_m3", 2))):
        t0 = time.time()
        text2, names2, lengths2 = make_workload_genome(synth, wl, lengths, probs)
        t_gen = time.time() - t0
        t0 = time.time()
        g2 = api.GenomeIndex.build(text2, device=torch.cuda.current_device())
        torch.cuda.synchronize()
        t_idx = time.time() - t0
        try:
            rows[key] = dict(timed_row(torch, api, g2, text2, names2, lengths2, L, P, 3, 20000, steps, False, 1000),
                             genome_gen_s=round(t_gen, 1), index_build_s=round(t_idx, 1))
            rows[key]["roofline"] = row_roofline(wl, 20000, 3, rows[key])
        finally:
            g2.close()
        del text2
    return rows, None


def e2e_prepare(torch, api, synth, gidx, text, names, lengths, L, P, n_guides, check_guides=100_000):
    """The in-process half of the end-to-end row, on the resident index: rates (i) and (ii) of SURVEY 8d on the row's own
    guide set, and the CSV text of its first `check_guides` guides from the in-process formatter (gs_score +
    gs_format_guides_scored: the same encoder the CLI calls batch by batch) - what the CLI's file must begin with."""
    import ctypes as C
    import hashlib
    seqs, pams, pos, strands = synth.sample_guides(text, n_guides, seed=7777)
    d_s, d_p = torch.from_numpy(seqs).cuda(), torch.from_numpy(pams).cuda()
    ms_dev = []
    for _ in range(3):   # (i) inputs and results in HBM; the first call sizes the workspace
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        gidx.enumerate_device(d_s.data_ptr(), n_guides, L, d_p.data_ptr(), P, mismatches=3)
        torch.cuda.synchronize()
        ms_dev.append((time.perf_counter() - t0) * 1e3)
    L_ = api.lib()
    best_h = None
    for _ in range(2):   # (ii) host pointers in, host CSR out (PCIe both ways)
        r = C.c_void_p()
        t0 = time.perf_counter()
        rc = L_.gs_enumerate(gidx._h, seqs.ctypes.data, n_guides, L, pams.ctypes.data, P, None, 0, 3, 0, C.byref(r))
        dt = time.perf_counter() - t0
        if rc != 0:
            raise RuntimeError(L_.gs_status_string(rc).decode())
        L_.gs_result_free(r)
        best_h = dt if best_h is None else min(best_h, dt)
    nc = min(check_guides, n_guides)
    gs = api.make_genome_structure(names, lengths)
    off, hits, _ = gidx.enumerate(seqs[:nc], pams[:nc], mismatches=3)
    _, spec = gidx.score(gs, seqs[:nc], P, off, hits, want_cfd=False)
    ids = [f"g{i}" for i in range(nc)]
    body = api.format_guides(gs, ids, [seqs[i].tobytes().decode() for i in range(nc)], ["NGG"] * nc,
                             [chr(strands[i]) == "+" for i in range(nc)], off, hits, spec, 3, sam=False, complete=True)
    head = api.format_header(gs, sam=False, complete=True).encode()
    expect = head + body
    return {"seqs": seqs, "pos": pos, "strands": strands, "n": n_guides, "expect_prefix": expect,
            "in_hbm_ms": min(ms_dev[1:]), "host_pointer_ms": best_h * 1e3, "check_guides": nc,
            "expect_sha256": hashlib.sha256(expect).hexdigest()}


def e2e_cli_row(prep, text, names, lengths, fmt="csv"):
    """SURVEY 8d's rate (iii) under this run's clock: the built `guidescan enumerate` (the reference's command line,
    src/guidescan.cxx:42-73,181-258) on the same genome and the row's guides, kmers CSV in -> database out, in a process
    of its own that builds its own index.  The reference's timer starts after the index is loaded (src/guidescan.cxx:239)
    and so does `Processed N kmers in S seconds`: guides/s = N / S; the index build is reported beside it."""
    import hashlib
    import shutil
    import tempfile
    base = os.environ.get("GS_E2E_DIR") or ("/dev/shm" if os.path.isdir("/dev/shm") else "/tmp")
    d = Path(tempfile.mkdtemp(prefix="gs_e2e_", dir=base))
    try:
        np.asarray(text).tofile(d / "g.dna")
        (d / "g.gs").write_text("".join(f"{a}\n{b}\n" for a, b in zip(names, lengths)))
        n, seqs, pos, strands = prep["n"], prep["seqs"], prep["pos"], prep["strands"]
        rows = ["id,sequence,pam,chromosome,position,sense\n"]
        sq = [x.decode() for x in np.ascontiguousarray(seqs).view(f"S{seqs.shape[1]}").ravel()]
        rows += [f"g{i},{sq[i]},NGG,chr1,{int(pos[i]) + 1},{chr(strands[i])}\n" for i in range(n)]
        (d / "k.csv").write_text("".join(rows))
        del rows, sq
        cli = ROOT / "guidescan-cli_amd" / "bin" / "guidescan"
        out_file = d / f"o.{fmt}"
        import re
        # the command twice, the faster run reported and both times listed: one run in four on this pool showed a
        # device stage of 1.5 s instead of 0.2 (a box's first large allocations), which says nothing about the path
        runs = []
        for attempt in range(2):
            if out_file.exists():
                out_file.unlink()
            t0 = time.perf_counter()
            r = subprocess.run([str(cli), "enumerate", str(d / "g"), "-f", str(d / "k.csv"), "-o", str(out_file), "-m", "3",
                                "--format", fmt, "--mode", "complete"], capture_output=True, text=True, timeout=900)
            wall = time.perf_counter() - t0
            if r.returncode != 0:
                raise RuntimeError(f"guidescan enumerate failed: {r.stderr[-400:]}")
            m_proc = re.search(r"Processed (\d+) kmers in ([0-9.eE+-]+) seconds", r.stdout)
            runs.append((float(m_proc.group(2)), wall, r))
        each_run = [x[0] for x in runs]
        _, wall, r = min(runs, key=lambda x: x[0])
        m_proc = re.search(r"Processed (\d+) kmers in ([0-9.eE+-]+) seconds", r.stdout)
        m_build = re.search(r"index on .* in ([0-9.eE+-]+) s", r.stdout)
        m_st = re.search(r"device ([0-9.eE+-]+) s, text formatting ([0-9.eE+-]+) s, file writes ([0-9.eE+-]+) s", r.stdout)
        secs = float(m_proc.group(2))
        size = out_file.stat().st_size
        out = {"format": fmt, "guides": n, "mismatches": 3,
               "i_results_in_hbm_ms": prep["in_hbm_ms"], "i_guides_per_s": n / (prep["in_hbm_ms"] * 1e-3),
               "ii_host_pointers_ms": prep["host_pointer_ms"], "ii_guides_per_s": n / (prep["host_pointer_ms"] * 1e-3),
               "iii_cli_seconds_after_index_load": secs, "iii_guides_per_s": n / secs, "iii_cli_seconds_each_run": each_run,
               "cli_index_build_s": float(m_build.group(1)) if m_build else None,
               "cli_wall_s_incl_text_read_and_index_build": wall,
               "cli_stage_seconds_overlapping": ({"device": float(m_st.group(1)), "text_formatting": float(m_st.group(2)),
                                                  "file_writes": float(m_st.group(3))} if m_st else None),
               "output_bytes": size, "writer_GB_per_s": size / secs / 1e9, "output_on": base}
        if os.environ.get("GS_DEBUG"):   # the library's own account of the CLI's batches (forms, tables, arena)
            out["cli_stderr_tail"] = r.stderr[-6000:]
        if fmt == "csv":
            exp = prep["expect_prefix"]
            with open(out_file, "rb") as fh:
                got = fh.read(len(exp))
            out.update({"first_guides_checked": prep["check_guides"], "checked_bytes": len(exp),
                        "first_bytes_equal_in_process_formatter": got == exp,
                        "sha256_first_bytes_cli": hashlib.sha256(got).hexdigest(), "sha256_in_process": prep["expect_sha256"]})
        return out
    finally:
        shutil.rmtree(d, ignore_errors=True)


def row_roofline(workload, batch, m, row):
    """the roofline of an extra row's search, by the headline's definition: achieved = the algorithm's own bytes (64 B per
    distinct line its loads ask for, counted by the counting instantiation on the row's guides, + 16 B per match record)
    over the launch time measured in this run, frac = achieved / 8 TB/s.  Beside it the memory side from the recorded PMC
    passes of `bench.py --workload W --batch B --mismatches M` (tools/profile_round.sh, stamped with the kernel sources'
    hash): corrected bytes and frac_hbm - a different quantity (what HBM delivered, not what the algorithm asked for)."""
    launch_ms = row["k_search_ms_per_step"]
    rec = recorded_memory_side(workload, batch, m)
    mem = memory_side_fields(rec, launch_ms)
    alg = row.get("alg_bytes_per_launch")
    achieved = alg / (launch_ms * 1e-3) / 1e9 if alg and launch_ms else None
    form = row.get("search_form", 0)
    out = {"bound": "hbm",
           "kernel": {1: "k_search (heavy instantiation)", 2: "k_search (publishing) + k_search (helpers)",
                      3: "k_seed_b + k_seed_a"}.get(form, "k_search"),
           "avg_launch_ms": launch_ms, "peak": HBM_PEAK_GBS, "unit": "GB/s", "achieved": achieved,
           "frac": achieved / HBM_PEAK_GBS if achieved else None,
           # the lines are counted per load instruction: on a repeat-rich batch neighbouring hits ask for the same lines and the
           # L2s serve them, so the algorithm's bytes can pass what HBM delivers - frac_hbm is then the memory side's share
           "frac_note": ("counted lines exceed what HBM delivered: many are served by L2 (see frac_hbm / traffic)"
                         if achieved and achieved > HBM_PEAK_GBS * 0.8 else None),
           "traffic": (rec["fetch_bytes"] + rec["write_bytes"]) if rec else None,
           "alg_bytes_per_launch": alg,
           "alg_bytes_are": "64 B per distinct line the search's loads ask for (counting instantiation, this row's guides) + 16 B "
                            "per match record: the headline's definition"}
    out.update(mem)
    return out


def verify_last_batch(torch, gidx, d_seqs, d_pams, batch, i, L, P, m, text, seqs):
    """size-independent properties at the full workload size: CSR offsets monotone, keys ascending
    within each guide, every guide (sampled from the genome) has a distance-0 hit whose 23-mer in
    the genome equals guide+PAM.  Reads the device result back through ctypes."""
    import ctypes as C
    api = import_module("guidescan-cli_amd.api")
    synth = import_module("guidescan-cli_amd.synth")
    s = d_seqs[i * batch:(i + 1) * batch]
    p = d_pams[i * batch:(i + 1) * batch]
    d_off, d_hits, st = gidx.enumerate_device(s.data_ptr(), batch, L, p.data_ptr(), P, mismatches=m)
    n_hits = st["n_hits"]
    off = torch.empty(batch + 1, dtype=torch.int64, device="cuda")
    hits = torch.empty((n_hits, 2), dtype=torch.int64, device="cuda")
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    assert hip.hipMemcpy(off.data_ptr(), d_off, 8 * (batch + 1), 3) == 0
    assert hip.hipMemcpy(hits.data_ptr(), d_hits, 16 * n_hits, 3) == 0
    off = off.cpu().numpy()
    hits = hits.cpu().numpy()
    pos, key = hits[:, 0], hits[:, 1].view(np.uint64)
    assert off[0] == 0 and off[-1] == n_hits and np.all(np.diff(off) >= 0)
    seg = np.repeat(np.arange(batch), np.diff(off))
    same = seg[1:] == seg[:-1]
    assert np.all(key[1:][same] >= key[:-1][same]), "keys not ascending within a guide"
    d0 = (key >> np.uint64(61)) == 0
    has0 = np.zeros(batch, dtype=bool)
    has0[seg[d0]] = True
    assert has0.all(), "a sampled guide has no distance-0 hit"
    # check a sample of distance-0 hits against the genome text itself
    idx = np.nonzero(d0)[0][:: max(1, int(d0.sum()) // 20000)]
    Lg = text.shape[0]
    g = seqs[i * batch:(i + 1) * batch]
    for h in idx:
        q = np.concatenate([g[seg[h]], np.frombuffer(b"NGG", np.uint8)])
        if (key[h] >> np.uint64(60)) & np.uint64(1):          # reverse index: + strand, pos = end
            w = text[pos[h] - 22:pos[h] + 1]
        else:                                                  # forward index: - strand, pos = -start
            w = synth.reverse_complement_bytes(text[-pos[h]:-pos[h] + 23])
        assert np.array_equal(w[:20], q[:20]) and w[21] == ord("G") and w[22] == ord("G")
    out = {"guides": int(batch), "hits": int(n_hits), "distance0_hits_checked_vs_text": int(idx.size)}
    # the same batch through the slower, more literal paths of the library must give the SAME
    # BYTES: (a) one-sided seeding only (GS_NO_BIDIR: every site comes from this strand's table,
    # no second class, the full plan), the whole batch; (b) the reference-order walk from the
    # root (GS_FLAG_FAITHFUL_WALK: no table, no context arrays), the first 20,000 guides
    off_d = torch.from_numpy(off.astype(np.int64)).cuda()
    hits_d = torch.from_numpy(hits).cuda()

    def fetch(n_g, **kw):
        d_o, d_h, st2 = gidx.enumerate_device(s.data_ptr(), n_g, L, p.data_ptr(), P, mismatches=m, **kw)
        o2 = torch.empty(n_g + 1, dtype=torch.int64, device="cuda")
        h2 = torch.empty((st2["n_hits"], 2), dtype=torch.int64, device="cuda")
        assert hip.hipMemcpy(o2.data_ptr(), d_o, 8 * (n_g + 1), 3) == 0
        assert hip.hipMemcpy(h2.data_ptr(), d_h, 16 * st2["n_hits"], 3) == 0
        return o2, h2, st2

    gidx.set_option("GS_NO_BIDIR", "1")
    try:
        o2, h2, st2 = fetch(batch)
    finally:
        gidx.set_option("GS_NO_BIDIR", None)
    assert torch.equal(o2, off_d) and torch.equal(h2, hits_d), "one-sided and two-sided seeding differ"
    out["one_sided_identical_bytes"] = {"guides": int(batch), "hits": int(st2["n_hits"]),
                                        "k_search_ms": round(st2["ms_search"], 1)}
    nw = min(batch, 20000)
    o3, h3, st3 = fetch(nw, faithful=True)
    nh = int(off[nw])
    assert torch.equal(o3, off_d[:nw + 1]) and torch.equal(h3, hits_d[:nh]), "walk and fast path differ"
    out["reference_order_walk_identical_bytes"] = {"guides": int(nw), "hits": nh,
                                                   "k_search_ms": round(st3["ms_search"], 1)}
    return out


def cpu_baseline(text, gidx, seqs, pams, m, sample):
    """The CPU oracle (oracle/gs_oracle.c, a port of the reference's algorithm) timed on this
    host's cores on a bounded sample of the same guides.  The oracle only serves as the
    reported baseline here, never as the measured path."""
    import oracle_lib as ol
    cores = os.cpu_count() or 1
    if sample < 0:
        sample = max(cores * 32, 256)
    sample = min(sample, seqs.shape[0])
    oidx = ol.OracleIndex(text, sa_provider=lambda s: gidx.suffix_array(s), nthreads=cores)
    opts = ol.make_opts(mismatches=m)
    t0 = time.perf_counter()
    tot, counts, ctr = oidx.enumerate_batch(seqs[:sample], pams[:sample], opts, nthreads=cores)
    dt = time.perf_counter() - t0
    oidx.close()
    return {"value": sample / dt, "unit": "guides/s", "cores": cores, "kind": "port",
            "sample": f"first {sample} guides of rank 0's batch, {cores} threads (guide i -> thread i mod n "
                      f"as src/guidescan.cxx:229-231), {dt:.1f} s",
            "n_ext_per_guide": ctr.n_ext / sample, "rank_bwt_per_guide": ctr.n_rank / sample}


def physical_cores():
    """distinct (package, core) pairs of /proc/cpuinfo; None where that cannot be read"""
    try:
        seen, phys, core = set(), None, None
        for line in open("/proc/cpuinfo"):
            if line.startswith("physical id"):
                phys = line.split(":")[1].strip()
            elif line.startswith("core id"):
                core = line.split(":")[1].strip()
            elif not line.strip():
                if phys is not None and core is not None:
                    seen.add((phys, core))
                phys = core = None
        return len(seen) or None
    except OSError:
        return None


def cpu_baseline_reference(text, names, lengths, gidx, seqs, pams, m, sample, threads="", hwpopcnt=False):
    """The reference itself (oracle/_ref/gs_ref_enumerate: its index.hpp / process.hpp /
    printer.hpp / csa_wt compiled in place by oracle/Makefile) on this host's cores, timed by the
    same clock the reference prints ("Processed N kmers in S seconds", src/guidescan.cxx:239-256:
    search + text formatting + write, index load excluded).  Its SDSL index files are written
    from the suffix arrays copied back from the GPU through the compiled reference containers."""
    import re
    import shutil
    import subprocess
    import tempfile
    import oracle_lib as ol
    synth = import_module("guidescan-cli_amd.synth")
    ref = ol.ref()
    shim = ol.ORACLE_DIR / "_ref" / "gs_ref_enumerate"
    if ref is None or not shim.exists():
        raise SystemExit("--cpu-kind reference needs oracle/_ref (built where /root/reference exists)")
    cores = os.cpu_count() or 1
    if sample < 0:
        sample = max(cores * 32, 256)
    sample = min(sample, seqs.shape[0])
    td = tempfile.mkdtemp(prefix="gsref_")
    try:
        n = text.shape[0] + 1
        t0 = time.time()
        def write_strand(strand, suffix):  # ctypes releases the GIL: both strands at once
            sa = gidx.suffix_array(strand)
            st = np.ascontiguousarray(text if strand == 0 else synth.reverse_complement_bytes(text))
            h = ref.ref_index_build_text(st.ctypes.data, sa.ctypes.data, n,
                                         os.path.join(td, f"tmp{strand}.sdsl").encode())
            assert ref.ref_write_index_file(h, os.path.join(td, "g" + suffix).encode()) == 0
            ref.ref_index_free(h)

        import threading
        ths = [threading.Thread(target=write_strand, args=a) for a in ((0, ".forward"), (1, ".reverse"))]
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        with open(os.path.join(td, "g.gs"), "w") as f:
            f.write("".join(f"{a}\n{b}\n" for a, b in zip(names, lengths)))
        synth.write_kmers_csv(os.path.join(td, "k.csv"), [f"g{i}" for i in range(sample)],
                              [seqs[i].tobytes().decode() for i in range(sample)],
                              [pams[i].tobytes().decode() for i in range(sample)], [names[0]] * sample,
                              [1] * sample, ["+"] * sample)
        t_files = time.time() - t0
        # the reference deals guide i to thread i mod n (src/guidescan.cxx:229-231) and is latency bound: the thread
        # count that serves it best is found, not assumed - each count on the same guides, the best reported
        tlist = sorted({min(cores, max(1, int(x))) for x in threads.split(",") if x.strip()} or {cores})
        sweep = []
        best_dt, best_t = None, None
        for nthr in tlist:
            env = dict(os.environ, GS_REF_THREADS=str(nthr))
            r = subprocess.run([str(shim), os.path.join(td, "g"), os.path.join(td, "k.csv"), os.path.join(td, f"o{nthr}.csv"),
                                "csv", "complete", str(m), "0", "0", "-1", "-1", "0"], env=env, check=True,
                               timeout=3600, stderr=subprocess.PIPE, stdout=subprocess.DEVNULL)
            dt_n = float(re.search(r"kmers in ([0-9.eE+-]+) s", r.stderr.decode()).group(1))
            sweep.append({"threads": nthr, "guides_per_thread": sample // nthr, "seconds": round(dt_n, 2),
                          "guides_per_s": sample / dt_n})
            if best_dt is None or dt_n < best_dt:
                best_dt, best_t = dt_n, nthr
        dt = best_dt
        os.replace(os.path.join(td, f"o{best_t}.csv"), os.path.join(td, "o.csv"))
        env = dict(os.environ, GS_REF_THREADS=str(best_t))
        # the same run by the build with hardware POPCNT (oracle/Makefile: -march=x86-64-v3; the box's own
        # -march=native cannot be built there, the reference's sources do not travel)
        hw = None
        shim_hw = ol.ORACLE_DIR / "_ref" / "gs_ref_enumerate_hwpopcnt"
        if hwpopcnt and shim_hw.exists():
            try:
                r2 = subprocess.run([str(shim_hw), os.path.join(td, "g"), os.path.join(td, "k.csv"), os.path.join(td, "o2.csv"),
                                     "csv", "complete", str(m), "0", "0", "-1", "-1", "0"], env=env, check=True,
                                    timeout=3600, stderr=subprocess.PIPE, stdout=subprocess.DEVNULL)
                dt2 = float(re.search(r"kmers in ([0-9.eE+-]+) s", r2.stderr.decode()).group(1))
                with open(os.path.join(td, "o.csv"), "rb") as f1, open(os.path.join(td, "o2.csv"), "rb") as f2:
                    same = sorted(f1.read().splitlines()) == sorted(f2.read().splitlines())
                hw = {"value": sample / dt2, "unit": "guides/s", "cores": best_t, "kind": "reference",
                      "sample": f"the same {sample} guides through the reference built -O3 -DNDEBUG -march=x86-64-v3 "
                                f"(hardware POPCNT in sdsl bits::cnt), {best_t} threads, {dt2:.1f} s",
                      "same_lines_as_the_as_shipped_build": same}
            except Exception as e:   # a host CPU below x86-64-v3, or any failure of the side run: no second row
                print(f"[bench] hardware-POPCNT reference row skipped: {e!r}", file=sys.stderr)
        # the baseline's output doubles as a parity check at this size: its data lines (row order
        # varies with the thread count, so as a sorted list) against the product's lines for the
        # same guides - device search, device scoring, gs_format_guide_scored
        with open(os.path.join(td, "o.csv")) as f:
            want = sorted(f.read().splitlines()[1:])
        api = import_module("guidescan-cli_amd.api")
        gs = api.make_genome_structure(names, lengths)
        off, hits, _ = gidx.enumerate(seqs[:sample], pams[:sample], mismatches=m)
        _, spec = gidx.score(gs, seqs[:sample], pams.shape[1], off, hits, want_cfd=False)
        got = []
        for i in range(sample):
            got += api.format_guide(gs, f"g{i}", seqs[i].tobytes().decode(), pams[i].tobytes().decode(), True,
                                    hits[off[i]:off[i + 1]], m, specificity=spec[i]).splitlines()
        got.sort()
        parity = {"lines": len(want), "identical_to_product": got == want}
        if got != want:
            print(f"[bench] PARITY FAILURE: {len(want)} reference lines vs {len(got)} product lines", file=sys.stderr)
    finally:
        shutil.rmtree(td, ignore_errors=True)
    out = {"value": sample / dt, "unit": "guides/s", "cores": best_t, "kind": "reference",
           "sample": f"first {sample} guides of rank 0's batch through the compiled reference "
                     f"(process_kmers_to_stream, CSV out, built -O3 -DNDEBUG as its Release build), best of "
                     f"{len(sweep)} thread counts: {best_t} threads, {dt:.1f} s; SDSL index files written in "
                     f"{t_files:.0f} s (untimed)",
           "host": {"logical_cpus": cores, "physical_cores": physical_cores()},
           "thread_sweep": sweep, "parity": parity}
    if hw:
        out["hwpopcnt"] = hw
    return out


if __name__ == "__main__":
    main()
