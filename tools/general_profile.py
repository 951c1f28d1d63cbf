#!/usr/bin/env python3
"""One general-path batch at hg38 size for a kernel trace: rocprofv3 --kernel-trace --stats -- python3 tools/general_profile.py
[--guides 4096] [--mismatches 1 --rna 1 --dna 1]"""
import argparse
import sys
import time
from importlib import import_module
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
ap = argparse.ArgumentParser()
ap.add_argument("--guides", type=int, default=4096)
ap.add_argument("--mismatches", type=int, default=1)
ap.add_argument("--rna", type=int, default=1)
ap.add_argument("--dna", type=int, default=1)
ap.add_argument("--workload", default="hg38")
args = ap.parse_args()
api = import_module("guidescan-cli_amd.api")
synth = import_module("guidescan-cli_amd.synth")
lengths = synth.GRCH38_LENGTHS if args.workload == "hg38" else [synth.CHR1_LENGTH]
text, names, lens = synth.make_genome(lengths, seed=1)
gidx = api.GenomeIndex.build(text, device=0)
seqs, pams, _, _ = synth.sample_guides(text, args.guides, seed=77)
gidx.enumerate_general(seqs[:4], pams[:4], mismatches=args.mismatches, rna_bulges=args.rna, dna_bulges=args.dna)
for _ in range(2):
    t0 = time.perf_counter()
    off, hx = gidx.enumerate_general(seqs, pams, mismatches=args.mismatches, rna_bulges=args.rna, dna_bulges=args.dna)
    dt = time.perf_counter() - t0
    print(f"{args.guides} guides, m={args.mismatches} rna={args.rna} dna={args.dna}: {dt * 1e3:.1f} ms, {args.guides / dt:.0f} guides/s, "
          f"{int(off[-1])} hits", flush=True)
gidx.close()
