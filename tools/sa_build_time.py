#!/usr/bin/env python3
"""Index build time with the two suffix array builders on one box, same text: the first builder (every row every round,
GS_SA_PLAIN=1) and the one that leaves sorted suffixes alone (gs_suffix.hip, the default); the second one's arrays proved
row by row against the text (gs_index_verify_sa, every row).
Usage (GPU box, repo root): python tools/sa_build_time.py [workload=hg38] [plain,new]"""
import os
import sys
import time
from importlib import import_module
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def main():
    import torch
    torch.zeros(1, device="cuda")
    bench = import_module("bench")
    api = import_module("guidescan-cli_amd.api")
    synth = import_module("guidescan-cli_amd.synth")
    workload = sys.argv[1] if len(sys.argv) > 1 else "hg38"
    which = (sys.argv[2] if len(sys.argv) > 2 else "plain,new").split(",")
    lens_name, batch, probs = bench.WORKLOADS[workload]
    lengths = [synth.CHR1_LENGTH] if lens_name == "CHR1" else getattr(synth, lens_name)
    text, names, lengths = bench.make_workload_genome(synth, workload, lengths, probs)
    print(f"{workload}: {text.shape[0]} symbols, {int((text == ord('N')).sum())} of them N", flush=True)
    for w in which:
        if w == "plain":
            os.environ["GS_SA_PLAIN"] = "1"
        else:
            os.environ.pop("GS_SA_PLAIN", None)
        os.environ["GS_DEBUG"] = "1"
        t0 = time.time()
        g = api.GenomeIndex.build(text, device=0)
        torch.cuda.synchronize()
        t = time.time() - t0
        print(f"builder {w}: index built in {t:.2f} s", flush=True)
        if w != "plain":
            t0 = time.time()
            for s in (0, 1):
                rep = g.verify_sa(text, strand=s, samples="all")
                bad = rep["not_permutation"] + rep["out_of_order"] + rep["undecided"] + rep["bwt_mismatch"]
                print(f"  strand {s}: {rep['rows']} rows, every adjacent pair checked: {bad} bad", flush=True)
                assert bad == 0, rep
            print(f"  verified in {time.time() - t0:.2f} s", flush=True)
        g.close()
        del g


if __name__ == "__main__":
    main()
