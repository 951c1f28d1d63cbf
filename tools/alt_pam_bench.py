#!/usr/bin/env python3
"""k_search time with an alt PAM whose pair of concrete bases differs from the guides' own (NGG + NAG:
two PAM-pair tables): python tools/alt_pam_bench.py [workload=hg38] [batch=1000000] [m=3] [alt=NAG]"""
import sys
import time
from importlib import import_module
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402

api = import_module("guidescan-cli_amd.api")
synth = import_module("guidescan-cli_amd.synth")


def main():
    workload = sys.argv[1] if len(sys.argv) > 1 else "hg38"
    batch = int(sys.argv[2]) if len(sys.argv) > 2 else 1000000
    m = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    alt = tuple(sys.argv[4].split(",")) if len(sys.argv) > 4 else ("NAG",)
    lname, _, probs = bench.WORKLOADS[workload]
    lens = [synth.CHR1_LENGTH] if lname == "CHR1" else getattr(synth, lname)
    text, names, lengths = bench.make_workload_genome(synth, workload, lens, probs)
    t0 = time.time()
    g = api.GenomeIndex.build(text, device=0)
    print(f"index {time.time() - t0:.1f} s, {g.device_bytes / 1e9:.1f} GB", flush=True)
    seqs, pams, _, _ = synth.sample_guides(text, batch, seed=5)
    d_s = torch.from_numpy(np.ascontiguousarray(seqs)).cuda()
    d_p = torch.from_numpy(np.ascontiguousarray(pams)).cuda()
    for label, alts in (("own PAM only", ()), ("with alt " + ",".join(alt), alt)):
        for rep in range(3):
            _, _, st = g.enumerate_device(d_s.data_ptr(), batch, seqs.shape[1], d_p.data_ptr(), pams.shape[1], mismatches=m,
                                          alt_pams=alts)
            c = g.last_counters()
            print(f"{label}: run {rep}: k_search {st['ms_search']:.1f} ms, step {st['ms_total']:.1f} ms, hits {st['n_hits']}, "
                  f"items through PAM-pair tables {c['items_pair_tables']} of {2 * batch}, index now {g.device_bytes / 1e9:.1f} GB",
                  flush=True)


if __name__ == "__main__":
    main()
