#!/usr/bin/env python3
"""The forms of the search for a batch whose every pattern has its PAM-pair + deep tables, on one handle, same guides:
GS_SEED_FORM 0 = k_search_fast_pd (one launch, every item sets itself up), 1 = the two launches from descriptors
(gs_seed.hip) in the order given, 2 = ... with the guides scheduled by their symbols, each XCD its own piece.
Prints k_search per step and the CRC-32 of offsets + hits: the bytes must not depend on the form.
Usage (GPU box, repo root): python tools/seed_forms.py [workload] [batch] [m] [forms, e.g. 0,1,2]"""
import ctypes as C
import sys
import zlib
from importlib import import_module
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def main():
    import torch
    bench = import_module("bench")
    api = import_module("guidescan-cli_amd.api")
    synth = import_module("guidescan-cli_amd.synth")
    workload = sys.argv[1] if len(sys.argv) > 1 else "hg38"
    lens_name, batch, probs = bench.WORKLOADS[workload]
    if len(sys.argv) > 2 and int(sys.argv[2]):
        batch = int(sys.argv[2])
    m = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    forms = [int(x) for x in (sys.argv[4] if len(sys.argv) > 4 else "0,1,2").split(",")]
    lengths = [synth.CHR1_LENGTH] if lens_name == "CHR1" else getattr(synth, lens_name)
    text, names, lengths = bench.make_workload_genome(synth, workload, lengths, probs)
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    hip.hipMemcpy.restype = C.c_int
    g = api.GenomeIndex.build(text, device=0)
    try:
        seqs, pams, _, _ = synth.sample_guides(text, batch, seed=1000)
        s, p = torch.from_numpy(seqs).cuda(), torch.from_numpy(pams).cuda()
        ref = None
        for form in forms:
            g.set_option("GS_SEED_FORM", str(form))
            ms, tot = [], []
            for _ in range(4):
                torch.cuda.synchronize()
                d_off, d_hits, st = g.enumerate_device(s.data_ptr(), batch, seqs.shape[1], p.data_ptr(), pams.shape[1], mismatches=m)
                ms.append(round(st["ms_search"], 2))
                tot.append(round(st["ms_total"], 2))
            n_hits = int(st["n_hits"])
            off = np.empty(batch + 1, np.uint64)
            assert hip.hipMemcpy(off.ctypes.data, d_off, 8 * (batch + 1), 2) == 0
            crc = zlib.crc32(off.tobytes())
            buf = np.empty(1 << 26, np.uint8)
            for pos in range(0, 16 * n_hits, buf.size):
                c = min(buf.size, 16 * n_hits - pos)
                assert hip.hipMemcpy(buf.ctypes.data, d_hits + pos, c, 2) == 0
                crc = zlib.crc32(buf[:c].tobytes(), crc)
            print(f"form {form}: k_search {ms} ms, step {tot} ms, {n_hits} hits, crc32 {crc:08x}", flush=True)
            ref = crc if ref is None else ref
            if crc != ref:
                print("DIFFERENT BYTES", flush=True)
    finally:
        g.close()


if __name__ == "__main__":
    main()
