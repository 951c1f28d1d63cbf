#!/usr/bin/env python3
"""A light batch with a few giant items: 1 M guides with few hits each + the H heaviest guides of a repeat family (an Alu-like
family: 10^5 .. 10^6 hits per guide), on one resident index, under the forms k_search has for heavy items - every item with
its wave (plain), the heavy form in one launch (GS_HEAVY=1), the plain form publishing + the heavy form beside it
(GS_SPLIT_SHARE=2) - and what the handle picks by itself on the second batch of the shape (default).
Usage (GPU box, repo root): python tools/mixed_batch.py [workload=hg38alu] [light=1000000] [heavy=8] [steps=3]"""
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
    workload = sys.argv[1] if len(sys.argv) > 1 else "hg38alu"
    n_light = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
    n_heavy = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    steps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
    lens_name, _, probs = bench.WORKLOADS[workload]
    lengths = [synth.CHR1_LENGTH] if lens_name == "CHR1" else getattr(synth, lens_name)
    text, names, lengths = bench.make_workload_genome(synth, workload, lengths, probs)
    g = api.GenomeIndex.build(text, device=0)
    try:
        # hits per guide of a large sample, 20,000 at a time (the handle's own choice of form; only the counts are used)
        pool = int(1.6 * n_light)
        seqs, pams, _, _ = synth.sample_guides(text, pool, seed=4242)
        d_s, d_p = torch.from_numpy(seqs).cuda(), torch.from_numpy(pams).cuda()
        counts = np.zeros(pool, np.int64)
        import ctypes as C
        hip = C.CDLL("libamdhip64.so")
        hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        hip.hipMemcpy.restype = C.c_int
        B = 20000
        for lo in range(0, pool, B):
            n = min(B, pool - lo)
            d_off, _, _ = g.enumerate_device(d_s[lo:].data_ptr(), n, 20, d_p[lo:].data_ptr(), 3, mismatches=3)
            off = np.empty(n + 1, np.uint64)
            assert hip.hipMemcpy(off.ctypes.data, d_off, 8 * (n + 1), 2) == 0
            counts[lo:lo + n] = np.diff(off.astype(np.int64))
        light = np.nonzero(counts < 200)[0][:n_light]
        heavy = np.argsort(counts)[-n_heavy:]
        assert light.shape[0] == n_light, (light.shape[0], "light guides in the pool")
        rng = np.random.default_rng(7)
        pick = np.concatenate([light, heavy])
        rng.shuffle(pick)
        b_s, b_p = torch.from_numpy(seqs[pick]).cuda(), torch.from_numpy(pams[pick]).cuda()
        n = pick.shape[0]
        print(json.dumps({"workload": workload, "light_guides": n_light, "light_hits": int(counts[light].sum()),
                          "heavy_guides": n_heavy, "heavy_hits": [int(x) for x in counts[heavy]]}), flush=True)
        used = []
        for s in ["GS_HEAVY=0", "GS_HEAVY=1", "GS_SPLIT_SHARE=2", "default"]:
            for k in used:
                g.set_option(k, None)
            used.clear()
            kv = dict(x.split("=") for x in s.split(",")) if s != "default" else {}
            g.set_options(**kv)
            used.extend(kv)
            rows, crc = [], 0
            for i in range(steps + 2):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                d_off, d_hits, st = g.enumerate_device(b_s.data_ptr(), n, 20, b_p.data_ptr(), 3, mismatches=3)
                torch.cuda.synchronize()
                dt = 1e3 * (time.perf_counter() - t0)
                if i >= 2:
                    rows.append((dt, st["ms_search"], st["n_hits"]))
            off = np.empty(n + 1, np.uint64)
            assert hip.hipMemcpy(off.ctypes.data, d_off, 8 * (n + 1), 2) == 0
            crc = zlib.crc32(off.tobytes())
            a = np.array(rows, dtype=np.float64)
            sh = g.last_sharing()
            print(json.dumps({"setting": s, "guides": n, "step_ms": round(a[:, 0].mean(), 2), "k_search_ms": round(a[:, 1].mean(), 2),
                              "hits_per_step": int(a[:, 2].mean()), "shared_items": sh["shared_items"], "packages": sh["packages"],
                              "guides_ordered_alone": sh["guides_ordered_device_wide_alone"], "crc32_offsets": f"{crc:08x}"}), flush=True)
    finally:
        g.close()


if __name__ == "__main__":
    main()
