#!/usr/bin/env python3
"""k_search's sharing of heavy items among waves (gs_search.hip, `shq`) on one resident index: the same batches under
several (GS_SHARE_MIN, GS_SHARE_MAX) settings - the library's switches are per handle (gs_index_set_option), so one
index build serves the sweep.  Per setting: two warm-up steps, then `steps` fresh batches; step time by the wall clock,
k_search by the library's own events, the packages handed out, and a checksum of (offsets, hits) that must not move.
Usage (GPU box, repo root): python tools/rep_share_sweep.py [workload=hg38rep] [guides=20000] [m=3] [steps=4] [min:max ...]"""
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
    workload = sys.argv[1] if len(sys.argv) > 1 else "hg38rep"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
    m = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    steps = int(sys.argv[4]) if len(sys.argv) > 4 else 4
    settings = sys.argv[5:] or ["0:2048", "512:2048", "256:1024", "1024:4096", "128:512"]
    lens_name, _, probs = bench.WORKLOADS[workload]
    lengths = [synth.CHR1_LENGTH] if lens_name == "CHR1" else getattr(synth, lens_name)
    text, names, lengths = bench.make_workload_genome(synth, workload, lengths, probs)
    g = api.GenomeIndex.build(text, device=0)
    hip = bench_hip()
    try:
        seqs, pams, _, _ = synth.sample_guides(text, n * (steps + 2), seed=1000)
        d_s, d_p = torch.from_numpy(seqs).cuda(), torch.from_numpy(pams).cuda()
        used = []
        for si, s in enumerate(settings):
            for k in used:
                g.set_option(k, None)
            used.clear()
            if "=" in s or s == "default":   # KEY=V,KEY=V: any per-handle switches ("default": none - what a caller gets)
                kv = dict(x.split("=") for x in s.split(",")) if s != "default" else {}
                smin, smax = kv.get("GS_SHARE_MIN", "512"), kv.get("GS_SHARE_MAX", "2048")
            else:
                smin, smax = s.split(":")
                # share_min 0: the plain instantiation; else the heavy one (what a handle picks by itself after a batch with
                # heavy passes), with sharing from share_min row groups on
                kv = dict(GS_SHARE_MIN=smin, GS_SHARE_MAX=smax, GS_HEAVY="0" if int(smin) == 0 else "1")
            g.set_options(**kv)
            used.extend(kv)
            rows, crc = [], 0
            for i in range(steps + 2):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                d_off, d_hits, st = g.enumerate_device(d_s[i * n:].data_ptr(), n, 20, d_p[i * n:].data_ptr(), 3, mismatches=m)
                torch.cuda.synchronize()
                dt = 1e3 * (time.perf_counter() - t0)
                sh = g.last_sharing()
                if i >= 2:
                    rows.append((dt, st["ms_search"], st["n_hits"], sh["shared_items"], sh["packages"]))
                if i == steps + 1:   # checksum of the last batch: offsets and hit records as the library left them
                    off = np.empty(n + 1, np.uint64)
                    assert hip.hipMemcpy(off.ctypes.data, d_off, 8 * (n + 1), 2) == 0
                    crc = zlib.crc32(off.tobytes())
                    left, pos, buf = int(off[n]) * 16, 0, np.empty(1 << 28, np.uint8)
                    while left:
                        c = min(left, buf.size)
                        assert hip.hipMemcpy(buf.ctypes.data, d_hits + pos, c, 2) == 0
                        crc = zlib.crc32(buf[:c].tobytes(), crc)
                        pos += c
                        left -= c
            a = np.array(rows, dtype=np.float64)
            print(json.dumps({"workload": workload, "guides": n, "m": m, "setting": s, "share_min": int(smin), "share_max": int(smax),
                              "step_ms": round(a[:, 0].mean(), 2), "k_search_ms": round(a[:, 1].mean(), 2),
                              "k_search_ms_min_max": [round(a[:, 1].min(), 2), round(a[:, 1].max(), 2)],
                              "hits_per_step": int(a[:, 2].mean()), "shared_items": int(a[:, 3].mean()),
                              "packages": int(a[:, 4].mean()), "queue_packages": sh["queue_packages"],
                              "crc32_last_batch": f"{crc:08x}"}), flush=True)
    finally:
        g.close()


def bench_hip():
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    hip.hipMemcpy.restype = C.c_int
    return hip


if __name__ == "__main__":
    main()
