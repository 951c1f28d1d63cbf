#!/usr/bin/env python3
"""The paths beside the headline at hg38 size, each timed and compared line by line with the compiled reference
(oracle/_ref/gs_ref_enumerate on the box's host cores, index files written through its own SDSL containers):

  cas12a   TTTN + --start, 20-mers and 23-mers (2L + 3P = 52 and 58 key bits) through the fast path
  bulges   -m 1 --rna-bulges 1 --dna-bulges 1 through the general path (index.hpp:250-375)
  odd      guides with an N in them at -m 3 through the general path (index.hpp:218-247)

Usage (GPU box, repo root): python tools/general_bench.py [--guides 64] [--cas 1024] [--skip-ref]
Prints one JSON object; profiles/r04_general_paths.json is a run of it."""
import argparse
import json
import os
import subprocess
import sys
import time
from importlib import import_module
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
for p in (str(ROOT), str(ROOT / "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def ref_lines(h, tag, rows, m, rna=0, dna=0, start=False, threads=None):
    """rows: (id, sequence, pam).  The compiled reference's CSV data lines, sorted, and its own timer's seconds."""
    import re
    full = import_module("test_gpu_fullsize")
    synth = import_module("guidescan-cli_amd.synth")
    prefix = h.reference_prefix()
    kcsv, out = os.path.join(h.dir, tag + ".kmers.csv"), os.path.join(h.dir, tag + ".out.csv")
    synth.write_kmers_csv(kcsv, [r[0] for r in rows], [r[1] for r in rows], [r[2] for r in rows], [h.names[0]] * len(rows),
                          [1] * len(rows), ["+"] * len(rows))
    env = dict(os.environ, GS_REF_THREADS=str(threads or os.cpu_count() or 8))
    r = subprocess.run([str(full.SHIM), prefix, kcsv, out, "csv", "complete", str(m), str(rna), str(dna), "-1", "-1",
                        "1" if start else "0"], env=env, check=True, timeout=3000, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
    secs = float(re.search(r"kmers in ([0-9.eE+-]+) s", r.stderr.decode()).group(1))
    with open(out) as f:
        lines = f.read().splitlines()[1:]
    os.unlink(out)
    return sorted(lines), secs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--guides", type=int, default=64, help="guides of the general-path rows")
    ap.add_argument("--cas", type=int, default=1024, help="guides of each Cas12a row")
    ap.add_argument("--skip-ref", action="store_true", help="rates only, no comparison with the compiled reference")
    args = ap.parse_args()
    api = import_module("guidescan-cli_amd.api")
    synth = import_module("guidescan-cli_amd.synth")
    full = import_module("test_gpu_fullsize")
    h = full.Hg38()
    out = {"genome_bp": int(h.text.shape[0]), "index_build_s": round(h.t_build, 1), "host_threads": os.cpu_count()}
    try:
        gs = h.gs
        # ---- Cas12a: TTTN + protospacer, PAM at the 5' end -------------------------------------------------
        for L in (20, 23):
            seqs, pams = cas_guides(synth, h.text, args.cas, L, 60 + L)
            row = {"guides": args.cas, "L": L, "pam": "TTTN --start", "key_bits": 2 * L + 12}
            h.gidx.enumerate(seqs[:64], pams[:64], mismatches=3, start=True)   # derived tables + workspace
            t0 = time.perf_counter()
            off, hits, st = h.gidx.enumerate(seqs, pams, mismatches=3, start=True)
            dt = time.perf_counter() - t0
            ctr = h.gidx.last_counters()
            row.update(seconds=round(dt, 4), guides_per_s=args.cas / dt, k_search_ms=round(st["ms_search"], 2),
                       hits=int(off[-1]), items_two_sided=ctr["items_two_sided"], items_one_sided=ctr["items_one_sided"])
            if not args.skip_ref:
                ids = [f"c{L}_{i}" for i in range(args.cas)]
                want, secs = ref_lines(h, f"cas{L}", [(ids[i], seqs[i].tobytes().decode(), "TTTN") for i in range(args.cas)], 3,
                                       start=True)
                _, spec = h.gidx.score(gs, seqs, 4, off, hits, want_cfd=False, start=True)
                got = []
                for i in range(args.cas):
                    got += api.format_guide(gs, ids[i], seqs[i].tobytes().decode(), "TTTN", True, hits[off[i]:off[i + 1]], 3,
                                            specificity=spec[i], start=True).splitlines()
                got.sort()
                row.update(reference_seconds=round(secs, 2), reference_guides_per_s=args.cas / secs, lines=len(want),
                           identical_to_reference=got == want)
            out[f"cas12a_{L}mers"] = row
            print(f"[general_bench] cas12a {L}-mers: {row}", file=sys.stderr, flush=True)
        # ---- the general path -----------------------------------------------------------------------------------
        n = args.guides
        seqs, pams, _, _ = synth.sample_guides(h.text, n, seed=77)
        odd = seqs.copy()
        rng = np.random.default_rng(5)
        for i in range(n):
            odd[i, int(rng.integers(0, 20))] = ord("N")
        for name, s, m, rna, dna in (("bulges_m1_rna1_dna1", seqs, 1, 1, 1), ("guides_with_N_m3", odd, 3, 0, 0)):
            row = {"guides": n, "mismatches": m, "rna_bulges": rna, "dna_bulges": dna}
            h.gidx.enumerate_general(s[:4], pams[:4], mismatches=m, rna_bulges=rna, dna_bulges=dna)
            print(f"[general_bench] {name}: {n} guides ...", file=sys.stderr, flush=True)
            t0 = time.perf_counter()
            off, hx = h.gidx.enumerate_general(s, pams, mismatches=m, rna_bulges=rna, dna_bulges=dna)
            dt = time.perf_counter() - t0
            row.update(seconds=round(dt, 3), guides_per_s=n / dt, hits=int(off[-1]), hits_per_s=int(off[-1]) / dt)
            if not args.skip_ref:
                ids = [f"{name[:3]}{i}" for i in range(n)]
                want, secs = ref_lines(h, name[:6], [(ids[i], s[i].tobytes().decode(), "NGG") for i in range(n)], m, rna, dna)
                got = []
                for i in range(n):
                    got += api.format_guide_ex(gs, ids[i], s[i].tobytes().decode(), "NGG", True, hx[off[i]:off[i + 1]], m).splitlines()
                got.sort()
                row.update(reference_seconds=round(secs, 2), reference_guides_per_s=n / secs, lines=len(want),
                           identical_to_reference=got == want, speedup_vs_reference=(n / dt) / (n / secs))
            out[name] = row
            print(f"[general_bench] {name}: {row}", file=sys.stderr, flush=True)
    finally:
        h.close()
    print(json.dumps(out), flush=True)


def cas_guides(synth, text, n, L, seed):
    """protospacers read off the + strand behind TTTN (PAM at the 5' end: site = PAM + protospacer)"""
    rng = np.random.default_rng(seed)
    seqs = np.empty((n, L), dtype=np.uint8)
    got = 0
    acgt = np.zeros(256, dtype=bool)
    acgt[list(b"ACGT")] = True
    T = ord("T")
    while got < n:
        cand = rng.integers(0, text.shape[0] - (L + 4), size=200_000)
        ok = (text[cand] == T) & (text[cand + 1] == T) & (text[cand + 2] == T) & acgt[text[cand + 3]]
        cand = cand[ok]
        w = text[cand[:, None] + 4 + np.arange(L)[None, :]]
        good = acgt[w].all(axis=1)
        w = w[good][: n - got]
        seqs[got:got + w.shape[0]] = w
        got += w.shape[0]
    pams = np.tile(np.frombuffer(b"TTTN", np.uint8), (n, 1))
    return seqs, pams


if __name__ == "__main__":
    main()
