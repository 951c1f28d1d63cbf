#!/usr/bin/env python3
"""profiles/traffic.json from PMC summaries: one record per (workload, batch, mismatches) with the
FETCH_SIZE / WRITE_SIZE bytes of the timed k_search dispatch and a hash of the kernel sources
(bench.py only reports `roofline.traffic` when the hash matches the sources it runs).

    tools/make_traffic_json.py TAG WORKLOAD BATCH M FETCH.json WRITE.json [SQ.json]

FETCH.json / WRITE.json are tools/pmc_summary.py outputs of `rocprofv3 --pmc FETCH_SIZE` /
`--pmc WRITE_SIZE` passes of `bench.py --steps 1 --warmup 0 --cpu-sample 0` (tools/profile_round.sh);
the search dispatches there are: the reference-order walk sample (k_search_walk), the counting pass
(k_search_count), the timed step, side steps - the first k_search_fast dispatch after the counting
pass is the timed step.  Counters are in KB."""
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402


def main():
    tag, workload, batch, m, ffile, wfile = sys.argv[1:7]

    def timed_dispatch(d):
        """the timed step's search launch: the LAST launch of the heavy instantiation when the run has one (a handle
        takes it from its second batch of a repeat-rich shape on: profile with --steps 3 --warmup 2), else the first
        k_search_fast dispatch after the counting pass (k_search_count)"""
        heavy = sorted(d.get("k_search_heavy", []) + d.get("k_search_heavy_pd", []), key=lambda e: e["dispatch_id"])
        if heavy:
            return heavy[-1], "the last k_search_heavy dispatch"
        if d.get("k_seed_b") and d.get("k_seed_a"):
            # the two-launch form (gs_seed.hip): the first k_seed_b / k_seed_a pair after the counting pass, counters and
            # durations added up (one search = the two launches)
            after = max((e["dispatch_id"] for e in d.get("k_seed_count_a", []) + d.get("k_seed_count_b", [])), default=-1)
            b = sorted([e for e in d["k_seed_b"] if e["dispatch_id"] > after], key=lambda e: e["dispatch_id"])[0]
            a = sorted([e for e in d["k_seed_a"] if e["dispatch_id"] > b["dispatch_id"]], key=lambda e: e["dispatch_id"])[0]
            row = {k: (a[k] + b[k]) if isinstance(a[k], (int, float)) and k not in ("dispatch_id", "vgpr", "lds") else a[k] for k in a}
            row["per_launch"] = {"k_seed_b": b, "k_seed_a": a}
            return row, "the first k_seed_b + k_seed_a pair after the counting pass (counters and durations added)"
        counting = d.get("k_search_count", []) + d.get("k_search_count_pd", [])
        after = max((e["dispatch_id"] for e in counting), default=-1)
        rows = sorted([e for e in d.get("k_search_fast", []) + d.get("k_search_fast_pd", []) if e["dispatch_id"] > after],
                      key=lambda e: e["dispatch_id"])
        return rows[0], "the first k_search_fast dispatch after the counting pass k_search_count"

    def pick(path, counter, scale=1024.0):
        row, which = timed_dispatch(json.loads(Path(path).read_text()))
        return row[counter] * scale, row["duration_ms"], which
    fetch, dur_f, which = pick(ffile, "FETCH_SIZE")
    write, dur_w, _ = pick(wfile, "WRITE_SIZE")
    issue = None
    if len(sys.argv) > 7 and Path(sys.argv[7]).exists():   # the SQ pass: instruction issue and waiting of the same dispatch
        valu, dur_s, _ = pick(sys.argv[7], "SQ_INSTS_VALU", 1.0)
        salu, _, _ = pick(sys.argv[7], "SQ_INSTS_SALU", 1.0)
        issue = {"valu_wave_instructions": valu, "salu_wave_instructions": salu, "duration_ms": dur_s}
        try:
            wc, _, _ = pick(sys.argv[7], "SQ_WAVE_CYCLES", 1.0)
            wa, _, _ = pick(sys.argv[7], "SQ_WAIT_ANY", 1.0)
            issue["wave_cycles"] = wc
            issue["wait_any_share"] = wa / wc if wc else None
        except KeyError:
            pass
    # the memory side in REQUESTS (tools/gather_calib: FETCH_SIZE tallies every read request as 64 bytes whether it brings
    # 16, 64 or 128; the L2's read requests to the fabric by size say what came): optional passes
    ea = {}
    for path in sys.argv[8:]:
        if not Path(path).exists():
            continue
        row, _ = timed_dispatch(json.loads(Path(path).read_text()))
        for k, v in row.items():
            if k.startswith("TCC_"):
                ea[k] = v
    out = ROOT / "profiles" / "traffic.json"
    recs = json.loads(out.read_text()) if out.exists() else []
    recs = [r for r in recs if not (r["workload"] == workload and r["batch"] == int(batch) and r["mismatches"] == int(m))]
    rec = {"workload": workload, "batch": int(batch), "mismatches": int(m), "fetch_bytes": int(fetch),
           "write_bytes": int(write), "duration_ms_under_pmc": [dur_f, dur_w],
           "issue": issue,
           "kernel_sha": bench.kernel_stamp(),
           "source": f"profiles/{tag}_pmc_fetch_size.json + {tag}_pmc_write_size.json: rocprofv3 --pmc FETCH_SIZE / "
                     f"WRITE_SIZE (separate passes, tools/profile_round.sh) of `bench.py --cpu-sample 0`, {which}"}
    if ea:
        rd = ea.get("TCC_EA0_RDREQ_sum")
        r32, r64, r128 = ea.get("TCC_EA0_RDREQ_32B_sum"), ea.get("TCC_EA0_RDREQ_64B_sum"), ea.get("TCC_EA0_RDREQ_128B_sum")
        rec["ea"] = ea
        rec["read_requests"] = rd
        if rd is not None and r32 is not None:
            if r64 is not None and r128 is not None:
                rec["read_bytes_corrected"] = 32.0 * r32 + 64.0 * r64 + 128.0 * r128
                rec["read_bytes_corrected_from"] = "32 x RDREQ_32B + 64 x RDREQ_64B + 128 x RDREQ_128B"
            else:
                # without the per-size counters: every request that is not a 32-byte one brings the block it was asked
                # for - bounds: 64 bytes each (what FETCH_SIZE says) .. 128 bytes each
                rec["read_bytes_corrected_bounds"] = [32.0 * r32 + 64.0 * (rd - r32), 32.0 * r32 + 128.0 * (rd - r32)]
    recs.append(rec)
    out.write_text(json.dumps(recs, indent=1))
    print(json.dumps(recs[-1]))


if __name__ == "__main__":
    main()
