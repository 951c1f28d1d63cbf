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

    def pick(path, counter):
        """the timed step = the first k_search_fast dispatch after the counting pass (k_search_count)"""
        d = json.loads(Path(path).read_text())
        # k_search_fast / k_search_count, or their instantiation for batches served by pair + deep tables (_pd)
        counting = d.get("k_search_count", []) + d.get("k_search_count_pd", [])
        after = max((e["dispatch_id"] for e in counting), default=-1)
        rows = sorted([e for e in d.get("k_search_fast", []) + d.get("k_search_fast_pd", []) if e["dispatch_id"] > after],
                      key=lambda e: e["dispatch_id"])
        return rows[0][counter] * 1024.0, rows[0]["duration_ms"]
    fetch, dur_f = pick(ffile, "FETCH_SIZE")
    write, dur_w = pick(wfile, "WRITE_SIZE")
    issue = None
    if len(sys.argv) > 7 and Path(sys.argv[7]).exists():   # the SQ pass: instruction issue of the same dispatch
        valu, dur_s = pick(sys.argv[7], "SQ_INSTS_VALU")
        salu, _ = pick(sys.argv[7], "SQ_INSTS_SALU")
        issue = {"valu_wave_instructions": valu / 1024.0, "salu_wave_instructions": salu / 1024.0, "duration_ms": dur_s}
    out = ROOT / "profiles" / "traffic.json"
    recs = json.loads(out.read_text()) if out.exists() else []
    recs = [r for r in recs if not (r["workload"] == workload and r["batch"] == int(batch) and r["mismatches"] == int(m))]
    recs.append({"workload": workload, "batch": int(batch), "mismatches": int(m), "fetch_bytes": int(fetch),
                 "write_bytes": int(write), "duration_ms_under_pmc": [dur_f, dur_w],
                 "issue": issue,
                 "kernel_sha": bench.kernel_stamp(),
                 "source": f"profiles/{tag}_pmc_fetch_size.json + {tag}_pmc_write_size.json: rocprofv3 --pmc FETCH_SIZE / "
                           f"WRITE_SIZE (separate passes, tools/profile_round.sh) of `bench.py --steps 1 --warmup 0 "
                           f"--cpu-sample 0`, the timed k_search_fast dispatch (the first after the counting pass k_search_count)"})
    out.write_text(json.dumps(recs, indent=1))
    print(json.dumps(recs[-1]))


if __name__ == "__main__":
    main()
