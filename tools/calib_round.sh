#!/bin/bash
# The request ceiling's calibration (tools/gather_calib.hip): timing, then PMC passes of the same program - one row per
# (pattern, table, region) dispatch.  Usage (GPU box, repo root): bash tools/calib_round.sh <tag>   ->  gpurun_out/<tag>_gather_calib*.{txt,json}
set -o pipefail
TAG=${1:-rXX}
export TMPDIR=/tmp
mkdir -p gpurun_out
tools/gather_calib 40960 > gpurun_out/${TAG}_gather_calib.txt 2>&1 || { tail -3 gpurun_out/${TAG}_gather_calib.txt; exit 1; }
rocprofv3 -L 2>/dev/null | grep -o "\b\(TCC_[A-Z0-9_]*\|TCP_[A-Z0-9_]*TLB[A-Z0-9_]*\|[A-Z0-9_]*UTCL[A-Z0-9_]*\|FETCH_SIZE\|WRITE_SIZE\|MALL[A-Z0-9_]*\)\b" | sort -u | tr '\n' ' ' > gpurun_out/${TAG}_counters_avail.txt
for pass in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_REQ_sum TCC_READ_sum"; do
  name=$(echo $pass | cut -d' ' -f1 | tr 'A-Z' 'a-z')
  rm -rf /tmp/calib_$name
  rocprofv3 --pmc $pass -f csv -d /tmp/calib_$name -- tools/gather_calib 40960 > /tmp/calib_$name.txt 2> /tmp/calib_$name.err || { echo "pmc $name failed"; tail -3 /tmp/calib_$name.err; continue; }
  python3 - /tmp/calib_$name gpurun_out/${TAG}_gather_calib.txt > gpurun_out/${TAG}_gather_calib_pmc_$name.json <<'PY'
import csv, glob, json, sys
rows = {}
for f in glob.glob(sys.argv[1] + "/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_calib" not in r["Kernel_Name"]:
            continue
        e = rows.setdefault(int(r["Dispatch_Id"]), {"kernel": r["Kernel_Name"].split("(")[0].replace("void ", "")})
        e[r["Counter_Name"]] = float(r["Counter_Value"])
# dispatches come in pairs (warm-up + timed) in the order of the program's lines
lines = [json.loads(l) for l in open(sys.argv[2]) if l.startswith("{")]
ids = sorted(rows)
out = []
for i, l in enumerate(lines):
    if 2 * i + 1 < len(ids):
        out.append(dict(l, counters=rows[ids[2 * i + 1]]))
print(json.dumps(out, indent=1))
PY
  echo "pmc $name ok"
done
cat gpurun_out/${TAG}_gather_calib.txt
