#!/bin/bash
# what the publishing code costs the headline batch: per-kernel times (rocprofv3 --kernel-trace --stats) of the default bench,
# plain form against the two-launch form (GS_SPLIT_SHARE=2: k_search_pub_pd + k_search_heavy_pd without items)
set -o pipefail
export TMPDIR=/tmp
mkdir -p gpurun_out
for mode in plain split; do
  out=/tmp/abp_$mode
  rm -rf $out
  if [ $mode = split ]; then export GS_SPLIT_SHARE=2; else unset GS_SPLIT_SHARE; fi
  timeout -k 10 400 rocprofv3 --kernel-trace --stats -f csv -d $out -- python3 bench.py --steps 4 --warmup 1 --cpu-sample 0 --extra-rows off > $out.json 2> $out.err || { echo "$mode: failed"; tail -3 $out.err; continue; }
  f=$(find $out -name '*kernel_stats.csv' | head -1)
  python3 - "$mode" "$f" "$out.json" <<'PY'
import csv, json, sys
j = json.loads(open(sys.argv[3]).read().strip().splitlines()[-1])
out = {}
for r in csv.reader(open(sys.argv[2])):
    if r[0].startswith("k_search") and "count" not in r[0] and "walk" not in r[0]:
        out[r[0][:24]] = {"calls": int(r[1]), "avg_ms": round(float(r[3]) / 1e6, 3)}
print(sys.argv[1], "step", round(j["ms_per_step"], 2), "ms, library's k_search", j["detail"]["k_search_ms_per_step"], json.dumps(out))
PY
done
