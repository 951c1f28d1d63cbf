#!/bin/bash
# A/B builds of the tile ordering's kernels on the repeat-rich batch: per-kernel times (rocprofv3 --kernel-trace --stats) of
# bench.py --workload hg38rep with each library.  Usage (GPU box, repo root): bash tools/ab_tile.sh libgsamd.so libgsamd_x....so ...
set -o pipefail
export TMPDIR=/tmp
mkdir -p gpurun_out
for lib in "$@"; do
  if [ ! -f guidescan-cli_amd/$lib ]; then echo "$lib: not built"; continue; fi
  out=/tmp/abt_${lib%.so}
  rm -rf $out
  export GS_LIB_PATH=$PWD/guidescan-cli_amd/$lib
  timeout -k 10 400 rocprofv3 --kernel-trace --stats -f csv -d $out -- python3 bench.py --workload hg38rep --mismatches 3 --batch 20000 --steps 3 --warmup 2 --cpu-sample 0 --extra-rows off > $out.json 2> $out.err || { echo "$lib: failed"; tail -3 $out.err; continue; }
  f=$(find $out -name '*kernel_stats.csv' | head -1)
  python3 - "$lib" "$f" "$out.json" <<'PY'
import csv, json, sys
j = json.loads(open(sys.argv[3]).read().strip().splitlines()[-1])
rows = {r[0][:40]: r for r in csv.reader(open(sys.argv[2]))}
out = {}
for k, r in rows.items():
    if k.startswith(("k_to_", "k_search", "void k_to_", "k_share", "void k_share", "k_order", "k_locate")):
        out[k.replace("void ", "")[:28]] = round(float(r[3]) / 1e6, 3)   # AverageNs -> ms
print(sys.argv[1], "step", round(j["ms_per_step"], 2), "ms", json.dumps(out))
PY
done
