#!/bin/bash
# One PMC pass of the search kernel of a short bench run, summarised as JSON on stdout.
# Usage (GPU box, repo root): bash tools/pmc_quick.sh <tag> "<counters>" [bench.py args ...]   (environment GS_* switches apply)
set -o pipefail
TAG=$1; CNT=$2; shift 2
export TMPDIR=/tmp
OUT=/tmp/pmcq_$TAG
rm -rf $OUT; mkdir -p $OUT gpurun_out
rocprofv3 --pmc $CNT -f csv --kernel-include-regex 'k_search' -d $OUT -- python3 bench.py --cpu-sample 0 --steps 1 --warmup 0 --extra-rows off "$@" > $OUT/bench.json 2> $OUT/err.txt || { tail -5 $OUT/err.txt; exit 1; }
python3 tools/pmc_summary.py $OUT k_search > gpurun_out/pmcq_$TAG.json
python3 - gpurun_out/pmcq_$TAG.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
for k, rows in d.items():
    if "count" in k or "walk" in k:
        continue
    for r in rows[-2:]:
        print(k, json.dumps({a: (round(b, 3) if isinstance(b, float) else b) for a, b in r.items()}))
PY
