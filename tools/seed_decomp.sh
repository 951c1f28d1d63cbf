#!/bin/bash
# Where the two seeding launches' time goes: the same batch with the verification off (GS_DBG_SKIP=1), with no seed kept (2),
# with no recipe at all (4: what an item costs before its first seed).  Kernel times from rocprofv3 --kernel-trace --stats.
# Usage (GPU box, repo root): bash tools/seed_decomp.sh <workload> <batch> <m> <form> [skips, default "0 1 2 4"]
set -o pipefail
WL=$1; BATCH=$2; M=$3; FORM=$4; SKIPS=${5:-"0 1 2 4"}
export TMPDIR=/tmp
for S in $SKIPS; do
  OUT=/tmp/sdec_$S
  rm -rf $OUT; mkdir -p $OUT
  GS_DBG_SKIP=$S rocprofv3 --kernel-trace --stats -f csv -d $OUT -- python3 tools/seed_forms.py $WL $BATCH $M $FORM > $OUT/out.txt 2> $OUT/err.txt || { tail -5 $OUT/err.txt; exit 1; }
  f=$(find $OUT -name '*kernel_stats.csv' | head -1)
  echo "GS_DBG_SKIP=$S $(tail -1 $OUT/out.txt)"
  python3 - "$f" <<'PY'
import csv, sys
for row in csv.DictReader(open(sys.argv[1])):
    n = row["Name"]
    if any(k in n for k in ("k_seed", "k_search", "k_describe", "k_sched")):
        print("   ", n.split("(")[0][:40], "calls", row["Calls"], "avg_us", round(float(row["AverageNs"]) / 1e3, 1), "min_us", round(float(row["MinNs"]) / 1e3, 1))
PY
done
