#!/bin/bash
# Kernel-trace summary of one bench workload with EVERY kernel (rocPRIM's included), names cut to 120 characters.
# Usage (GPU box, repo root): bash tools/trace_stats.sh <tag> <workload> <mismatches> [batch] [extra bench args]
set -o pipefail
TAG=$1; WL=$2; M=$3; BATCH=${4:-0}; shift 4
NAME=${TAG}_${WL}_m${M}
OUT=/tmp/trace_$NAME
SUM=gpurun_out/prof_summary
mkdir -p $OUT $SUM
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -f csv -d $OUT -- python3 bench.py --workload $WL --mismatches $M --batch $BATCH --cpu-sample 0 "$@" \
  > $SUM/${NAME}_trace_bench.json 2> $OUT/err.log
echo "rc=$? $(tail -c 300 $OUT/err.log)"
f=$(find $OUT -name '*kernel_stats.csv' | head -1)
python3 - "$f" > $SUM/${NAME}_all_kernel_stats.csv <<'PY'
import csv, sys
r = list(csv.reader(open(sys.argv[1])))
w = csv.writer(sys.stdout)
for row in r:
    row[0] = row[0][:120]
    w.writerow(row)
PY
head -40 $SUM/${NAME}_all_kernel_stats.csv
