#!/bin/bash
# Profiles of the default bench workload for profiles/: kernel-trace stats, then separate PMC passes
# (FETCH_SIZE, WRITE_SIZE, SQ counters), each reduced to a small summary under gpurun_out/prof_summary/.
# Usage (GPU box, repo root): bash tools/profile_round.sh <tag>
set -o pipefail
TAG=${1:-rXX}
OUT=/tmp/prof_$TAG   # raw rocprof output stays off gpurun_out/ (64 MiB limit)
SUM=gpurun_out/prof_summary
mkdir -p $OUT $SUM
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -f csv -d $OUT/stats -- python3 bench.py --cpu-sample 0 > $SUM/${TAG}_bench_under_rocprof.json 2> $OUT/stats.err
echo "stats rc=$?"
f=$(find $OUT/stats -name '*kernel_stats.csv' | head -1)
if [ -n "$f" ]; then (head -1 "$f"; grep -E '"?k_(search|order|locate|prepare|scan|score|km_|collect|gather|patch|huge)' "$f") > $SUM/${TAG}_hg38_kernel_stats.csv; fi
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_ANY"; do
  name=$(echo $pass | cut -d' ' -f1 | tr 'A-Z' 'a-z')
  rocprofv3 --pmc $pass -f csv --kernel-include-regex 'k_search|k_order|k_locate' -d $OUT/pmc_$name -- python3 bench.py --steps 1 --warmup 0 --cpu-sample 0 > $OUT/pmc_$name.json 2> $OUT/pmc_$name.err
  echo "pmc $name rc=$?"
  python3 tools/pmc_summary.py $OUT/pmc_$name k_search k_order k_locate > $SUM/${TAG}_hg38_pmc_$name.json
done
find $OUT -type f | head -30
ls -la $SUM
