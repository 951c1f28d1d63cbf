#!/bin/bash
# Profiles of one bench workload for profiles/: kernel-trace stats, then separate PMC passes
# (FETCH_SIZE, WRITE_SIZE, SQ counters), each reduced to a small summary under gpurun_out/prof_summary/,
# and the traffic.json record bench.py reads (stamped with the kernel sources' hash).
# Usage (GPU box, repo root): bash tools/profile_round.sh <tag> [workload=hg38] [mismatches=3] [batch=0 (default)] [steps=1] [warmup=0]
# (a repeat-rich workload: steps=3 warmup=2 - the handle takes the heavy instantiation of k_search from its second batch on)
set -o pipefail
TAG=${1:-rXX}
WL=${2:-hg38}
M=${3:-3}
BATCH=${4:-0}
STEPS=${5:-1}
WARM=${6:-0}
NAME=${TAG}_${WL}_m${M}
OUT=/tmp/prof_$NAME   # raw rocprof output stays off gpurun_out/ (64 MiB limit)
SUM=gpurun_out/prof_summary
rm -rf $OUT   # (a box may be one an earlier call ran on: summaries must not pick up its files)
mkdir -p $OUT $SUM
export TMPDIR=/tmp
ARGS="--workload $WL --mismatches $M --batch $BATCH --cpu-sample 0 --extra-rows off"
rocprofv3 --kernel-trace --stats -f csv -d $OUT/stats -- python3 bench.py $ARGS --steps $(( STEPS > 3 ? STEPS : 3 )) --warmup $(( WARM > 1 ? WARM : 1 )) > $SUM/${NAME}_bench_under_rocprof.json 2> $OUT/stats.err
echo "stats rc=$?"
f=$(find $OUT/stats -name '*kernel_stats.csv' | head -1)
# every kernel of the run (rocPRIM's sorts and scans included), names cut to 110 characters; the index
# builder's kernels (k_sa_*, k_ctx_build, ...) are in there too: they run once, before the timed steps
if [ -n "$f" ]; then python3 - "$f" > $SUM/${NAME}_kernel_stats.csv <<'PY'
import csv, sys
w = csv.writer(sys.stdout)
for row in csv.reader(open(sys.argv[1])):
    row[0] = row[0][:110]
    w.writerow(row)
PY
fi
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_ANY" \
            "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" \
            "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS"; do
  name=$(echo $pass | cut -d' ' -f1 | tr 'A-Z' 'a-z')
  rocprofv3 --pmc $pass -f csv --kernel-include-regex 'k_search|k_seed|k_order|k_locate|k_big2|k_score|k_to_|k_share' -d $OUT/pmc_$name -- python3 bench.py $ARGS --steps $STEPS --warmup $WARM > $OUT/pmc_$name.json 2> $OUT/pmc_$name.err
  echo "pmc $name rc=$?"
  python3 tools/pmc_summary.py $OUT/pmc_$name k_search k_seed k_order k_locate k_big2 k_score k_to_ k_share > $SUM/${NAME}_pmc_$name.json
done
B=$(python3 -c "import json;print(json.loads(open('$SUM/${NAME}_bench_under_rocprof.json').read().strip().splitlines()[-1])['config']['guides_per_step_per_gpu'])")
python3 tools/make_traffic_json.py $NAME $WL $B $M $SUM/${NAME}_pmc_fetch_size.json $SUM/${NAME}_pmc_write_size.json $SUM/${NAME}_pmc_sq_wave_cycles.json \
  $SUM/${NAME}_pmc_tcc_ea0_rdreq_sum.json $SUM/${NAME}_pmc_tcc_ea0_rdreq_64b_sum.json && cp profiles/traffic.json $SUM/traffic.json
ls -la $SUM
