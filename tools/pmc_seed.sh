#!/bin/bash
# PMC passes of the two seeding launches (k_seed_b, k_seed_a) - or of k_search with form 0 - on one batch at a workload's size.
# Usage (GPU box, repo root): bash tools/pmc_seed.sh <tag> <workload> <batch> <m> <form>   (GS_* switches apply)
set -o pipefail
TAG=$1; WL=$2; BATCH=$3; M=$4; FORM=$5
export TMPDIR=/tmp
mkdir -p gpurun_out
i=0
for CNT in "TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum" "SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_WAVE_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  OUT=/tmp/pmcs_${TAG}_$i
  rm -rf $OUT; mkdir -p $OUT
  rocprofv3 --pmc $CNT -f csv --kernel-include-regex 'k_seed|k_search' -d $OUT -- python3 tools/seed_forms.py $WL $BATCH $M $FORM > $OUT/out.txt 2> $OUT/err.txt || { tail -5 $OUT/err.txt; exit 1; }
  python3 tools/pmc_summary.py $OUT k_seed k_search > gpurun_out/pmcs_${TAG}_$i.json
  python3 - gpurun_out/pmcs_${TAG}_$i.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
for k, rows in d.items():
    for r in rows[-1:]:
        print(k, json.dumps({a: (round(b, 3) if isinstance(b, float) else b) for a, b in r.items()}))
PY
  i=$((i+1))
done
