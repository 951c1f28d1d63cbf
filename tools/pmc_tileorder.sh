#!/bin/bash
# PMC passes for the tile-ordering kernels on hg38rep
set -o pipefail
OUT=/tmp/pmc_to; SUM=gpurun_out/prof_summary; mkdir -p $OUT $SUM
export TMPDIR=/tmp
ARGS="--workload hg38rep --mismatches 3 --cpu-sample 0 --steps 1 --warmup 0"
i=0
for pass in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_ANY" "SQ_BUSY_CYCLES SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $pass -f csv --kernel-include-regex 'k_search|k_to_' -d $OUT/p$i -- python3 bench.py $ARGS > $OUT/p$i.json 2> $OUT/p$i.err
  echo "pass $i rc=$?"
  python3 tools/pmc_summary.py $OUT/p$i k_search k_to_ > $SUM/r04_hg38rep_m3_pmc_pass$i.json
done
