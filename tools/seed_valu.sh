#!/bin/bash
# Vector / scalar instructions of the two seeding launches with the verification off (GS_DBG_SKIP=1) and with no recipe at all
# (4): where the instructions of an item go.  Usage (GPU box, repo root): bash tools/seed_valu.sh <workload> <batch> <m>
set -o pipefail
WL=$1; BATCH=$2; M=$3
export TMPDIR=/tmp
for S in 0 1 4; do
  OUT=/tmp/sval_$S
  rm -rf $OUT; mkdir -p $OUT
  GS_SEED_TAKE=$([ $S = 4 ] && echo 8 || echo 1) GS_DBG_SKIP=$S rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_ANY -f csv --kernel-include-regex 'k_seed' -d $OUT -- python3 tools/seed_forms.py $WL $BATCH $M 2 > $OUT/out.txt 2> $OUT/err.txt || { tail -5 $OUT/err.txt; exit 1; }
  python3 tools/pmc_summary.py $OUT k_seed > /tmp/sval_$S.json
  python3 - /tmp/sval_$S.json $S <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
for k, rows in d.items():
    r = rows[-1]
    print("skip", sys.argv[2], k, {a: (round(b / 1e6, 1) if a.startswith("SQ_") else round(b, 2)) for a, b in r.items() if a.startswith("SQ_") or a == "duration_ms"}, "(SQ_* in millions)")
PY
done
