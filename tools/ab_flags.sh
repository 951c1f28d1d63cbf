#!/bin/bash
# k_search with each library given (variants of the same sources): the headline batch, then the repeat-rich batch in the one-launch form
mkdir -p gpurun_out
for lib in "$@"; do
  echo "== $lib"
  GS_LIB_PATH=$PWD/guidescan-cli_amd/$lib timeout -k 10 200 python tools/rep_share_sweep.py hg38 1000000 3 3 default 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    j = json.loads(l); print('   hg38 1M', j['setting'], 'k_search', j['k_search_ms'], 'step', j['step_ms'], j['crc32_last_batch'])"
  if [ -z "$AB_NO_REP" ]; then
  GS_LIB_PATH=$PWD/guidescan-cli_amd/$lib timeout -k 10 200 python tools/rep_share_sweep.py hg38rep 20000 3 4 GS_HEAVY=1 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    j = json.loads(l); print('   hg38rep 20k', j['setting'], 'k_search', j['k_search_ms'], 'step', j['step_ms'], j['crc32_last_batch'])"
  fi
done
