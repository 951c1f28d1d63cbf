#!/bin/bash
# the repeat-rich batch's k_search in the one-launch form with each library given
for lib in "$@"; do
  echo "== $lib"
  GS_LIB_PATH=$PWD/guidescan-cli_amd/$lib timeout -k 10 200 python tools/rep_share_sweep.py hg38rep 20000 3 6 GS_HEAVY=1 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    j = json.loads(l); print('   hg38rep 20k', j['setting'], 'k_search', j['k_search_ms'], j['k_search_ms_min_max'], 'step', j['step_ms'], j['crc32_last_batch'])"
done
