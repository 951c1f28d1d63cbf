#!/bin/bash
# bench.py with alternative builds of the library (GS_LIB_PATH): usage tools/sweep_weu.sh LIB... ; m = 3 and 6
for lib in "$@"; do
  for cfg in "3 1000000" "6 20000"; do set -- $cfg
    GS_LIB_PATH=$PWD/$lib timeout -k 10 300 python bench.py --mismatches $1 --batch $2 --steps 3 --warmup 1 --cpu-sample 0 > gpurun_out/w.json 2> gpurun_out/w.err
    python3 -c "
import json
j=json.loads(open('gpurun_out/w.json').read().strip().splitlines()[-1]); print('$lib', $1, round(j['value']), round(j['ms_per_step'],2), j['detail']['k_search_ms_per_step'], j['detail']['hits_per_guide'])"
  done
done
