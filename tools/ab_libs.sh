#!/bin/bash
# the default bench with each of the libraries given (paths relative to the repo root), on one box: step and k_search times
mkdir -p gpurun_out
for lib in "$@"; do
  GS_LIB_PATH=$PWD/$lib timeout -k 10 300 python bench.py --cpu-sample 0 --steps 5 --warmup 1 --extra-rows off > /tmp/abl.json 2> /tmp/abl.err || { echo "$lib: failed"; tail -3 /tmp/abl.err; continue; }
  python3 - "$lib" /tmp/abl.json <<'PY'
import json, sys
j = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
print(sys.argv[1], round(j["value"]), "guides/s", round(j["ms_per_step"], 2), "ms/step, k_search", j["detail"]["k_search_ms_per_step"], "ms")
PY
done
