#!/bin/bash
# bench.py with the strand tables built at another depth (GS_PREFIX_K: a build-time choice of the index builder)
# usage: bash tools/ab_prefix_k.sh "<bench args>" k [k ...]
mkdir -p gpurun_out
ARGS=$1; shift
for k in "$@"; do
  GS_PREFIX_K=$k timeout -k 10 300 python bench.py --cpu-sample 0 --steps 4 --warmup 1 --extra-rows off $ARGS > /tmp/abk.json 2> /tmp/abk.err || { echo "k=$k: failed"; tail -3 /tmp/abk.err; continue; }
  python3 - "$k" /tmp/abk.json "$ARGS" <<'PY'
import json, sys
j = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
r = j["roofline"]
print("k", sys.argv[1], sys.argv[3], round(j["value"]), "guides/s", round(j["ms_per_step"], 2), "ms/step, k_search", j["detail"]["k_search_ms_per_step"], "requests/guide", round(r["random_requests"]["per_guide"]), {a: round(b) for a, b in r["random_requests"]["by_kind_per_guide"].items()}, "resident GB", j["detail"].get("index_device_gb"))
PY
done
