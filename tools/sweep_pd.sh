#!/bin/bash
# k_search_fast_pd at 8 (the shipped build), 7 and 6 waves per SIMD: profiles/r03_sweep_pd.txt.
# Builds the two variants next to the shipped library (HERE, where hipcc is: the GPU box only runs them), then - on the
# GPU box, repo root - times m <= 3, 4, 5 with each.  Usage: bash tools/sweep_pd.sh build | run
set -euo pipefail
ROOT=$(cd "$(dirname "$0")/.." && pwd)
CSRC=$ROOT/guidescan-cli_amd/csrc
case "${1:-run}" in
build)
  for w in 7 6; do
    make -s -C "$CSRC" lib OUT="$ROOT/guidescan-cli_amd/libgsamd_pd$w.so" BUILD=build_pd$w EXTRA=-DGS_WAVES_EU_PD=$w
  done
  ls -la "$ROOT"/guidescan-cli_amd/libgsamd*.so
  ;;
run)
  mkdir -p "$ROOT/gpurun_out"
  for lib in libgsamd.so libgsamd_pd7.so libgsamd_pd6.so; do
    if [ ! -f "$ROOT/guidescan-cli_amd/$lib" ]; then echo "$lib not built (bash tools/sweep_pd.sh build): skipped"; continue; fi
    for cfg in "3 1000000" "4 200000" "5 100000"; do
      set -- $cfg
      if ! GS_LIB_PATH=$ROOT/guidescan-cli_amd/$lib timeout -k 10 300 python "$ROOT/bench.py" --mismatches "$1" --batch "$2" \
           --steps 3 --warmup 1 --cpu-sample 0 > "$ROOT/gpurun_out/w.json" 2> "$ROOT/gpurun_out/w.err"; then
        echo "$lib m=$1: bench failed: $(tail -c 200 "$ROOT/gpurun_out/w.err")"; continue
      fi
      python3 - "$lib" "$1" "$ROOT/gpurun_out/w.json" <<'PY'
import json, sys
j = json.loads(open(sys.argv[3]).read().strip().splitlines()[-1])
print(sys.argv[1], "m<=" + sys.argv[2], round(j["value"]), "guides/s", round(j["ms_per_step"], 2), "ms/step, k_search",
      j["detail"]["k_search_ms_per_step"], "ms")
PY
    done
  done
  ;;
*) echo "usage: bash tools/sweep_pd.sh build | run"; exit 2 ;;
esac
