#!/bin/bash
# A/B builds of k_search with one round-4 addition compiled out at a time: the per-class match
# counters (GS_X_NO_CLS), the 59-bit path (GS_X_PB=0u: the 52-bit layout) and the literal-N window buckets
# (GS_X_NO_BUCKETS).  Usage: bash tools/ab_variants.sh build   (here, where hipcc is)
#                            bash tools/ab_variants.sh run     (GPU box, repo root: the default bench with each library)
#                            bash tools/ab_variants.sh clean
set -euo pipefail
ROOT=$(cd "$(dirname "$0")/.." && pwd)
CSRC=$ROOT/guidescan-cli_amd/csrc
VARIANTS=("nocls:-DGS_X_NO_CLS" "pb0:-DGS_X_PB=0u" "nobk:-DGS_X_NO_BUCKETS" "all3:-DGS_X_NO_CLS -DGS_X_PB=0u -DGS_X_NO_BUCKETS")
case "${1:-run}" in
build)
  for v in "${VARIANTS[@]}"; do
    n=${v%%:*}; f=${v#*:}
    make -s -C "$CSRC" lib OUT="$ROOT/guidescan-cli_amd/libgsamd_x$n.so" BUILD=build_x$n EXTRA="$f"
  done
  ls -la "$ROOT"/guidescan-cli_amd/libgsamd*.so
  ;;
run)
  mkdir -p "$ROOT/gpurun_out"
  for lib in libgsamd.so libgsamd_xnocls.so libgsamd_xpb0.so libgsamd_xnobk.so libgsamd_xall3.so; do
    if [ ! -f "$ROOT/guidescan-cli_amd/$lib" ]; then echo "$lib not built (bash tools/ab_variants.sh build): skipped"; continue; fi
    if ! GS_LIB_PATH=$ROOT/guidescan-cli_amd/$lib python "$ROOT/bench.py" --cpu-sample 0 --steps 5 --warmup 1 \
         > "$ROOT/gpurun_out/ab.json" 2> "$ROOT/gpurun_out/ab.err"; then
      echo "$lib: bench failed: $(tail -c 200 "$ROOT/gpurun_out/ab.err")"; continue
    fi
    python3 - "$lib" "$ROOT/gpurun_out/ab.json" <<'PY'
import json, sys
j = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
print(sys.argv[1], round(j["value"]), "guides/s", round(j["ms_per_step"], 2), "ms/step, k_search", j["detail"]["k_search_ms_per_step"], "ms")
PY
  done
  ;;
clean)
  rm -f "$ROOT"/guidescan-cli_amd/libgsamd_x*.so
  rm -rf "$CSRC"/build_x*
  ;;
*) echo "usage: bash tools/ab_variants.sh build | run | clean"; exit 2 ;;
esac
