#!/bin/bash
# tools/seed_forms.py (form 2; form 0 once, with the first library, as the box's yardstick) with each of the libraries given
# (paths relative to the repo root), on one box.  Usage: bash tools/ab_seed_libs.sh <workload> <batch> <m> lib1.so lib2.so ...
WL=$1; BATCH=$2; M=$3; shift 3
first=1
for lib in "$@"; do
  forms=2; [ $first = 1 ] && forms=0,2; first=0
  echo "== $lib"
  GS_LIB_PATH=$PWD/$lib timeout -k 10 300 python tools/seed_forms.py $WL $BATCH $M $forms 2>&1 | grep "^form"
done
