#!/bin/bash
# A/B timing of Lanczos x2 variants on one box: tools/lz_ab.sh <label>=<lib or ""> ...   (env vars via label@VAR=VAL)
for spec in "$@"; do
  label="${spec%%=*}"; lib="${spec#*=}"
  for pat in gradient noise; do
    echo "== $label $pat"
    if [ -n "$lib" ]; then export NUS_LIB_PATH="$lib"; else unset NUS_LIB_PATH; fi
    timeout -k 10 300 python tools/quick_bench.py --frames 300 --reps 5 --pattern $pat --only lanczos3 2>&1 | grep "fma"
  done
done
