#!/bin/bash
# nearest / bilinear at the P/Q factors (864p / 900p -> 1080p, 1080p -> 1800p, 864p -> 4K): fixed-ratio kernels next to the table kernels
# usage: bash tools/ratio_bench.sh   (second line of each pair: NUS_GENERAL=1 = option force_general, the table kernels)
root=${GRAFT_REPO_ROOT:-/root/repo}
for d in "1536 864 1920 1080" "1600 900 1920 1080" "1920 1080 3200 1800" "1536 864 3840 2160"; do
  for pat in gradient noise; do
    NUS_PATTERN=$pat python3 $root/tools/general_bench.py $d 128 2>&1 | grep -E "nearest|bilinear" | sed "s/^/$pat /"
    NUS_PATTERN=$pat NUS_GENERAL=1 python3 $root/tools/general_bench.py $d 128 2>&1 | grep -E "nearest|bilinear" | sed "s/^/$pat /"
  done
done
