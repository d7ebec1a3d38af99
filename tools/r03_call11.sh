#!/bin/bash
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/r03_call11
rm -rf $out && mkdir -p $out
cd $root
for i in 1 2 3; do timeout -k 10 300 python3 tools/debug_down.py 3 2>&1 | grep -E "bad px" ; done > $out/debug.txt; cat $out/debug.txt
if grep -q "bad px [1-9]" $out/debug.txt; then echo "STILL BAD"; exit 1; fi
bash tools/r03_call10.sh
