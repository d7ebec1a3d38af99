#!/bin/bash
# A/B of flow front-end variant libraries (tools/build_lz_variants.sh with UNIT=nus_k_flow) against the product, two interleaved
# rounds: tools/flow_variants_ab.sh name1 name2 ...   (run on the GPU box; dev tool)
cd ${GRAFT_REPO_ROOT:-/root/repo}
for r in 1 2; do
  for v in "$@"; do
    echo "== $v"; NUS_LIB_PATH=tools/_ablate/lib_$v.so timeout -k 10 200 python3 tools/flow_stream_bench.py 101 9 9 2>&1 | grep "flow stream" | tail -2
  done
  echo "== product"; timeout -k 10 200 python3 tools/flow_stream_bench.py 101 9 9 2>&1 | grep "flow stream" | tail -2
done
