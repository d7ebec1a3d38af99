#!/bin/bash
# Round 6: would a Jacobi kernel that forms level 0's luminance itself pay?  Timing-only A/B libraries (tools/_ablate/lib_*.so, built with
#   UNIT=nus_k_flow tools/build_lz_variants.sh nolum0="-DNUS_ABLATE_PYR_NO_LUM0_STORE" lumcost="-DNUS_ABLATE_HS_LUM0_COST" both="-D... -D..."):
#   nolum0   the level-0 pyramid pass without its luminance-plane store (the most such a kernel could SAVE)
#   lumcost  the level-0 Jacobi launches with the instructions the in-kernel luminance costs at least (what it would ADD)
#   both     the two together = the best case of the fused form
# against the product, interleaved, on the flow stream (us per 1080p pair) and on the motion step (ms per 300 units).
cd ${GRAFT_REPO_ROOT:-/root/repo}
out=gpurun_out/lum0_ablation.txt
{
for r in 1 2; do
  for v in product nolum0 lumcost both; do
    lib=""; [ $v != product ] && lib=tools/_ablate/lib_$v.so
    echo "== $v (round $r)"
    NUS_LIB_PATH=$lib timeout -k 10 200 python3 tools/flow_stream_bench.py 101 9 2>&1 | grep "flow stream" | tail -1
    NUS_LIB_PATH=$lib timeout -k 10 200 python3 tools/motion_default.py 2>&1 | grep "motion step"
  done
done
} > $out 2>&1
cat $out
