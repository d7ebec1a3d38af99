#!/bin/bash
# round 3, call 7: full GPU test suite, smoke, warp 2 x N variants, the bench line, 2-rank gloo rehearsal of the bench on one GPU
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/r03_call7
rm -rf $out && mkdir -p $out
cd $root
export NUS_EXPECT_GPU=1
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > $out/gpu_tests.txt 2>&1; rc=$?; echo "gpu tests rc=$rc"; tail -6 $out/gpu_tests.txt
[ $rc -eq 0 ] || exit 1
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.txt 2>&1; echo "smoke rc=$?"; tail -3 $out/smoke.txt
for v in product wx2r4 wx2r1 wx2r3 product; do
  echo "== warp kernel: $v"
  if [ $v = product ]; then unset NUS_LIB_PATH; else export NUS_LIB_PATH=$root/tools/_ablate/lib_$v.so; fi
  timeout -k 10 200 python3 tools/warp_bench.py 2>&1 | grep warp_blend_flow
done > $out/warp_layouts_2xN.txt 2>&1; unset NUS_LIB_PATH; cat $out/warp_layouts_2xN.txt
timeout -k 10 600 python3 bench.py > $out/bench_n1.log 2>&1; echo "bench rc=$?"; tail -1 $out/bench_n1.log > $out/bench_n1.json; cut -c1-300 $out/bench_n1.json
timeout -k 10 400 python3 bench.py --gpus 2 --backend gloo --force-device 0 --steps 30 --warmup 3 --units 100 --sustained-seconds 0 > $out/bench_n2_gloo.log 2>&1; echo "rehearsal rc=$?"; grep "^{" $out/bench_n2_gloo.log | tail -1 > $out/bench_n2_gloo.json; cut -c1-400 $out/bench_n2_gloo.json
