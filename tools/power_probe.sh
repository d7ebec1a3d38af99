#!/bin/bash
# Sample rocm-smi power / clocks of every GPU of the host while one kernel runs in a loop (read-only queries).
alg=${1:-lanczos3}; pat=${2:-gradient}; reps=${3:-3000}
echo "HIP_VISIBLE_DEVICES=$HIP_VISIBLE_DEVICES ROCR_VISIBLE_DEVICES=$ROCR_VISIBLE_DEVICES"
python tools/quick_bench.py --frames 300 --reps $reps --pattern $pat --only $alg > gpurun_out/power_run_$alg.txt 2>&1 &
pid=$!
sleep 9
for i in 1 2 3; do
  rocm-smi --showpower --showclocks 2>&1 | grep -E "Socket Graphics Package Power|sclk" | sed 's/clock level: [0-9S]*: //;s/Current Socket Graphics Package Power (W)/W/' | tr '\n' ' '; echo
  sleep 1
done
kill $pid 2>/dev/null; wait $pid 2>/dev/null
