#!/bin/bash
# Runs tools/probe_pageable_d2h on the GPU box: timings first, then once more under AMD_LOG_LEVEL=4 to see which road the
# runtime says it takes for one 4K frame (33 177 600 B) and for 512 KiB.  Output: gpurun_out/probe_pageable_d2h.txt
set -e
# the runtime the harness processes use is the one torch bundles (it is loaded first there): probe that one
export LD_LIBRARY_PATH=/usr/local/lib/python3.10/dist-packages/torch/lib:$LD_LIBRARY_PATH
mkdir -p gpurun_out
out=gpurun_out/probe_pageable_d2h.txt
{
  echo "== timings (33 177 600 B)"; ./tools/probe_pageable_d2h 33177600 24
  echo "== timings (4 MiB)"; ./tools/probe_pageable_d2h 4194304 24
  echo "== AMD_LOG_LEVEL=4, one 4K frame into pageable memory: what the runtime logs between the call and its return (2nd copy of the pageable case)"
  AMD_LOG_LEVEL=4 ./tools/probe_pageable_d2h 33177600 2 > /dev/null 2> gpurun_out/probe_amdlog.txt || true
  grep -n "hipMemcpy" gpurun_out/probe_amdlog.txt | head -20
  # the 4th hipMemcpy of the run is the second copy into the same pageable block (after two into the pinned block and one into this one)
  awk '/hipMemcpy \(/{n++} n==4' gpurun_out/probe_amdlog.txt | cut -c1-230 | head -60
  echo "== the same for the 3rd hipMemcpy (first copy into that pageable block)"
  awk '/hipMemcpy \(/{n++} n==3' gpurun_out/probe_amdlog.txt | cut -c1-230 | head -60
  rm -f gpurun_out/probe_amdlog.txt
} > $out 2>&1
cat $out
