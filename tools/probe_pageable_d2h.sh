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
  echo "== AMD_LOG_LEVEL=4, one 4K frame: lines that name the copy's road"
  AMD_LOG_LEVEL=4 ./tools/probe_pageable_d2h 33177600 3 2>&1 | grep -i "pinned resource\|staging resource\|Unpinned\|staging D2H\|pin" | sort | uniq -c | sort -rn | head -12 || true
  echo "== AMD_LOG_LEVEL=4, 512 KiB"
  AMD_LOG_LEVEL=4 ./tools/probe_pageable_d2h 524288 3 2>&1 | grep -i "pinned resource\|staging resource\|Unpinned\|staging D2H\|pin" | sort | uniq -c | sort -rn | head -12 || true
} > $out 2>&1
cat $out
