#!/bin/bash
# every path for at most 64 vectors by rule (default) against the slab table kernel switched off and against everything new switched off (development tool)
S=""
for m in 64 1000 4096 20000 65536 1048576; do for l in 600 1000 2048 4096 20000 65536; do for n in 1 8 16 32 64; do
  if [ $((m * l / 8)) -le 2200000000 ]; then S="$S $m,$l,$n,auto,0,30"; fi; done; done; done
echo "== default"; python tools/shape_bench.py $S 2>&1 | grep -v amdgpu
echo "== round-2 paths"; M4RI_HIP_TS7=0 M4RI_HIP_WIDEVEC=0 M4RI_HIP_TS_LONG_MIN_ROWS=2048 python tools/shape_bench.py $S 2>&1 | grep -v amdgpu
