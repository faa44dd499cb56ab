#!/bin/bash
# A/B of the row-band plans (development tool; run on the GPU box): the automatic plans of ragged sizes with and without bands
S="4096,4096,4096,m4rm,0,50"
for n in 10000 12700 14000 17000 20000 22000 26000 28000 30000 33000; do S="$S $n,$n,$n,auto,0,20"; done
for n in 36000 40000 45000 52000 57000 60000 63000 66000 70000; do S="$S $n,$n,$n,auto,0,5"; done
S="$S 8512,65536,65536,auto,0,5 17000,65536,65536,auto,0,5 8512,1024,32768,m4rm,0,50 17152,4096,8192,strassen,2,20"
for b in 1 0 1 0; do
  echo "== bands $b"; M4RI_HIP_ROW_BANDS=$b python tools/shape_bench.py $S 2>&1 | grep -v amdgpu
done
