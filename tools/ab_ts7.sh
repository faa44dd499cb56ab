#!/bin/bash
# the slab-wise 4-bit table kernel for up to 64 vectors (forced: M4RI_HIP_TS7=2) against the other paths (M4RI_HIP_TS7=0) on a grid (development tool)
S=""
for m in 64 1000 4096 20000 65536 1048576; do for l in 600 1000 2048 4096 20000 65536; do for n in 1 8 16 32 64; do
  if [ $((m * l / 8)) -le 2200000000 ]; then S="$S $m,$l,$n,auto,0,30"; fi; done; done; done
echo "== ts7 forced"; M4RI_HIP_TS7=2 python tools/shape_bench.py $S 2>&1 | grep -v amdgpu
echo "== ts7 off"; M4RI_HIP_TS7=0 python tools/shape_bench.py $S 2>&1 | grep -v amdgpu
