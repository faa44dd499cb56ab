#!/bin/bash
# the wave-per-row kernel against the older paths on a grid of (m, l, n) (development tool; run on the GPU box)
S=""
for m in 1000 65536; do for l in 600 1024 2048 4096 20000 65536; do for n in 1 8 16 32 64; do S="$S $m,$l,$n,auto,0,50"; done; done; done
for m in 1 8 15; do for n in 1 64; do S="$S $m,65536,$n,auto,0,50"; done; done
echo "== widevec 1"; M4RI_HIP_WIDEVEC=1 python tools/shape_bench.py $S 2>&1 | grep -v amdgpu
echo "== widevec 0"; M4RI_HIP_WIDEVEC=0 python tools/shape_bench.py $S 2>&1 | grep -v amdgpu
