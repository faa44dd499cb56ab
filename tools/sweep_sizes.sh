#!/bin/bash
# auto against m4rm over mid sizes, the per-rank shapes of the multi-GPU bench and the small squares (development tool; run on the GPU box)
S="4096,4096,4096,m4rm,0,200 4096,4096,4096,auto,0,200 2048,2048,2048,m4rm,0,200 8192,8192,8192,m4rm,0,50 8192,8192,8192,auto,0,50"
# (where auto keeps zero levels both lines run the very same launches: 20 repetitions at the small sizes keep the scatter of the pair under 1 %)
for n in 10000 12288 14000 16384 17000 20000 24576 28000 30000; do S="$S $n,$n,$n,auto,0,20 $n,$n,$n,m4rm,0,20"; done
for n in 33000 36000 40000 45000 49152 52000 57000 61440 63000 66000 70000; do S="$S $n,$n,$n,auto,0,3 $n,$n,$n,m4rm,0,3"; done
python tools/shape_bench.py $S 2>&1 | grep -v amdgpu
python tools/shape_bench.py 65536,65536,65536 65536,65536,65536,m4rm 32768,32768,32768 8192,65536,65536 8192,65536,16384 16384,65536,65536 16384,65536,16384 32768,65536,65536 \
   60000,60000,60000 60000,60000,60000,m4rm 65600,65600,65600 2>&1 | grep -v amdgpu
