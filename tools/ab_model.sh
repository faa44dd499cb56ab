#!/bin/bash
# A/B of the planner's defaults ("new") against older constants given as OLD="VAR=value ..." (development tool; run on the GPU box), e.g.
#   OLD="M4RI_HIP_V8_QUAD_NS=712,974,1498,2546 M4RI_HIP_SHAPE_MIN_GAIN_PCT=8"   the tile model before the per-tile overhead was taken out
#   OLD="M4RI_HIP_V8_UNPACKED_LONG_PCT=100"                                     unpacked A priced the same whatever the row length
#   OLD="M4RI_HIP_OLDER_LONG_PCT=100 M4RI_HIP_OLDER_KSPLIT_PPM=0"               the v3 / v6 model without its long-row and per-slice surcharges
OLD=${OLD:-M4RI_HIP_V8_UNPACKED_LONG_PCT=100}
S="4096,4096,4096,m4rm,0,200"
for n in 6144 8192 10000 10240 12288 14000 14336 16384 17000 18432 20000 20480 22528 24576 26000 28000 28672 30000; do S="$S $n,$n,$n,auto,0,20"; done
for n in 33000 36000 36864 40000 45000 45056 49152 52000 57000 57344 60000 61440 63000 66000 70000; do S="$S $n,$n,$n,auto,0,5"; done
S="$S 8192,65536,65536,auto,0,5 16384,65536,65536,auto,0,5 32768,65536,65536,auto,0,5 32768,32768,32768,auto,0,10 65536,65536,65536,auto,0,5"
S="$S 2048,33000,600,auto,0,100 9000,33000,300,auto,0,100 33000,9000,300,auto,0,100 300,33000,2048,auto,0,100 9000,9000,300,auto,0,100 20000,20000,400,auto,0,50 2048,9000,600,auto,0,100 33000,2048,600,auto,0,100 300,9000,9000,auto,0,100 128,9000,33000,auto,0,100 1000,20000,1000,auto,0,100 4096,20000,1000,auto,0,100"
S="$S 65536,65536,200,auto,0,10 65536,20000,256,auto,0,20 65536,8192,512,auto,0,20 65536,4096,512,auto,0,50 16384,65536,512,auto,0,20 65536,65536,1024,auto,0,10 65536,16384,200,auto,0,20 4096,65536,1000,auto,0,20 20000,20000,1000,auto,0,20 65536,6000,300,auto,0,20 30000,30000,700,auto,0,20 8192,12000,2048,auto,0,20"
for b in new old new old; do
  echo "== $b"
  if [ $b = new ]; then python tools/shape_bench.py $S 2>&1 | grep -v amdgpu
  else env $OLD python tools/shape_bench.py $S 2>&1 | grep -v amdgpu; fi
done
