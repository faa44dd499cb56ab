S=""
for m in 4096 20000 65536 262144 1048576; do for l in 600 1000; do for n in 1 8 16 32 64 200 256; do S="$S $m,$l,$n,auto,0,50"; done; done; done
echo "== current"; python tools/shape_bench.py $S 2>&1 | grep -v amdgpu
echo "== no long ts, widevec from 513"; M4RI_HIP_TS_LONG_MIN_ROWS=2000000000 M4RI_HIP_WIDEVEC_MINL16=513 M4RI_HIP_WIDEVEC_MINL32=513 python tools/shape_bench.py $S 2>&1 | grep -v amdgpu
echo "== no long ts, no widevec"; M4RI_HIP_TS_LONG_MIN_ROWS=2000000000 M4RI_HIP_WIDEVEC=0 python tools/shape_bench.py $S 2>&1 | grep -v amdgpu
