#!/bin/bash
# Collects the round's measurement evidence on the GPU box into gpurun_out/r03f/ (copied into profiles/r03_* afterwards):
#   tools/collect_profiles.sh        (run from the repo root through gpurun)
# rocprofv3 runs from /tmp with TMPDIR=/tmp, the program directly after `--`; counter passes use --kernel-trace only.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03f
mkdir -p $O
cd $R
python3 bench.py --steps 20 --warmup 5 > $O/final_bench.json 2> $O/final_bench.err
echo "bench done"
python3 tools/lpn_bench.py --json > $O/lpn_cold_warm.jsonl 2>/dev/null
tools/bench_shapes > $O/bench_shapes.jsonl 2>/dev/null
python3 tools/host_path_bench.py > $O/host_path.txt 2>/dev/null
python3 tools/pcie_duplex.py >> $O/host_path.txt 2>/dev/null
tools/sweep_sizes.sh > $O/size_sweep.txt 2>/dev/null
python3 tools/shape_bench.py 131072,131072,131072,auto,0,2 >> $O/size_sweep.txt 2>/dev/null
python3 tools/levels_sweep.py 4096x4096x4096 6144x6144x6144 8192x8192x8192 10240x10240x10240 12288x12288x12288 14336x14336x14336 16384x16384x16384 18432x18432x18432 20480x20480x20480 \
   22528x22528x22528 24576x24576x24576 28672x28672x28672 32768x32768x32768 36864x36864x36864 40960x40960x40960 45056x45056x45056 49152x49152x49152 57344x57344x57344 65536x65536x65536 \
   8192x65536x65536 16384x65536x65536 16384x65536x16384 8192x65536x16384 > $O/levels_sweep.txt 2>/dev/null
python3 tools/stream_bench.py > $O/stream_reference.txt 2>/dev/null
python3 tools/hbm_rates.py > $O/hbm_rates.txt 2>/dev/null
python3 tools/elim_bench.py 2>/dev/null | grep "^n=" > $O/elim.txt
# at most 64 vectors against long rows, few rows against a matrix (wave-per-row, slab table and v*A kernels)
python3 tools/shape_bench.py 65536,65536,1,naive,0,50 65536,65536,1,auto,0,50 65536,65536,8,auto,0,50 65536,65536,16,auto,0,30 65536,65536,32,auto,0,30 65536,65536,64,auto,0,30 \
   1048576,4096,64,auto,0,30 1048576,1000,8,auto,0,50 4096,65536,64,auto,0,100 20000,20000,64,auto,0,100 65536,4096,16,auto,0,100 1000,1000000,1,naive,0,50 \
   1,65536,65536,auto,0,50 8,65536,65536,auto,0,50 8,65536,256,auto,0,200 8,64,64,auto,0,500 65536,65536,128,auto,0,20 65536,65536,256,auto,0,20 20000,20000,128,auto,0,50 \
   65536,8192,512,auto,0,20 16,200000,600,auto,0,100 64,65536,4096,auto,0,50 2>/dev/null | grep -v amdgpu > $O/vector_paths.txt
python3 tools/transpose_bench.py 2>/dev/null | grep -v amdgpu > $O/transpose.txt
echo "timings done"
( cd tools && { echo "== (the first kernel a process times runs 4-5 % slow: every list below starts with a throw-away entry)";
  echo "== 343 leaves of 4096^3, packed A: legacy v7 (90), v8 with 4096 / 2048 / 1024 / 512-row tiles (9-12)"; APACK=1 ./kbench 4096 343 3 9 90 9 10 11 12;
  echo "== the same, unpacked A (and v6 = 8)"; ./kbench 4096 343 3 9 9 10 11 12 8;
  echo "== 2401 leaves, packed"; APACK=1 ./kbench 4096 2401 2 9 90 9;
  echo "== 4096^3 alone, every tile cut into stream-K segments, unpacked"; NREM=-1 ./kbench 4096 1 200 9 10 11 12;
  echo "== 4096^3 alone, uniform split-K of the older kernels"; KSPLIT=32 ./kbench 4096 1 200 8; KSPLIT=8 ./kbench 4096 1 200 7;
  echo "== 49 leaves of 4096^3 packed: whole tiles, then stream-K on the last round"; APACK=1 ./kbench 4096 49 5 9 10; APACK=1 NREM=-2 ./kbench 4096 49 5 9 10;
  echo "== read window of the lookup loop on 343 packed leaves: two steps (9, 10, 11), three (13, 14, 15), one (17, 18, 19)"; APACK=1 ./kbench 4096 343 3 9 9 13 17 10 14 18 11 15 19;
  echo "== static wave priority (GF2K_V8_FLAGS=1: waves 4-7, 2: waves 0-3)"; for f in 0 1 2; do GF2K_V8_FLAGS=$f APACK=1 ./kbench 4096 343 3 9; done;
  echo "== tall narrow tiles (v9 experiment: 21 / 22 / 23 = 4096 / 2048 / 1024 x 128) against v8 (9): 343 leaves packed, unpacked; 4096^3 and 8192^3 alone";
  APACK=1 ./kbench 4096 343 3 9 9 21 22 23; ./kbench 4096 343 3 9 9 21 22; NREM=-1 ./kbench 4096 1 200 11 21 22 23; APACK=1 NREM=-1 ./kbench 4096 1 200 11 21; NREM=-1 ./kbench 8192 1 50 10 21 22; } ) > $O/tile_variants.txt 2>&1
echo "kbench done"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu --no-configs > $O/prof_bench.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu --no-configs --no-parity > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu --no-configs --no-parity > $O/pmc_write.log 2>&1
echo "bench profiles done"
# BASELINE config 2 (4096^3, M4RM only) by itself: kernel summary and HBM counters
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c2 -- python3 $R/tools/shape_bench.py 4096,4096,4096,m4rm,0,200 > $O/prof_c2.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_c2_fetch -- python3 $R/tools/shape_bench.py 4096,4096,4096,m4rm,0,50 > $O/pmc_c2_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_c2_write -- python3 $R/tools/shape_bench.py 4096,4096,4096,m4rm,0,50 > $O/pmc_c2_write.log 2>&1
# the other configurations of the bench line (config 3 and the LPN products), kernel summary
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_configs -- python3 $R/tools/configs_bench.py > $O/prof_configs.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_lpn -- python3 $R/tools/lpn_pmc.py > $O/prof_lpn.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_lpn_fetch -- python3 $R/tools/lpn_pmc.py > $O/pmc_lpn_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_lpn_write -- python3 $R/tools/lpn_pmc.py > $O/pmc_lpn_write.log 2>&1
echo "config profiles done"
# SQ counters of the shipped tile kernel (v8, 4096-row tiles, packed A): 32 products of 8192^3, four counters per pass
i=0
for ctrs in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU" \
            "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INSTS_VMEM_RD"; do
  i=$((i+1))
  APACK=1 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $O/pmc_sq$i -- $R/tools/kbench 8192 32 1 9 > $O/pmc_sq$i.log 2>&1
done
echo "sq counters done"
cd $R
cp $(ls $O/prof_bench/*/*kernel_stats.csv | head -1) $O/final_kernel_stats.csv
cp $(ls $O/prof_c2/*/*kernel_stats.csv | head -1) $O/config2_kernel_stats.csv
cp $(ls $O/prof_configs/*/*kernel_stats.csv | head -1) $O/configs_kernel_stats.csv
cp $(ls $O/prof_lpn/*/*kernel_stats.csv | head -1) $O/lpn_kernel_stats.csv
python3 tools/pmc_summary.py FETCH_SIZE=$(ls $O/pmc_fetch/*/*counter_collection.csv | head -1) WRITE_SIZE=$(ls $O/pmc_write/*/*counter_collection.csv | head -1) > $O/final_pmc_summary.json
python3 tools/pmc_summary.py FETCH_SIZE=$(ls $O/pmc_c2_fetch/*/*counter_collection.csv | head -1) WRITE_SIZE=$(ls $O/pmc_c2_write/*/*counter_collection.csv | head -1) > $O/config2_pmc_summary.json
python3 tools/pmc_summary.py FETCH_SIZE=$(ls $O/pmc_lpn_fetch/*/*counter_collection.csv | head -1) WRITE_SIZE=$(ls $O/pmc_lpn_write/*/*counter_collection.csv | head -1) > $O/lpn_pmc_summary.json
python3 tools/sq_summary.py $O/pmc_sq1 $O/pmc_sq2 $O/pmc_sq3 > $O/v8_sq_counters.json
rm -rf $O/prof_bench $O/pmc_fetch $O/pmc_write $O/prof_c2 $O/pmc_c2_fetch $O/pmc_c2_write $O/prof_configs $O/prof_lpn $O/pmc_lpn_fetch $O/pmc_lpn_write $O/pmc_sq1 $O/pmc_sq2 $O/pmc_sq3
python3 bench.py --gpus 2 --backend gloo --check --dim 16384 --no-cpu --steps 3 --warmup 1 > $O/two_rank_rehearsal.json 2> $O/two_rank_rehearsal.err
ls -la $O
