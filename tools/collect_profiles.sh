#!/bin/bash
# Collects the round's measurement evidence on the GPU box into gpurun_out/r02/ (copied into profiles/ afterwards):
#   tools/collect_profiles.sh        (run from the repo root through gpurun)
# rocprofv3 runs from /tmp with TMPDIR=/tmp; counter passes use --kernel-trace only (no other trace domain).
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02
mkdir -p $O
cd $R
python3 bench.py --steps 20 --warmup 5 > $O/final_bench.json 2> $O/final_bench.err
echo "bench done" 
python3 tools/configs_bench.py > $O/configs_single_gpu.jsonl 2>/dev/null
python3 tools/lpn_bench.py --json >> $O/configs_single_gpu.jsonl 2>/dev/null
echo "configs done"
tools/bench_shapes > $O/bench_shapes.jsonl 2>/dev/null
python3 tools/host_path_bench.py > $O/host_path.txt 2>/dev/null
python3 tools/shape_bench.py 65536,65536,65536 65536,65536,65536,m4rm 32768,32768,32768 8192,65536,65536 8192,65536,16384 16384,65536,65536 32768,65536,65536 \
   60000,60000,60000 60000,60000,60000,m4rm 65600,65600,65600 65600,65600,65600,m4rm 131072,131072,131072,auto,0,2 > $O/shapes.txt 2>/dev/null
python3 tools/stream_bench.py > $O/stream_reference.txt 2>/dev/null
echo "shapes done"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu > $O/prof_bench.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu > $O/pmc_write.log 2>&1
echo "bench profiles done"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_lpn -- python3 $R/tools/lpn_pmc.py > $O/prof_lpn.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_lpn_fetch -- python3 $R/tools/lpn_pmc.py > $O/pmc_lpn_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_lpn_write -- python3 $R/tools/lpn_pmc.py > $O/pmc_lpn_write.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_configs -- python3 $R/tools/configs_bench.py > $O/prof_configs.log 2>&1
echo "lpn profiles done"
cd $R
cp $(ls $O/prof_bench/*/*kernel_stats.csv | head -1) $O/final_kernel_stats.csv
cp $(ls $O/prof_lpn/*/*kernel_stats.csv | head -1) $O/lpn_kernel_stats.csv
cp $(ls $O/prof_configs/*/*kernel_stats.csv | head -1) $O/configs_kernel_stats.csv
python3 tools/pmc_summary.py FETCH_SIZE=$(ls $O/pmc_fetch/*/*counter_collection.csv | head -1) WRITE_SIZE=$(ls $O/pmc_write/*/*counter_collection.csv | head -1) > $O/final_pmc_summary.json
python3 tools/pmc_summary.py FETCH_SIZE=$(ls $O/pmc_lpn_fetch/*/*counter_collection.csv | head -1) WRITE_SIZE=$(ls $O/pmc_lpn_write/*/*counter_collection.csv | head -1) > $O/lpn_pmc_summary.json
rm -rf $O/prof_bench $O/pmc_fetch $O/pmc_write $O/prof_lpn $O/pmc_lpn_fetch $O/pmc_lpn_write $O/prof_configs
python3 bench.py --gpus 2 --backend gloo --check --dim 16384 --no-cpu --steps 3 --warmup 1 > $O/two_rank_rehearsal.json 2> $O/two_rank_rehearsal.err
ls -la $O
