#!/bin/bash
# bench.py plain, then under rocprofv3 --kernel-trace --stats; results into gpurun_out/
R=$PWD
mkdir -p $R/gpurun_out
python bench.py 2>/dev/null | tail -1 > $R/gpurun_out/r01_v7_bench.json
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_v4 -o v4 -- python3 $R/bench.py > /tmp/prof_v4.log 2>/dev/null
tail -1 /tmp/prof_v4.log > $R/gpurun_out/r01_v7_bench_under_rocprof.json
cp $(find /tmp/prof_v4 -name "*kernel_stats.csv" | head -1) $R/gpurun_out/r01_v7_kernel_stats.csv
head -c 600 $R/gpurun_out/r01_v7_bench.json; echo; head -5 $R/gpurun_out/r01_v7_kernel_stats.csv | cut -c1-160
