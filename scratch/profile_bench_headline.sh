#!/bin/bash
# bench.py --no-other-models under rocprofv3 --kernel-trace --stats: every gemm_nt_f64_dma_kernel<false> call is a
# main scores launch of the headline workload, so its average is directly comparable with roofline.avg_launch_ms
R=$PWD
mkdir -p $R/gpurun_out
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_v5b -o v5b -- python3 $R/bench.py --no-other-models > /tmp/prof_v5b.log 2>/dev/null
tail -1 /tmp/prof_v5b.log > $R/gpurun_out/r01_v7_headline_bench_under_rocprof.json
cp $(find /tmp/prof_v5b -name "*kernel_stats.csv" | head -1) $R/gpurun_out/r01_v7_headline_kernel_stats.csv
head -4 $R/gpurun_out/r01_v7_headline_kernel_stats.csv | cut -c1-200
