#!/bin/bash
# round-3 evidence: bench.py plain, under rocprofv3 --kernel-trace --stats, and the FETCH_SIZE / WRITE_SIZE passes
TAG=${1:-r03_v4}
R=$PWD
mkdir -p $R/gpurun_out
python bench.py 2>/dev/null | tail -1 > $R/gpurun_out/${TAG}_bench.json
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/prof_$TAG /tmp/pmc_fetch /tmp/pmc_write
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o p -- python3 $R/bench.py --no-cpu-baseline > /tmp/prof_$TAG.log 2>/dev/null
tail -1 /tmp/prof_$TAG.log > $R/gpurun_out/${TAG}_bench_under_rocprof.json
cp $(find /tmp/prof_$TAG -name "*kernel_stats.csv" | head -1) $R/gpurun_out/${TAG}_kernel_stats.csv
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pmc_fetch -- python3 $R/bench.py --steps 5 --warmup 2 --em-steps 3 --prewarm-ms 0 --data device --no-cpu-baseline --no-other-models > /tmp/pmc_f.log 2>&1 || tail -3 /tmp/pmc_f.log
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/pmc_write -- python3 $R/bench.py --steps 5 --warmup 2 --em-steps 3 --prewarm-ms 0 --data device --no-cpu-baseline --no-other-models > /tmp/pmc_w.log 2>&1 || tail -3 /tmp/pmc_w.log
python3 $R/profiles/summarize_pmc.py /tmp/pmc_fetch /tmp/pmc_write $R/gpurun_out/${TAG}_pmc_traffic.json > /dev/null
head -c 1200 $R/gpurun_out/${TAG}_bench.json; echo; head -8 $R/gpurun_out/${TAG}_kernel_stats.csv | cut -c1-170; grep -A8 "estep_fused" $R/gpurun_out/${TAG}_pmc_traffic.json | head -40
