#!/bin/bash
# HBM-side traffic per kernel: separate FETCH_SIZE / WRITE_SIZE passes (MI355X_MICROARCH.md), then summarize
R=$PWD
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/pmc_fetch /tmp/pmc_write
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pmc_fetch -- python3 $R/bench.py --steps 5 --warmup 2 --em-steps 3 --no-cpu-baseline --no-other-models > /tmp/pmc_f.log 2>&1 || tail -3 /tmp/pmc_f.log
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/pmc_write -- python3 $R/bench.py --steps 5 --warmup 2 --em-steps 3 --no-cpu-baseline --no-other-models > /tmp/pmc_w.log 2>&1 || tail -3 /tmp/pmc_w.log
mkdir -p $R/gpurun_out
python3 $R/profiles/summarize_pmc.py /tmp/pmc_fetch /tmp/pmc_write $R/gpurun_out/r01_pmc_traffic_v6.json | head -60
