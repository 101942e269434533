#!/bin/bash
cd "$(dirname "$0")/.."
OUT=gpurun_out/f8_test.log
PM_FUSED_TILE=8 timeout 900 python -m pytest tests/test_bsc_gpu.py -x -q -m gpu -k "fused or golden or config2 or oracle or speculati" 2>&1 | tail -15 > $OUT
cat $OUT
