#!/bin/bash
cd "$(dirname "$0")/.."
OUT=gpurun_out/f8_test.log
PM_FUSED_TILE=${1:-8} timeout 1200 python -m pytest tests/test_bsc_gpu.py -x -q -m gpu -k "not gemm and not spd and not moments" 2>&1 | tail -15 > $OUT
cat $OUT
