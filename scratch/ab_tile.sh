#!/bin/bash
# A/B of the fused E-step's tile (4 wavefronts per workgroup vs 8): correctness on the fused-path tests, then timing.
R=$PWD
OUT=$R/gpurun_out/ab_tile.log
: > $OUT
for tile in 8; do
  echo "== tests PM_FUSED_TILE=$tile" >> $OUT
  PM_FUSED_TILE=$tile timeout 900 python -m pytest tests/test_bsc_gpu.py -x -q -m gpu -k "fused or golden or config2 or oracle or speculation or spd_inverse_warm" 2>&1 | tail -15 >> $OUT
done
for rep in 1 2; do
for tile in 4 8; do
  echo "== bench PM_FUSED_TILE=$tile" >> $OUT
  PM_FUSED_TILE=$tile timeout 600 python scratch/bench_bsc_estep.py 2>&1 | tail -2 >> $OUT
done
done
cat $OUT
