#!/bin/bash
# Timing ablations of the fused MCA E-step + M-statistics kernel (run on the GPU box): rebuilds the library with
# -DPM_MCA_ABL=n and times scratch/bench_mca.py.  The ablated builds compute wrong results by design.
cd "$(dirname "$0")/.."
for a in ${ABLS:-0 1 2 3 4 5 6 7}; do
  touch prosper_amd/csrc/mca_kernels.hip
  PM_EXTRA_FLAGS=-DPM_MCA_ABL=$a bash prosper_amd/csrc/build.sh > /dev/null 2>&1
  echo "ABL $a: $(python scratch/mca_kernel_time.py 2>/dev/null | tail -1 | tr '\n' ' ')"
done
