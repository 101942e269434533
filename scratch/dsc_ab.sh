#!/bin/bash
# A/B of compile-time variants of dsc_kernels.hip on the GPU box: scratch/dsc_ab.sh "" "-DPM_DSC_M16_WPE=4" ...
cd "$(dirname "$0")/.."
cp prosper_amd/libprosper_hip.so /tmp/lib.keep
for f in "$@"; do
  touch prosper_amd/csrc/dsc_kernels.hip
  PM_EXTRA_FLAGS="$f" bash prosper_amd/csrc/build.sh > /dev/null 2>&1
  echo "flags '$f': $(python scratch/bench_dsc.py 2>/dev/null | tail -3 | head -2 | tr '\n' ' ')"
done
cp /tmp/lib.keep prosper_amd/libprosper_hip.so
touch prosper_amd/csrc/dsc_kernels.hip
