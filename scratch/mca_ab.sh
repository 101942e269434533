#!/bin/bash
# A/B of compile-time variants of mca_kernels.hip on the GPU box: scratch/mca_ab.sh "" "-DPM_MCA_VSKIP" ...
cd "$(dirname "$0")/.."
cp prosper_amd/libprosper_hip.so /tmp/lib.keep
for f in "$@"; do
  touch prosper_amd/csrc/mca_kernels.hip
  PM_EXTRA_FLAGS="$f" bash prosper_amd/csrc/build.sh > /dev/null 2>&1
  echo "flags '$f': $(PYTHONPATH=. python ${MCA_SCRIPT:-scratch/mca_kernel_time.py} 2>/dev/null | tail -1 | tr '\n' ' ')"
done
cp /tmp/lib.keep prosper_amd/libprosper_hip.so
touch prosper_amd/csrc/mca_kernels.hip
