#!/bin/bash
# Same-box A/B of mca_kernels.hip variants (scratch/mca_variant.sh): correctness of each (the MCA oracle tests, run with the
# variant copied over the shipped library in this scratch copy of the tree) and scratch/mca_em_time.py, alternating.
# usage (on the GPU box): scratch/mca_ab.sh <rounds> <name> [<name> ...]   (name "ship" = prosper_amd/libprosper_hip.so)
export PYTHONPATH=.
rounds=$1; shift
cp prosper_amd/libprosper_hip.so /tmp/ship.so
for n in "$@"; do
  [ "$n" = ship ] && continue
  cp scratch/ab_mca/lib_$n.so prosper_amd/libprosper_hip.so
  echo "== tests $n: $(python -m pytest tests/test_mca_gpu.py tests/test_mmca_gpu.py -x -q -m gpu 2>&1 | tail -1)"
done
cp /tmp/ship.so prosper_amd/libprosper_hip.so
for r in $(seq $rounds); do
  for n in "$@"; do
    if [ "$n" = ship ]; then lib=prosper_amd/libprosper_hip.so; else lib=scratch/ab_mca/lib_$n.so; fi
    echo "$n: $(PM_LIB_PATH=$lib python scratch/mca_em_time.py 2>&1 | tail -1)"
  done
done
