#!/bin/bash
# A/B two builds of the library in one job, interleaved: scratch/ab_libs.sh lib_A.so lib_C.so [rounds]
cd "$(dirname "$0")/.."
cp prosper_amd/libprosper_hip.so /tmp/keep.so
for r in $(seq 1 ${3:-3}); do
  for l in $1 $2; do
    cp scratch/$l prosper_amd/libprosper_hip.so
    echo -n "$l N=200000: "; python scratch/fused_probe.py 2>&1 | grep "pass ms" | tail -1
    echo -n "$l N=196608: "; N=196608 python scratch/fused_probe.py 2>&1 | grep "pass ms" | tail -1
  done
done
cp /tmp/keep.so prosper_amd/libprosper_hip.so
