#!/bin/bash
# What the per-K-step barrier of the fused E-step kernel costs, in phase and with the second resident workgroup set
# started half a tile later (timing only: the no-barrier build computes garbage).
cd "$(dirname "$0")/.."
for flags in "" "-DPM_FUSED_NO_BARRIER"; do
  touch prosper_amd/csrc/bsc_fused.hip
  PM_EXTRA_FLAGS="$flags" bash prosper_amd/csrc/build.sh > /dev/null 2>&1
  for st in 0 60; do
    echo "flags='$flags' stagger_us=$st: $(PM_FUSED_STAGGER_US=$st python scratch/fused_probe.py 2>/dev/null | tail -1)"
  done
done
