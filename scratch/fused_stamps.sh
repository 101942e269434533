#!/bin/bash
# diagnostic build of bsc_fused.hip with per-workgroup time stamps, run, then restore the shipped library
set -e
cd "$(dirname "$0")/.."
cp prosper_amd/libprosper_hip.so /tmp/libprosper_hip.keep
touch prosper_amd/csrc/bsc_fused.hip
PM_EXTRA_FLAGS=-DPM_FUSED_STAMPS bash prosper_amd/csrc/build.sh > /dev/null
python scratch/fused_stamps.py 2>&1 | grep -v amdgpu.ids

cp /tmp/libprosper_hip.keep prosper_amd/libprosper_hip.so
touch prosper_amd/csrc/bsc_fused.hip
