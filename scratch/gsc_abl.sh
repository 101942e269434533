#!/bin/bash
# timing-only ablations of gsc_estep_kernel (PM_GSC_ABL bits: 1 no multi-cause loop, 2 no singleton exponentials, 4 no xs/xsz
# stores, 8 no LDS column-sum atomics, 16 no pair-block global atomics, 32 one selection round, 64 no list emission); built here,
# run on the box:  bash scratch/gsc_abl.sh build | run
cd "$(dirname "$0")/.."
if [ "$1" = build ]; then
  mkdir -p scratch/libs
  FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Iinclude -Iprosper_amd/csrc"
  for a in ${2:-1 2 4 8 16 32 64 128}; do
    /opt/rocm/bin/hipcc $FL -DPM_GSC_ABL=$a -c prosper_amd/csrc/gsc_kernels.hip -o /tmp/gsc_abl$a.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o scratch/libs/libgabl$a.so $(ls prosper_amd/csrc/build/*.o | grep -v gsc_kernels.o) /tmp/gsc_abl$a.o &
  done
  wait
else
  python scratch/gsc_estep_time.py save
  for l in prosper_amd/libprosper_hip.so scratch/libs/libgabl*.so; do
    echo "$l $(PM_LIB_PATH=$l python scratch/gsc_estep_time.py 2>&1 | tail -1)"
  done
fi
