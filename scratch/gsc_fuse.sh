#!/bin/bash
# fused-scores variants of gsc_estep_kernel against the GEMM launch in front (FUSE=0).  Variants are libraries under scratch/libs/
# built here with extra -D flags:  bash scratch/gsc_fuse.sh build name "-DFLAG ..." ; on the box: bash scratch/gsc_fuse.sh run
cd "$(dirname "$0")/.."
if [ "$1" = build ]; then
  mkdir -p scratch/libs
  FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Iinclude -Iprosper_amd/csrc"
  /opt/rocm/bin/hipcc $FL $3 -c prosper_amd/csrc/gsc_kernels.hip -o /tmp/gsc_$2.o &&
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o scratch/libs/libg$2.so $(ls prosper_amd/csrc/build/*.o | grep -v gsc_kernels.o) /tmp/gsc_$2.o
else
  python scratch/gsc_estep_time.py save
  echo "unfused: $(FUSE=0 python scratch/gsc_estep_time.py 2>&1 | tail -1)"
  for l in prosper_amd/libprosper_hip.so scratch/libs/libg*.so; do
    echo "$l $(PM_LIB_PATH=$l python scratch/gsc_estep_time.py 2>&1 | tail -1)"
  done
  echo "em unfused: $(FUSE=0 python scratch/gsc_em_time.py 2>&1 | tail -1)"
  for l in prosper_amd/libprosper_hip.so scratch/libs/libg*.so; do
    echo "em $l $(PM_LIB_PATH=$l python scratch/gsc_em_time.py 2>&1 | tail -1)"
  done
fi
