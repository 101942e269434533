#!/bin/bash
# standalone builds of bsc_wp_sparse.hip: scratch/sparse_variants.sh name:"-DFLAGS" ...  -> gpurun_out/sp_<name>.so
set -e
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for v in "$@"; do
  name="${v%%:*}"; flags="${v#*:}"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -munsafe-fp-atomics -Iinclude -Iprosper_amd/csrc $flags \
     prosper_amd/csrc/bsc_wp_sparse.hip -o scratch/sp_$name.so &
done
wait
ls scratch/sp_*.so
