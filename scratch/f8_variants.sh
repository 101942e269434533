#!/bin/bash
# Build diagnostic variants of the library (bsc_fused8.hip with extra -D flags) into scratch/libs/ and the harness.
# usage: scratch/f8_variants.sh name1:"-DFLAG ..." name2:"..."      (run here, on the CPU box; the .so files travel)
set -e
cd "$(dirname "$0")/.."
HIPCC=/opt/rocm/bin/hipcc
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-function -Iinclude -Iprosper_amd/csrc"
mkdir -p scratch/libs
bash prosper_amd/csrc/build.sh > /dev/null
others=$(ls prosper_amd/csrc/build/*.o | grep -v bsc_fused8.o)
for spec in "$@"; do
  name=${spec%%:*}; extra=${spec#*:}
  [ "$extra" = "$spec" ] && extra=""
  $HIPCC $FLAGS $extra -c prosper_amd/csrc/bsc_fused8.hip -o scratch/libs/f8_$name.o &
done
wait
for spec in "$@"; do
  name=${spec%%:*}
  $HIPCC --offload-arch=gfx950 -shared -fPIC -o scratch/libs/libpm_$name.so scratch/libs/f8_$name.o $others
done
$HIPCC --offload-arch=gfx950 -O2 -std=c++17 scratch/f8_bench.hip -o scratch/libs/f8_bench -ldl
ls scratch/libs/
