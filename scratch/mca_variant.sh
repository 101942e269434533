#!/bin/bash
# Build scratch/ab_mca/lib_<name>.so: the shipped objects with mca_kernels.hip compiled from <source> with <extra flags>.
# usage: scratch/mca_variant.sh <name> <source .hip> [extra hipcc flags...]
set -euo pipefail
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
name="$1"; src="$2"; shift 2
mkdir -p "$ROOT/scratch/ab_mca" /tmp/mca_variant
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -I"$ROOT/include" -I"$ROOT/prosper_amd/csrc" "$@" -c "$src" -o /tmp/mca_variant/$name.o
objs=$(ls "$ROOT"/prosper_amd/csrc/build/*.o | grep -v mca_kernels.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$ROOT/scratch/ab_mca/lib_$name.so" $objs /tmp/mca_variant/$name.o
python "$ROOT/scratch/kmeta.py" /tmp/mca_variant/$name.o estep_fused 2>&1 | grep "ILi4ELi8ELb0ELi21" | cut -c150-260
