#!/bin/bash
# Build scratch/ab_mca/lib_<name>.so: the shipped objects with ONE source file recompiled with extra flags.
# usage: scratch/file_variant.sh <name> <file.hip (in prosper_amd/csrc)> [extra hipcc flags...]
set -euo pipefail
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
name="$1"; f="$2"; shift 2
mkdir -p "$ROOT/scratch/ab_mca" /tmp/file_variant
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -I"$ROOT/include" -I"$ROOT/prosper_amd/csrc" "$@" -c "$ROOT/prosper_amd/csrc/$f" -o /tmp/file_variant/$name.o
objs=$(ls "$ROOT"/prosper_amd/csrc/build/*.o | grep -v "/${f%.hip}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$ROOT/scratch/ab_mca/lib_$name.so" $objs /tmp/file_variant/$name.o
python "$ROOT/scratch/kmeta.py" /tmp/file_variant/$name.o 2>&1 | awk '{n=$1; sub(/^_ZN12_GLOBAL__N_1[0-9]*/,"",n); print substr(n,1,60), $0}' | sed -E "s/ _Z[^ ]* / /" | grep -v "spill_count': 0, '.vgpr_count': [0-9]*, '.vgpr_spill_count': 0" | head -5 || true
