#!/bin/bash
# Build libprosper_hip.so for gfx950 in-tree (cross-compiles without a GPU), and libprosper_hip_det.so: the same sources with
# -DPM_DETERMINISTIC (order-independent reductions, pm_common.h; `model.deterministic = True` loads it).
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
ROOT="$(cd "$HERE/../.." && pwd)"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wall -Wno-unused-function -Wno-unused-variable -I$ROOT/include -I$HERE"
build_one() {   # <object dir> <output .so> <extra flags>
  local dir="$1" out="$2" extra="$3"
  mkdir -p "$dir"
  local objs=()
  for src in "$HERE"/*.hip; do
    local obj="$dir/$(basename "${src%.hip}").o"
    if [ ! -f "$obj" ] || [ "$src" -nt "$obj" ] || [ "$HERE/pm_common.h" -nt "$obj" ] || [ "$HERE/bsc_rows16_body.h" -nt "$obj" ] || [ "$ROOT/include/prosper_hip.h" -nt "$obj" ]; then
      local per_file=""
      # the MCA state loop is two interleaved dependent chains at two wavefronts per SIMD: the ILP-first scheduler
      # (239 instead of 225 registers, both under 256) is worth 2-3 % of the pass
      [ "$(basename "$src")" = mca_kernels.hip ] && per_file="-mllvm -amdgpu-sched-strategy=max-ilp"
      echo "hipcc $(basename "$src") $extra $per_file"
      "$HIPCC" $FLAGS $extra $per_file ${PM_EXTRA_FLAGS:-} -c "$src" -o "$obj" &
    fi
    objs+=("$obj")
  done
  wait
  "$HIPCC" --offload-arch=gfx950 -shared -fPIC -o "$out" "${objs[@]}"
  echo "built $out"
}
build_one "$HERE/build" "$HERE/../libprosper_hip.so" ""
if [ "${PM_SKIP_DET:-0}" != "1" ]; then
  build_one "$HERE/build_det" "$HERE/../libprosper_hip_det.so" "-DPM_DETERMINISTIC"
fi
