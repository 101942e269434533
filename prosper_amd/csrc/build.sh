#!/bin/bash
# Build libprosper_hip.so for gfx950 in-tree (cross-compiles without a GPU).
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
ROOT="$(cd "$HERE/../.." && pwd)"
OUT="$HERE/../libprosper_hip.so"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wall -Wno-unused-function -I$ROOT/include -I$HERE"
mkdir -p "$HERE/build"
objs=()
for src in "$HERE"/*.hip; do
  obj="$HERE/build/$(basename "${src%.hip}").o"
  if [ ! -f "$obj" ] || [ "$src" -nt "$obj" ] || [ "$HERE/pm_common.h" -nt "$obj" ] || [ "$ROOT/include/prosper_hip.h" -nt "$obj" ]; then
    echo "hipcc $(basename "$src")"
    "$HIPCC" $FLAGS ${PM_EXTRA_FLAGS:-} -c "$src" -o "$obj" &
  fi
  objs+=("$obj")
done
wait
"$HIPCC" --offload-arch=gfx950 -shared -fPIC -o "$OUT" "${objs[@]}"
echo "built $OUT"
