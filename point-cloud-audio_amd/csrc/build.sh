#!/bin/bash
# Build libpca_hip.so for gfx950 (MI355X).  hipcc cross-compiles without a GPU.
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
ROOT="$(cd "$HERE/../.." && pwd)"
# Diagnostic variants (extra -D flags: PCA_EXTRA_FLAGS) never share objects or the output with the
# product build: they need PCA_BUILD_DIR and PCA_OUT of their own (scripts/experiments/*.sh), so a
# later plain build.sh cannot link a diagnostic object into libpca_hip.so.
OUT="${PCA_OUT:-$HERE/../pca_hip/libpca_hip.so}"
BUILD="${PCA_BUILD_DIR:-$HERE/build}"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
EXTRA="${PCA_EXTRA_FLAGS:-}"
if [ -n "$EXTRA" ] && { [ -z "${PCA_BUILD_DIR:-}" ] || [ -z "${PCA_OUT:-}" ]; }; then
  echo "build.sh: PCA_EXTRA_FLAGS needs PCA_BUILD_DIR and PCA_OUT (a diagnostic build must not touch the product's objects)" >&2
  exit 2
fi
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -I$ROOT/include -I$HERE -Wall -Wno-unused-function $EXTRA"
mkdir -p "$BUILD"
objs=()
pids=()
for src in "$HERE"/*.hip; do
  obj="$BUILD/$(basename "${src%.hip}").o"
  objs+=("$obj")
  if [ ! -f "$obj" ] || [ "$src" -nt "$obj" ] || [ "$HERE/pca_common.h" -nt "$obj" ] || \
     [ "$ROOT/include/pca_hip.h" -nt "$obj" ] || \
     { ls "$HERE"/*.hpp >/dev/null 2>&1 && [ -n "$(find "$HERE" -name '*.hpp' -newer "$obj")" ]; }; then
    $HIPCC $FLAGS -c "$src" -o "$obj" &
    pids+=($!)
  fi
done
for p in "${pids[@]:-}"; do [ -n "$p" ] && wait "$p"; done
$HIPCC -shared -fPIC --offload-arch=gfx950 -Wl,-z,defs "${objs[@]}" -o "$OUT"
echo "built $OUT"
