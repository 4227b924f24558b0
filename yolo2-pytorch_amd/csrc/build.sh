#!/bin/bash
# Build libyolo2_hip.so for gfx950 (MI355X) in-tree.  hipcc cross-compiles without a GPU.
#   -ffp-contract=off : IoU/NMS arithmetic must keep the reference's one-rounding-per-op fp32 sequence
set -e
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
ROOT="$(cd "$HERE/../.." && pwd)"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
OUT="${Y2_OUT:-$HERE/libyolo2_hip.so}"          # experiments: Y2_OUT / Y2_OBJ / Y2_EXTRA_FLAGS build an A/B variant beside the product library
OBJ="${Y2_OBJ:-$HERE/build}"
mkdir -p "$OBJ"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -I$ROOT/include -I$HERE $Y2_EXTRA_FLAGS"
pids=()
for f in "$HERE"/*.hip; do
    o="$OBJ/$(basename "${f%.hip}").o"
    if [ ! -f "$o" ] || [ "$f" -nt "$o" ] || [ "$HERE/common.h" -nt "$o" ] || [ "$ROOT/include/yolo2_hip.h" -nt "$o" ]; then
        extra="$(sed -n 's|^// y2-build-flags: ||p' "$f" | head -1)"          # per-file code generation options, stated in the source
        $HIPCC $FLAGS $extra -c "$f" -o "$o" &
        pids+=($!)
    fi
done
for p in "${pids[@]}"; do wait "$p"; done
$HIPCC --offload-arch=gfx950 -shared -fPIC -o "$OUT" "$OBJ"/*.o
echo "built $OUT"
