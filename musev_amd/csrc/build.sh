#!/usr/bin/env bash
# Builds libmusev_hip.so (gfx950 only) next to this script.  hipcc cross-compiles without a GPU.
set -euo pipefail
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast ${MV_EXTRA_FLAGS:-}"   # MV_EXTRA_FLAGS: experiment builds (e.g. -DMV_TIMELINE)
mkdir -p build
pids=()
for f in lib gemm norm attention elementwise; do
  if [ ! -f build/$f.o ] || [ $f.hip -nt build/$f.o ] || [ common.h -nt build/$f.o ] || [ gemm_tuned.h -nt build/$f.o ] || [ ../../include/musev_hip.h -nt build/$f.o ]; then
    $HIPCC $FLAGS -c $f.hip -o build/$f.o &
    pids+=($!)
  fi
done
for p in "${pids[@]:-}"; do [ -n "$p" ] && wait $p; done
$HIPCC --offload-arch=gfx950 -shared -fPIC -o libmusev_hip.so build/lib.o build/gemm.o build/norm.o build/attention.o build/elementwise.o
echo "built $(pwd)/libmusev_hip.so"
