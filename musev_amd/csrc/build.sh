#!/usr/bin/env bash
# Builds libmusev_hip.so (gfx950 only) next to this script.  hipcc cross-compiles without a GPU.
#   MV_EXTRA_FLAGS  extra compiler flags of an EXPERIMENT build (e.g. -DMV_EXPERIMENT, -DMV_TIMELINE)
#   MV_LIB_NAME     output library (default libmusev_hip.so); experiment builds go to their own name + object directory
#                   (e.g. MV_LIB_NAME=libmusev_hip_exp.so), so the product library is never replaced by one
set -euo pipefail
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
LIB=${MV_LIB_NAME:-libmusev_hip.so}
OBJ=build
if [ "$LIB" != "libmusev_hip.so" ]; then OBJ=build_${LIB%.so}; fi
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast ${MV_EXTRA_FLAGS:-}"
mkdir -p $OBJ
pids=()
for f in lib gemm norm attention elementwise ffn tsa xab; do
  if [ ! -f $OBJ/$f.o ] || [ $f.hip -nt $OBJ/$f.o ] || [ common.h -nt $OBJ/$f.o ] || [ gemm_tuned.h -nt $OBJ/$f.o ] || [ ../../include/musev_hip.h -nt $OBJ/$f.o ]; then
    $HIPCC $FLAGS -c $f.hip -o $OBJ/$f.o &
    pids+=($!)
  fi
done
for p in "${pids[@]:-}"; do [ -n "$p" ] && wait $p; done
$HIPCC --offload-arch=gfx950 -shared -fPIC -o $LIB $OBJ/lib.o $OBJ/gemm.o $OBJ/norm.o $OBJ/attention.o $OBJ/elementwise.o $OBJ/ffn.o $OBJ/tsa.o $OBJ/xab.o
echo "built $(pwd)/$LIB"
