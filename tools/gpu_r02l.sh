#!/usr/bin/env bash
# Round 2, GPU call: block timelines of the GEMM kernel (experiment build with -DMV_TIMELINE, on the box only).
set -u
TAG=${1:-r02l}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
rm -f musev_amd/csrc/build/gemm.o
( MV_EXTRA_FLAGS=-DMV_TIMELINE bash musev_amd/csrc/build.sh 2>&1 | tail -1 )
( timeout 300 python tools/gpu_gemm_timeline.py 2>&1 | tail -40 ) > $OUT/${TAG}_gemm_timeline.log; cat $OUT/${TAG}_gemm_timeline.log
