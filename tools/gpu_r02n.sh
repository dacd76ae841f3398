#!/usr/bin/env bash
# Round 2, GPU call: GEMM epilogue v2 (accumulator init after the prologue issue, batched LDS reads, write-ahead): bench + timelines.
set -u
TAG=${1:-r02n}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
( timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-config4 2>&1 | tail -1 | cut -c1-1800 ) > $OUT/${TAG}_bench.log; cut -c1-400 $OUT/${TAG}_bench.log
( timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "gemm or conv or geglu" 2>&1 | tail -3 ) > $OUT/${TAG}_pytest_kernels.log; cat $OUT/${TAG}_pytest_kernels.log
rm -f musev_amd/csrc/build/gemm.o
( MV_EXTRA_FLAGS=-DMV_TIMELINE bash musev_amd/csrc/build.sh 2>&1 | tail -1 )
( timeout 300 python tools/gpu_gemm_timeline.py 2>&1 | tail -40 ) > $OUT/${TAG}_gemm_timeline.log; cat $OUT/${TAG}_gemm_timeline.log
