#!/usr/bin/env bash
# Round 2, GPU call 3: whole GPU suite on the new GroupNorm / config-1 test, ping-pong GEMM schedule A/B (catalogue ids 19-22 vs 6 / 7
# vs the table), bench.
set -u
TAG=${1:-r02c}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
( timeout 300 python tools/gpu_gemm_ab.py ${TAG} -1 6 7 19 20 21 22 2>&1 | tail -45 ) > $OUT/${TAG}_gemm_ab.log; cat $OUT/${TAG}_gemm_ab.log
( timeout 200 python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-1800 ) > $OUT/${TAG}_bench.log; cat $OUT/${TAG}_bench.log
( timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -12 ) > $OUT/${TAG}_pytest_gpu.log; cat $OUT/${TAG}_pytest_gpu.log
