#!/usr/bin/env bash
# Round 2, GPU call: tuner over the full catalogue (incl. the ping-pong schedules), table applied + rebuilt, whole GPU suite, bench.
set -u
TAG=${1:-r02h}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
( timeout 600 python tools/gpu_gemm_tune.py ${TAG} 2>&1 | tail -40 ) > $OUT/${TAG}_gemm_tune.log; tail -30 $OUT/${TAG}_gemm_tune.log
if [ -s $OUT/${TAG}_gemm_tuned.h ]; then
  cp $OUT/${TAG}_gemm_tuned.h musev_amd/csrc/gemm_tuned.h
  ( bash musev_amd/csrc/build.sh 2>&1 | tail -1 ) > $OUT/${TAG}_rebuild.log; cat $OUT/${TAG}_rebuild.log
fi
( timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-config4 2>&1 | tail -1 | cut -c1-1500 ) > $OUT/${TAG}_bench_tuned.log; cat $OUT/${TAG}_bench_tuned.log
( timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 ) > $OUT/${TAG}_pytest_gpu.log; cat $OUT/${TAG}_pytest_gpu.log
