#!/usr/bin/env bash
# Round-2 starter: find out why bench.py with the 256x320 / 256x256 GEMM tiles (MUSEV_GEMM_VARIANT=8) did not finish in
# r01n.  Each leg is time-boxed; the matrix isolates hipGraph replay and the two-stream schedule.
set -u
TAG=${1:-diag}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
for graph in 1 0; do
  for streams in 0 1; do
    name=${TAG}_v8_nograph${graph}_streams${streams}
    ( MUSEV_GEMM_VARIANT=8 MUSEV_NO_GRAPH=$graph MUSEV_HALF_STREAMS=$streams timeout 150 python bench.py --steps 4 --warmup 1 \
        --no-cpu-baseline --no-roofline 2>&1 | tail -2 | cut -c1-300 ; echo "rc=$?" ) > $OUT/$name.log
    echo "== $name"; cat $OUT/$name.log
  done
done
