#!/usr/bin/env bash
# A/B of the two-stream CFG-half experiment: loop parity tests + short bench, with and without MUSEV_HALF_STREAMS=1
set -u
TAG=${1:-ab}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
( MUSEV_HALF_STREAMS=1 timeout 600 python -m pytest tests/test_pipeline_gpu.py -m gpu -x -q 2>&1 | tail -4 ) > $OUT/${TAG}_streams_pytest.log
for v in 0 1; do
  ( MUSEV_HALF_STREAMS=$v timeout 600 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-roofline 2>&1 | tail -1 | cut -c1-260 ) > $OUT/${TAG}_streams_bench$v.log
done
tail -3 $OUT/${TAG}_streams_pytest.log; cat $OUT/${TAG}_streams_bench0.log; echo; cat $OUT/${TAG}_streams_bench1.log
