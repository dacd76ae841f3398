#!/usr/bin/env bash
# Round 2, GPU call: tuner re-run on the rewritten epilogue (catalogue ids 0..18: the whole-step A/B of r02h kept the ping-pong
# tiles out of the table), table applied + rebuilt on the box, whole-step bench with the old and the new table.
set -u
TAG=${1:-r02p}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
B="--steps 20 --warmup 3 --no-cpu-baseline --no-config4 --no-roofline"
( timeout 200 python bench.py $B 2>&1 | tail -1 | cut -c1-400 ) > $OUT/${TAG}_bench_old_table.log; cat $OUT/${TAG}_bench_old_table.log
( timeout 600 python tools/gpu_gemm_tune.py ${TAG} --max-cfg 18 2>&1 | tail -30 ) > $OUT/${TAG}_gemm_tune.log; cat $OUT/${TAG}_gemm_tune.log
if [ -s $OUT/${TAG}_gemm_tuned.h ]; then
  cp $OUT/${TAG}_gemm_tuned.h musev_amd/csrc/gemm_tuned.h
  ( bash musev_amd/csrc/build.sh 2>&1 | tail -1 ) > $OUT/${TAG}_rebuild.log; cat $OUT/${TAG}_rebuild.log
  ( timeout 200 python bench.py $B 2>&1 | tail -1 | cut -c1-400 ) > $OUT/${TAG}_bench_new_table.log; cat $OUT/${TAG}_bench_new_table.log
  ( MUSEV_HALF_STREAMS=0 timeout 200 python bench.py $B 2>&1 | tail -1 | cut -c1-400 ) > $OUT/${TAG}_bench_new_table_one_stream.log; cat $OUT/${TAG}_bench_new_table_one_stream.log
fi
