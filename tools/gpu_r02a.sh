#!/usr/bin/env bash
# Round 2, GPU call 1 (every leg time-boxed, own log under gpurun_out/): measure what round 1 left unmeasured on HEAD.
#   1. per-shape tile tuner over the 25-entry catalogue -> <tag>_gemm_tuned.h / <tag>_gemm_tune.json
#   2. table applied + rebuilt on the box, GEMM parity re-checked, whole-step bench: table vs rules vs m-major tile order
#   3. variant 8 (256x320 / 256x256 tiles) whole-step run, time-boxed (the r01n stall)
#   4. attention variant A/B
#   5. rocprofv3 --kernel-trace --stats of the tuned tree
set -u
TAG=${1:-r02a}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
B="--steps 10 --warmup 2 --no-cpu-baseline"
( timeout 420 python tools/gpu_gemm_tune.py ${TAG} 2>&1 | tail -50 ) > $OUT/${TAG}_gemm_tune.log
cat $OUT/${TAG}_gemm_tune.log
( timeout 200 python bench.py $B 2>&1 | tail -1 | cut -c1-1500 ) > $OUT/${TAG}_bench_rules.log
( MUSEV_GEMM_TILE_GROUP=0 timeout 200 python bench.py $B --no-roofline 2>&1 | tail -1 | cut -c1-600 ) > $OUT/${TAG}_bench_mmajor.log
( MUSEV_GEMM_VARIANT=8 timeout 150 python bench.py $B --no-roofline 2>&1 | tail -1 | cut -c1-600; echo "rc=$?" ) > $OUT/${TAG}_bench_v8.log
cat $OUT/${TAG}_bench_rules.log $OUT/${TAG}_bench_mmajor.log $OUT/${TAG}_bench_v8.log
if [ -s $OUT/${TAG}_gemm_tuned.h ]; then
  cp $OUT/${TAG}_gemm_tuned.h musev_amd/csrc/gemm_tuned.h
  ( bash musev_amd/csrc/build.sh 2>&1 | tail -2 ) > $OUT/${TAG}_rebuild.log; cat $OUT/${TAG}_rebuild.log
  ( timeout 300 python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "gemm or conv or geglu" 2>&1 | tail -4 ) > $OUT/${TAG}_pytest_tuned.log
  cat $OUT/${TAG}_pytest_tuned.log
  ( timeout 200 python bench.py $B 2>&1 | tail -1 | cut -c1-1500 ) > $OUT/${TAG}_bench_tuned.log
  cat $OUT/${TAG}_bench_tuned.log
fi
( timeout 300 python tools/gpu_gemm_ab.py ${TAG}_ab 2 2>&1 | grep -E "^attn variant [0-9]+:|variant 2:" | tail -12 ) > $OUT/${TAG}_attn_ab.log
cat $OUT/${TAG}_attn_ab.log
cd /tmp
( MUSEV_HALF_STREAMS=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof -o bench -- python $ROOT/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-roofline 2>&1 | tail -2 ) > $OUT/${TAG}_rocprof.log
cd $ROOT
find $OUT/${TAG}_prof -name "*kernel_trace.csv" -size +30M -delete
find $OUT/${TAG}_prof -name "*kernel_stats.csv" | head -1 | xargs -r head -25
