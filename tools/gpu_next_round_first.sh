#!/usr/bin/env bash
# First GPU call of the next round (about 25 GPU-minutes; every leg is time-boxed and writes its own log under gpurun_out/, so a
# cut-off call still leaves the earlier results).  Order = value per GPU-minute:
#   1. the GPU tests written after round 1's GPU budget was spent (tests/test_zz_late_gpu.py)
#   2. per-shape GEMM tile tuner -> gpurun_out/<tag>_gemm_tuned.h, applied + rebuilt + parity-checked + benched on the box
#      (afterwards: copy gpurun_out/<tag>_gemm_tuned.h to musev_amd/csrc/gemm_tuned.h in the repo and commit it)
#   3. kernel-level A/B: GEMM variants 2 / 8, attention variants 3 / 11 / 19 / 35 / 51
#   4. whole-step A/B of the tile order (plain m-major vs groups of 8 m-tiles on wide grids)
#   5. whole-step run with the 256x320 / 256x256 tiles (MUSEV_GEMM_VARIANT=8) + the matrix of tools/gpu_bigtile_diag.sh
set -u
TAG=${1:-r02a}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
( timeout 420 python -m pytest tests/test_zz_late_gpu.py -m gpu -q 2>&1 | tail -12 ) > $OUT/${TAG}_pytest_late.log
cat $OUT/${TAG}_pytest_late.log
( timeout 420 python tools/gpu_gemm_tune.py ${TAG} 2>&1 | tail -45 ) > $OUT/${TAG}_gemm_tune.log
cat $OUT/${TAG}_gemm_tune.log
# 2b. apply the table on the box (hipcc is in the image: ~1 min), re-check GEMM parity, bench with and without it
if [ -s $OUT/${TAG}_gemm_tuned.h ]; then
  cp musev_amd/csrc/gemm_tuned.h $OUT/${TAG}_gemm_tuned_before.h
  cp $OUT/${TAG}_gemm_tuned.h musev_amd/csrc/gemm_tuned.h
  ( bash musev_amd/csrc/build.sh 2>&1 | tail -2 ) > $OUT/${TAG}_rebuild.log; cat $OUT/${TAG}_rebuild.log
  ( timeout 300 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -m gpu -q -k "gemm or conv or geglu or small" 2>&1 | tail -4 ) > $OUT/${TAG}_pytest_tuned.log
  cat $OUT/${TAG}_pytest_tuned.log
  ( timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-1200 ) > $OUT/${TAG}_bench_tuned.log
  ( MUSEV_GEMM_FORCE=-2 timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-1200 ) > $OUT/${TAG}_bench_rules.log
  ( MUSEV_HALF_STREAMS=0 timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-1200 ) > $OUT/${TAG}_bench_tuned_one_stream.log
  cat $OUT/${TAG}_bench_tuned.log $OUT/${TAG}_bench_rules.log $OUT/${TAG}_bench_tuned_one_stream.log
fi
( timeout 420 python tools/gpu_gemm_ab.py ${TAG}_ab 2 8 2>&1 | tail -60 ) > $OUT/${TAG}_kernel_ab.log
tail -45 $OUT/${TAG}_kernel_ab.log
( MUSEV_GEMM_TILE_GROUP=0 timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-1200 ) > $OUT/${TAG}_bench_mmajor.log
( timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-1200 ) > $OUT/${TAG}_bench_grouped.log
cat $OUT/${TAG}_bench_mmajor.log $OUT/${TAG}_bench_grouped.log
( MUSEV_GEMM_VARIANT=8 timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-1200 ) > $OUT/${TAG}_bench_v8.log
cat $OUT/${TAG}_bench_v8.log
bash tools/gpu_bigtile_diag.sh $TAG
