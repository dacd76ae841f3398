#!/usr/bin/env bash
# First GPU call of the next round (about 6 GPU-minutes): validates what round 1 could not.
#   1. pytest tests/test_zz_late_gpu.py (checks written after round 1's GPU budget was spent: Euler loop, ReferenceNet2D, ...)
#   2. tools/gpu_bigtile_diag.sh        (why bench.py with MUSEV_GEMM_VARIANT=8 -- 256x320 / 256x256 tiles -- stalled in r01n)
#   3. bench.py with the big tiles if (2) is clean: expected -3..-4 ms per step
set -u
TAG=${1:-r02a}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
( timeout 300 python -m pytest tests/test_zz_late_gpu.py -m gpu -q 2>&1 | tail -8 ) > $OUT/${TAG}_pytest_pending.log
cat $OUT/${TAG}_pytest_pending.log
bash tools/gpu_bigtile_diag.sh $TAG
( MUSEV_GEMM_VARIANT=8 timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-1200 ) > $OUT/${TAG}_bench_v8.log
cat $OUT/${TAG}_bench_v8.log
# 4. tile order A/B (written after round 1's GPU budget): default (groups of 8 m-tiles on wide grids) vs plain m-major
( MUSEV_GEMM_TILE_GROUP=0 timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-1200 ) > $OUT/${TAG}_bench_mmajor.log
( timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-1200 ) > $OUT/${TAG}_bench_grouped.log
cat $OUT/${TAG}_bench_mmajor.log $OUT/${TAG}_bench_grouped.log
# 5. kernel-level A/B incl. the attention variants (3 default, 11 pkrtz, 19 buffer-descriptor K/V fetch) and GEMM variants 2 / 8
( timeout 420 python tools/gpu_gemm_ab.py ${TAG}_ab 2 8 2>&1 | tail -60 ) > $OUT/${TAG}_kernel_ab.log
tail -45 $OUT/${TAG}_kernel_ab.log
# 6. per-shape tile tuner: writes gpurun_out/${TAG}_gemm_tuned.h (copy to musev_amd/csrc/gemm_tuned.h, rebuild, re-bench)
( timeout 420 python tools/gpu_gemm_tune.py ${TAG} 2>&1 | tail -45 ) > $OUT/${TAG}_gemm_tune.log
cat $OUT/${TAG}_gemm_tune.log
