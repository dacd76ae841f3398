#!/usr/bin/env bash
# Round 2, GPU call: MFMA temporal attention + one-launch GroupNorm (small slabs): parity, per-launch timing, whole-step bench,
# kernel stats of the step.
set -u
TAG=${1:-r02i}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
( timeout 300 python tools/gpu_norm_tattn_bench.py 2>&1 | tail -50 ) > $OUT/${TAG}_norm_tattn.log; cat $OUT/${TAG}_norm_tattn.log
( timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-config4 2>&1 | tail -1 | cut -c1-1500 ) > $OUT/${TAG}_bench.log; cat $OUT/${TAG}_bench.log
( timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -m gpu -x -q 2>&1 | tail -6 ) > $OUT/${TAG}_pytest_gpu.log; cat $OUT/${TAG}_pytest_gpu.log
cd /tmp
( MUSEV_HALF_STREAMS=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof -o bench -- python $ROOT/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-roofline --no-config4 2>&1 | tail -2 ) > $OUT/${TAG}_rocprof.log
cd $ROOT
find $OUT/${TAG}_prof -name "*kernel_trace.csv" -size +30M -delete
find $OUT/${TAG}_prof -name "*kernel_stats.csv" | head -1 | xargs -r head -30 | cut -c1-200
