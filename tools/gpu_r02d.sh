#!/usr/bin/env bash
# Round 2, GPU call 4: rocprofv3 kernel stats of the bench (GroupNorm two-launch form), K sweep of the GEMM kernel.
set -u
TAG=${1:-r02d}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
( timeout 300 python tools/gpu_ksweep.py ${TAG} 2>&1 | tail -40 ) > $OUT/${TAG}_ksweep.log; cat $OUT/${TAG}_ksweep.log
cd /tmp
( MUSEV_HALF_STREAMS=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof -o bench -- python $ROOT/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-roofline 2>&1 | tail -2 ) > $OUT/${TAG}_rocprof.log
cd $ROOT
find $OUT/${TAG}_prof -name "*kernel_trace.csv" -size +30M -delete
find $OUT/${TAG}_prof -name "*kernel_stats.csv" | head -1 | xargs -r head -22 | cut -c1-200
