#!/usr/bin/env bash
# One gpurun call: GPU parity tests, smoke, bench A/B of the GEMM staging variants, rocprofv3 kernel-trace of the bench.
# usage (from the repo root on the GPU box):  bash tools/gpu_round.sh <tag>
set -u
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
( timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > $OUT/${TAG}_pytest_gpu.log
( timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5 ) > $OUT/${TAG}_smoke.log
for v in 0 1; do
  ( MUSEV_GEMM_VARIANT=$v timeout 900 python bench.py --steps 6 --warmup 2 --no-cpu-baseline 2>&1 | tail -2 ) > $OUT/${TAG}_bench_variant$v.log
done
cd /tmp
( timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_prof -o bench -- python $ROOT/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-roofline 2>&1 | tail -3 ) > $OUT/${TAG}_rocprof.log
cd $ROOT
find $OUT/${TAG}_prof -name "*kernel_stats*" | head -3
f=$(find $OUT/${TAG}_prof -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && head -60 "$f" > $OUT/${TAG}_kernel_stats_top.csv
# the raw per-dispatch trace is large: keep only the stats
find $OUT/${TAG}_prof -name "*kernel_trace.csv" -size +20M -delete
cat $OUT/${TAG}_pytest_gpu.log | tail -3; cat $OUT/${TAG}_smoke.log | tail -2
for v in 0 1; do python - <<PY
import json
try:
    l=[x for x in open("$OUT/${TAG}_bench_variant$v.log") if x.startswith("{")][-1]; d=json.loads(l)
    print("variant $v", d["value"], d["ms_per_step"], d["roofline"]["achieved"], d["roofline"]["by_mode"])
except Exception as e: print("variant $v: no json", e)
PY
done
