#!/usr/bin/env bash
# Final measurement of a build: parity tests, smoke, the default bench line, rocprofv3 kernel trace + stats of the bench
# command, and two PMC passes (FETCH_SIZE, WRITE_SIZE) over the GEMM launches for the roofline "traffic" figure.
set -u
TAG=${1:-r01f}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
if [ "${SKIP_AB:-0}" != "1" ]; then
  ( timeout 600 python tools/gpu_gemm_ab.py $TAG ${VARIANTS:-2} 2>&1 | tail -60 ) > $OUT/${TAG}_gemm_ab.log
fi
if [ "${PROFILE_ONLY:-0}" != "1" ]; then
( timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > $OUT/${TAG}_pytest_gpu.log
( timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5 ) > $OUT/${TAG}_smoke.log
( timeout 900 python bench.py 2>&1 | tail -2 ) > $OUT/${TAG}_bench.log
fi
cd /tmp
# per-kernel durations are profiled on ONE stream (MUSEV_HALF_STREAMS=0), like bench.py's own instrumented roofline pass
( MUSEV_HALF_STREAMS=0 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof -o bench -- python $ROOT/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-roofline --no-config4 2>&1 | tail -3 ) > $OUT/${TAG}_rocprof.log
for c in ${PMC_COUNTERS-FETCH_SIZE WRITE_SIZE}; do
  ( MUSEV_NO_GRAPH=1 MUSEV_HALF_STREAMS=0 timeout 900 rocprofv3 --pmc $c --output-format csv -d $OUT/${TAG}_pmc_$c -o pmc -- python $ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-config4 2>&1 | tail -3 ) > $OUT/${TAG}_pmc_$c.log
done
cd $ROOT
find $OUT/${TAG}_prof -name "*kernel_trace.csv" -size +30M -delete
python tools/pmc_summary.py $TAG > $OUT/${TAG}_pmc_summary.log 2>&1
find $OUT -name "*counter_collection.csv" -size +20M -delete
tail -30 $OUT/${TAG}_gemm_ab.log 2>/dev/null
tail -3 $OUT/${TAG}_pytest_gpu.log; tail -2 $OUT/${TAG}_smoke.log; tail -1 $OUT/${TAG}_bench.log | cut -c1-2500; tail -30 $OUT/${TAG}_pmc_summary.log
