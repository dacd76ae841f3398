#!/usr/bin/env bash
# Final measurement of a build (run through gpurun):  bash tools/gpu_profile.sh <tag>
#   1. the bench line of the command the driver runs (python bench.py --steps 20 --warmup 5)
#   2. rocprofv3 --kernel-trace --stats of THE SAME command (the profiler serialises the two HIP streams, so the per-kernel
#      durations are the kernels' own -- what bench.py's one-stream roofline replay measures as well)
#   3. two PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs, kernel trace / stats only) for the roofline "traffic" figure:
#      eager launches (MUSEV_NO_GRAPH=1), the loop's own stream setting (batch-1 launches, like the timed path)
# Copy gpurun_out/<tag>_* summaries into profiles/ afterwards (tools/pmc_summary.py writes <tag>_pmc_summary.json).
set -u
TAG=${1:-r03z}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
( timeout 900 python bench.py --steps 20 --warmup 5 2>&1 | tail -1 ) > $OUT/${TAG}_bench.json
cd /tmp
( timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof -o bench -- python $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -2 ) > $OUT/${TAG}_rocprof.log
for c in ${PMC_COUNTERS-FETCH_SIZE WRITE_SIZE}; do
  ( MUSEV_NO_GRAPH=1 timeout 900 rocprofv3 --pmc $c --output-format csv -d $OUT/${TAG}_pmc_$c -o pmc -- python $ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-config4 2>&1 | tail -2 ) > $OUT/${TAG}_pmc_$c.log
done
cd $ROOT
find $OUT/${TAG}_prof -name "*kernel_trace.csv" -delete
cp $(find $OUT/${TAG}_prof -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_rocprofv3_kernel_stats.csv 2>/dev/null
python tools/pmc_summary.py $TAG > $OUT/${TAG}_pmc_summary.log 2>&1
# per-dispatch rows of the implicit-GEMM kernels (small) stay, for tools/pmc_by_problem.py; the full counter tables do not
for c in ${PMC_COUNTERS-FETCH_SIZE WRITE_SIZE}; do
  f=$(find $OUT/${TAG}_pmc_$c -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && ( head -1 "$f"; grep -E "gemm2_kernel|splitk_reduce|ffn_geglu_kernel|tsa_kernel|xab_kernel" "$f" ) > $OUT/${TAG}_pmc_${c}_gemm_rows.csv
done
( MUSEV_NO_GRAPH=1 timeout 300 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-config4 --dump-gemm-launches $OUT/${TAG}_gemm_launches.json 2>&1 | tail -1 ) > /dev/null
python tools/pmc_by_problem.py $TAG > $OUT/${TAG}_pmc_by_problem.log 2>&1
find $OUT -name "*counter_collection.csv" -delete
cut -c1-1500 $OUT/${TAG}_bench.json; tail -2 $OUT/${TAG}_rocprof.log; head -12 $OUT/${TAG}_rocprofv3_kernel_stats.csv | cut -c1-160; tail -25 $OUT/${TAG}_pmc_summary.log
