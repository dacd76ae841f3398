#!/usr/bin/env bash
# End-of-round measurement of a build (through gpurun, ~12 GPU-minutes):  bash tools/gpu_final_profile.sh <tag>
#   1. tools/gpu_profile.sh: the driver's bench line, rocprofv3 kernel stats of the same command, PMC FETCH_SIZE / WRITE_SIZE passes
#      (-> <tag>_pmc_summary.json, <tag>_pmc_by_problem.log, <tag>_gemm_launches.json)
#   2. per-problem timing table of the step's matrix launches (+ torch.matmul yardstick)
#   3. bench lines WITH the roofline block for configs 3 and 5
#   4. tools/gpu_sq_counters.sh: SQ counter passes (MFMA busy / VALU / LDS / waits per kernel)
set -u
TAG=${1:-r05final}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
bash tools/gpu_profile.sh $TAG > $OUT/${TAG}_profile_stdout.log 2>&1
cut -c1-600 $OUT/${TAG}_bench.json
( timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-config4 --gemm-by-problem $OUT/${TAG}_gemm_by_problem.json 2>&1 | tail -1 | cut -c1-200 )
for w in config3 config5; do
  ( timeout 400 python bench.py --workload $w --steps 8 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 ) > $OUT/${TAG}_bench_$w.json
  cut -c1-300 $OUT/${TAG}_bench_$w.json
done
bash tools/gpu_sq_counters.sh $TAG > $OUT/${TAG}_sq_stdout.log 2>&1
tail -45 $OUT/${TAG}_sq_counters.log
