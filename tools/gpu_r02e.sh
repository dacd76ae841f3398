#!/usr/bin/env bash
# Round 2, GPU call 5: side-model tests (VAE / ImageProj / multi-shot), new kernel cases, bench lines of configs 2 (+ config-4 1-GPU
# denominator), 3, 4, 5; rocprofv3 kernel stats (GroupNorm three-launch form with the cheap fold).
set -u
TAG=${1:-r02e}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_side_models.py tests/test_kernels_gpu.py -m gpu -q -s -k "side or vae or decode or multi_shot or softmax or units_reduce or groupnorm" 2>&1 | grep -E "delta|passed|failed|Error|error|assert" | tail -20 ) > $OUT/${TAG}_pytest_side.log; cat $OUT/${TAG}_pytest_side.log
( timeout 400 python bench.py --steps 20 --warmup 3 2>&1 | tail -1 ) > $OUT/${TAG}_bench_config2.json; cut -c1-2200 $OUT/${TAG}_bench_config2.json
( timeout 300 python bench.py --workload config3 --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 ) > $OUT/${TAG}_bench_config3.json; cut -c1-900 $OUT/${TAG}_bench_config3.json
( timeout 300 python bench.py --workload config4 --steps 3 --warmup 1 --no-cpu-baseline --no-roofline 2>&1 | tail -1 ) > $OUT/${TAG}_bench_config4.json; cut -c1-700 $OUT/${TAG}_bench_config4.json
( timeout 400 python bench.py --workload config5 --steps 2 --warmup 1 --no-cpu-baseline --no-roofline 2>&1 | tail -1 ) > $OUT/${TAG}_bench_config5.json; cut -c1-900 $OUT/${TAG}_bench_config5.json
cd /tmp
( MUSEV_HALF_STREAMS=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof -o bench -- python $ROOT/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-roofline --no-config4 2>&1 | tail -2 ) > $OUT/${TAG}_rocprof.log
cd $ROOT
find $OUT/${TAG}_prof -name "*kernel_trace.csv" -size +30M -delete
find $OUT/${TAG}_prof -name "*kernel_stats.csv" | head -1 | xargs -r grep -E "gn_|layernorm|attn" | cut -c1-160
