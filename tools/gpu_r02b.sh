#!/usr/bin/env bash
# Round 2, GPU call 2: the parity tests written this round (BASELINE-size goldens / kernel cases, config-1 loop, 20-step drift,
# uniform_v2), the bench with the split-K rule, the tuner re-run over the cleaned-up catalogue incl. split factors.
set -u
TAG=${1:-r02b}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
B="--steps 10 --warmup 2 --no-cpu-baseline"
( timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x 2>&1 | tail -8 ) > $OUT/${TAG}_pytest_kernels.log; cat $OUT/${TAG}_pytest_kernels.log
( timeout 900 python -m pytest tests/test_model_gpu.py -m gpu -q -s -k "golden and (cfg2 or cfg3)" 2>&1 | grep -E "delta|passed|failed|Error" | tail -8 ) > $OUT/${TAG}_pytest_at_size.log; cat $OUT/${TAG}_pytest_at_size.log
( timeout 900 python -m pytest tests/test_pipeline_gpu.py -m gpu -q -s -k "config1 or twenty or uniform_v2" 2>&1 | grep -E "step|delta|passed|failed|Error|assert" | tail -40 ) > $OUT/${TAG}_pytest_loop.log; cat $OUT/${TAG}_pytest_loop.log
( timeout 200 python bench.py $B 2>&1 | tail -1 | cut -c1-1800 ) > $OUT/${TAG}_bench_split.log; cat $OUT/${TAG}_bench_split.log
( MUSEV_GEMM_SPLITK=1 timeout 200 python bench.py $B --no-roofline 2>&1 | tail -1 | cut -c1-400 ) > $OUT/${TAG}_bench_nosplit.log; cat $OUT/${TAG}_bench_nosplit.log
( timeout 500 python tools/gpu_gemm_tune.py ${TAG} 2>&1 | tail -70 ) > $OUT/${TAG}_gemm_tune.log; cat $OUT/${TAG}_gemm_tune.log
if [ -s $OUT/${TAG}_gemm_tuned.h ]; then
  cp $OUT/${TAG}_gemm_tuned.h musev_amd/csrc/gemm_tuned.h
  ( bash musev_amd/csrc/build.sh 2>&1 | tail -2 ) > $OUT/${TAG}_rebuild.log; cat $OUT/${TAG}_rebuild.log
  ( timeout 200 python bench.py $B 2>&1 | tail -1 | cut -c1-1800 ) > $OUT/${TAG}_bench_tuned.log; cat $OUT/${TAG}_bench_tuned.log
fi
