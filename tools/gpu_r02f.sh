#!/usr/bin/env bash
# Round 2, GPU call 6: the LDS-DMA attention kernel (attn3): parity cases, timings at the config-2 shapes, model goldens, bench.
set -u
TAG=${1:-r02f}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
( timeout 300 python tools/gpu_attn_bench.py 2>&1 | tail -30 ) > $OUT/${TAG}_attn_bench.log; cat $OUT/${TAG}_attn_bench.log
( timeout 900 python -m pytest tests/test_model_gpu.py -m gpu -q -s 2>&1 | grep -E "delta|passed|failed|Error|assert" | tail -12 ) > $OUT/${TAG}_pytest_model.log; cat $OUT/${TAG}_pytest_model.log
( timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-config4 2>&1 | tail -1 | cut -c1-1500 ) > $OUT/${TAG}_bench.log; cat $OUT/${TAG}_bench.log
