#!/usr/bin/env bash
# Round 2, GPU call: the multi-rank path on the real kernels with N gloo ranks sharing the box's one GPU (loop tests + bench rehearsal)
set -u
TAG=${1:-r02v}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
( timeout 900 python -m pytest tests/test_pipeline_gpu.py -m gpu -x -q -k "sharing" 2>&1 | tail -15 ) > $OUT/${TAG}_pytest_shared_gpu.log; cat $OUT/${TAG}_pytest_shared_gpu.log
for n in 2 4; do
( timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus $n --steps 2 --warmup 1 --rehearse-shared-gpu --no-cpu-baseline 2>&1 | tail -3 | cut -c1-1200 ) > $OUT/${TAG}_bench_rehearsal_n$n.log; cat $OUT/${TAG}_bench_rehearsal_n$n.log
done
