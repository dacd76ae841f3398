#!/usr/bin/env bash
# Same-box back-to-back A/B of environment switches in the whole denoise step (fresh boxes differ by several per cent on one binary:
# only legs of ONE call compare).   bash tools/gpu_ab.sh <log> "<extra bench.py args>" <name[:ENV=V ENV2=V ...]> ...
#   e.g.  bash tools/gpu_ab.sh gpurun_out/r05f_ab.log "" base "ws0:MUSEV_OPS=GEMM_WEIGHT_STATIONARY=0" base2
# every leg: python bench.py --steps 20 --warmup 5 (the driver's step count), no CPU baseline / roofline / config-4 legs
LOG=$1; ARGS=$2; shift 2
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-config4"
for tag in "$@"; do
  name=${tag%%:*}; envs=""; [ "$tag" != "$name" ] && envs=${tag#*:}
  ( env $envs timeout 300 $B $ARGS 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$name', round(d['ms_per_step'], 3), round(d['value'], 3))" ) >> $LOG 2>&1
done
cat $LOG
