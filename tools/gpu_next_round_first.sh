#!/usr/bin/env bash
# First GPU call of the next round (about 12 GPU-minutes; every leg is time-boxed and writes its own log under gpurun_out/, so a
# cut-off call still leaves the earlier results).  Order = value per GPU-minute.   bash tools/gpu_next_round_first.sh <tag>
#   1. the round's baseline on THIS box: driver bench line, rocprofv3 kernel stats, PMC traffic (tools/gpu_profile.sh)
#   2. same-box A/B of every knob the product still carries (all default on): GroupNorm statistics from the producer epilogue,
#      LayerNorm fold, two streams; config 3: softmax groups of the cross-attention; the odd-unit lane on the 8-rank unit lists
#   3. micro-benchmarks whose numbers DESIGN.md quotes: attention variants, norm / temporal-attention launches
#   4. re-tune of the tile table on the current epilogue (column statistics changed it) -> gpurun_out/<tag>_*_gemm_tune.json;
#      merge with `python tools/gpu_gemm_tune.py --merge musev_amd/csrc/gemm_tuned.h <json...>` and A/B the whole step before keeping it
set -u
TAG=${1:-r04a}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
bash tools/gpu_profile.sh $TAG > $OUT/${TAG}_profile_stdout.log 2>&1
cd $ROOT
cut -c1-400 $OUT/${TAG}_bench.json
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-config4"
for tag in all_on "nocs:MUSEV_COLSTATS=0" "nofold:MUSEV_LN_FOLD=0" "one_stream:MUSEV_HALF_STREAMS=0" all_on2; do
  name=${tag%%:*}; envs=""; [ "$tag" != "$name" ] && envs=${tag#*:}
  ( env $envs timeout 300 $B 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('config2 $name', d['ms_per_step'], d['value'])" ) >> $OUT/${TAG}_knobs_ab.log 2>&1
done
for tag in groups "separate:MUSEV_ATTN_GROUPS=0"; do
  name=${tag%%:*}; envs=""; [ "$tag" != "$name" ] && envs=${tag#*:}
  ( env $envs timeout 300 python bench.py --workload config3 --steps 12 --warmup 3 --no-cpu-baseline --no-roofline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('config3 $name', d['ms_per_step'], d['value'])" ) >> $OUT/${TAG}_knobs_ab.log 2>&1
done
cat $OUT/${TAG}_knobs_ab.log
( timeout 300 python tools/gpu_odd_unit_lane.py 2>&1 | grep "rank " ) > $OUT/${TAG}_odd_unit_lane.log; cat $OUT/${TAG}_odd_unit_lane.log
( timeout 300 python tools/gpu_norm_tattn_bench.py 2>&1 | tail -24 ) > $OUT/${TAG}_norm_bench.log; tail -8 $OUT/${TAG}_norm_bench.log
( timeout 420 python tools/gpu_gemm_tune.py ${TAG}_musev512 2>&1 | tail -12 ) > $OUT/${TAG}_tune_musev512.log; tail -4 $OUT/${TAG}_tune_musev512.log
