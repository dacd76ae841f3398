#!/usr/bin/env bash
# First GPU call of the next round (~8 GPU-minutes; every leg is time-boxed and writes its own log under gpurun_out/, so a cut-off call
# still leaves the earlier results).   bash tools/gpu_next_round_first.sh <tag>
#   0. (2 minutes) the kernel forms written at the end of round 4 with NO GPU minutes left -- the resident-K/V cross-attention, the
#      GroupNorm fold inside the apply pass, the weight-stationary GEMM order -- run their parity cases on hardware for the first
#      time (MUSEV_TEST_UNPROVEN=1); the at-size loop goldens that joined the suite unseen by a GPU (config 3, 20 steps) and the
#      refer_self_attn_emb "write" case
#   1. (1 minute) tools/gpu_xattn_bench.py: the resident cross-attention against the tiled kernel (+ the rows-per-block sweep), the
#      GroupNorm fold inside the apply pass against fold launch + apply
#   2. (4 minutes) same-box A/B of the three new switches in the whole step, config 2 and config 3 (all default OFF: switch on what
#      wins, in musev_amd/ops.py, then re-run tools/gpu_validate.sh)
#   MUSEV_FIRST_FULL=1 adds the round-4 legs: tools/gpu_final_profile.sh (driver bench line, rocprofv3 kernel stats, PMC traffic,
#   per-problem table, SQ counters; ~8 minutes), the A/B of the round-4 switches, tools/gpu_ffn_bench.py
set -u
TAG=${1:-r05a}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
( MUSEV_TEST_UNPROVEN=1 timeout 400 python -m pytest tests/test_kernels_gpu.py -q -k "resident or weight_stationary or colstats" 2>&1 | grep -v amdgpu.ids | tail -15 ) > $OUT/${TAG}_unproven_kernels.log
cat $OUT/${TAG}_unproven_kernels.log
( timeout 500 python -m pytest tests/test_pipeline_gpu.py tests/test_model_gpu.py -q -s -k "refnet_cfg3_loop20 or refer_self_write" 2>&1 | grep -E "free-running|written|passed|failed|Error" | cut -c1-600 ) > $OUT/${TAG}_new_gpu_cases.log
cat $OUT/${TAG}_new_gpu_cases.log
( timeout 200 python tools/gpu_xattn_bench.py 2>&1 | grep -v amdgpu.ids ) > $OUT/${TAG}_xattn_bench.log; cat $OUT/${TAG}_xattn_bench.log
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-config4"
ab() {  # ab <log> <bench args> <legs...>
  local log=$1 args=$2; shift 2
  for tag in "$@"; do
    name=${tag%%:*}; envs=""; [ "$tag" != "$name" ] && envs=${tag#*:}
    ( env $envs timeout 300 $B $args 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$name', d['ms_per_step'], d['value'])" ) >> $log 2>&1
  done
  cat $log
}
NEW=("base" "xattn_resident:MUSEV_XATTN_RESIDENT=1" "gn_fold_in_apply:MUSEV_GN_FOLD_IN_APPLY=1" "gemm_weight_stationary:MUSEV_GEMM_WEIGHT_STATIONARY=1"
     "all_three:MUSEV_XATTN_RESIDENT=1 MUSEV_GN_FOLD_IN_APPLY=1 MUSEV_GEMM_WEIGHT_STATIONARY=1" "base2")
ab $OUT/${TAG}_new_knobs_config2.log "" "${NEW[@]}"
ab $OUT/${TAG}_new_knobs_config3.log "--workload config3" "base" "xattn_resident:MUSEV_XATTN_RESIDENT=1" "all_three:MUSEV_XATTN_RESIDENT=1 MUSEV_GN_FOLD_IN_APPLY=1 MUSEV_GEMM_WEIGHT_STATIONARY=1" "base2"
if [ "${MUSEV_FIRST_FULL:-0}" = "1" ]; then
  bash tools/gpu_final_profile.sh $TAG > $OUT/${TAG}_final_profile_stdout.log 2>&1
  cut -c1-400 $OUT/${TAG}_bench.json
  ab $OUT/${TAG}_knobs_ab.log "" all_on "no_carry:MUSEV_CARRY=0" "no_shared_front:MUSEV_SHARE_PREFIX=0" "no_fused_ffn:MUSEV_FFN_FUSED=0" "no_colstats:MUSEV_COLSTATS=0" \
     "no_ln_fold:MUSEV_LN_FOLD=0" "one_stream:MUSEV_HALF_STREAMS=0" all_on2
  ( timeout 200 python tools/gpu_ffn_bench.py 2>&1 | grep -v amdgpu.ids ) > $OUT/${TAG}_ffn_bench.log; cat $OUT/${TAG}_ffn_bench.log
  [ -f musev_amd/csrc/libmusev_hip_exp.so ] && ( timeout 200 python tools/gpu_ffn_bench.py --ablate 2>&1 | grep -v amdgpu.ids ) > $OUT/${TAG}_ffn_ablate.log
fi
