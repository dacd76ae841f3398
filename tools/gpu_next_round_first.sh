#!/usr/bin/env bash
# First GPU call of the next round (~8 GPU-minutes; every leg is time-boxed and writes its own log under gpurun_out/, so a cut-off call
# still leaves the earlier results).   bash tools/gpu_next_round_first.sh <tag>
#   1. the round's baseline on THIS box: tools/gpu_final_profile.sh (driver bench line, rocprofv3 kernel stats, PMC traffic, per-problem
#      table, config 3 / 5 lines with the roofline block, SQ counters)
#   2. same-box A/B of every knob the product carries (all default on): two-fp16 carry, shared CFG front, fused level-0 feed-forward,
#      producer column statistics, LayerNorm fold, two streams
#   3. the fused feed-forward alone (tools/gpu_ffn_bench.py) and its ablation through the experiment build, when that library is there
#   0. (first: 2 minutes) the kernel forms written at the end of round 4 with NO GPU minutes left -- the resident-K/V cross-attention,
#      the GroupNorm fold inside the apply pass, the weight-stationary GEMM order: their parity cases on hardware
#      (MUSEV_TEST_UNPROVEN=1), then each as a same-box A/B leg of step 2 (all default OFF: switch on what wins, in ops.py)
set -u
TAG=${1:-r05a}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
( MUSEV_TEST_UNPROVEN=1 timeout 400 python -m pytest tests/test_kernels_gpu.py -q -k "resident or weight_stationary or colstats" 2>&1 | grep -v amdgpu.ids | tail -15 ) > $OUT/${TAG}_unproven_kernels.log
cat $OUT/${TAG}_unproven_kernels.log
( timeout 200 python tools/gpu_xattn_bench.py 2>&1 | grep -v amdgpu.ids ) > $OUT/${TAG}_xattn_bench.log; cat $OUT/${TAG}_xattn_bench.log
bash tools/gpu_final_profile.sh $TAG > $OUT/${TAG}_final_profile_stdout.log 2>&1
cut -c1-400 $OUT/${TAG}_bench.json
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-config4"
for tag in all_on "no_carry:MUSEV_CARRY=0" "no_shared_front:MUSEV_SHARE_PREFIX=0" "no_fused_ffn:MUSEV_FFN_FUSED=0" "no_colstats:MUSEV_COLSTATS=0" \
           "no_ln_fold:MUSEV_LN_FOLD=0" "one_stream:MUSEV_HALF_STREAMS=0" all_on2 \
           "xattn_resident:MUSEV_XATTN_RESIDENT=1" "gn_fold_in_apply:MUSEV_GN_FOLD_IN_APPLY=1" "gemm_weight_stationary:MUSEV_GEMM_WEIGHT_STATIONARY=1" \
           "all_three:MUSEV_XATTN_RESIDENT=1 MUSEV_GN_FOLD_IN_APPLY=1 MUSEV_GEMM_WEIGHT_STATIONARY=1" all_on3; do
  name=${tag%%:*}; envs=""; [ "$tag" != "$name" ] && envs=${tag#*:}
  ( env $envs timeout 300 $B 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('config2 $name', d['ms_per_step'], d['value'])" ) >> $OUT/${TAG}_knobs_ab.log 2>&1
done
cat $OUT/${TAG}_knobs_ab.log
( timeout 200 python tools/gpu_ffn_bench.py 2>&1 | grep -v amdgpu.ids ) > $OUT/${TAG}_ffn_bench.log; cat $OUT/${TAG}_ffn_bench.log
[ -f musev_amd/csrc/libmusev_hip_exp.so ] && ( timeout 200 python tools/gpu_ffn_bench.py --ablate 2>&1 | grep -v amdgpu.ids ) > $OUT/${TAG}_ffn_ablate.log
