cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
T=r04c
( timeout 300 python -m pytest tests/test_kernels_gpu.py -q -x -k "carry or gemm_epilogue or conv3x3 or tconv3 or colstats_conv_groupnorm" 2>&1 | tail -5 ) > gpurun_out/${T}_pytest_carry.log; cat gpurun_out/${T}_pytest_carry.log
( timeout 500 python tools/gpu_error_attribution.py --only carry --out gpurun_out/${T}_attribution.json 2>&1 | grep -v amdgpu.ids ) > gpurun_out/${T}_attribution.log; cat gpurun_out/${T}_attribution.log
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-config4"
for tag in carry_on "carry_off:MUSEV_CARRY=0" carry_on2 "carry_off2:MUSEV_CARRY=0"; do
  name=${tag%%:*}; envs=""; [ "$tag" != "$name" ] && envs=${tag#*:}
  ( env $envs timeout 300 $B 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('config2 $name', d['ms_per_step'], d['value'])" ) >> gpurun_out/${T}_carry_ab.log 2>&1
done
cat gpurun_out/${T}_carry_ab.log
( timeout 900 python -m pytest tests/test_pipeline_gpu.py -q -x -s -k "twenty_step or at_size" 2>&1 | grep -v amdgpu.ids | tail -60 ) > gpurun_out/${T}_pytest_loop.log; tail -40 gpurun_out/${T}_pytest_loop.log
for tag in share_on "share_off:MUSEV_SHARE_PREFIX=0" share_on2 "share_off2:MUSEV_SHARE_PREFIX=0"; do
  name=${tag%%:*}; envs=""; [ "$tag" != "$name" ] && envs=${tag#*:}
  ( env $envs timeout 300 $B 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('config2 $name', d['ms_per_step'], d['value'])" ) >> gpurun_out/${T}_share_ab.log 2>&1
done
cat gpurun_out/${T}_share_ab.log
