cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
T=r04d
( timeout 400 python -m pytest tests/test_kernels_gpu.py -q -x -k "ffn or tail_carry or carry or window_loop or conv_in_out or groupnorm or gemm_l0" 2>&1 | tail -5 ) > gpurun_out/${T}_pytest_kernels.log; cat gpurun_out/${T}_pytest_kernels.log
( timeout 200 python tools/gpu_ffn_bench.py 2>&1 | grep -v amdgpu.ids ) > gpurun_out/${T}_ffn_bench.log; cat gpurun_out/${T}_ffn_bench.log
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-config4"
for tag in ffn_on "ffn_off:MUSEV_FFN_FUSED=0" ffn_on2 "ffn_off2:MUSEV_FFN_FUSED=0"; do
  name=${tag%%:*}; envs=""; [ "$tag" != "$name" ] && envs=${tag#*:}
  ( env $envs timeout 300 $B 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('config2 $name', d['ms_per_step'], d['value'])" ) >> gpurun_out/${T}_ffn_ab.log 2>&1
done
cat gpurun_out/${T}_ffn_ab.log
( timeout 600 python -m pytest tests/test_pipeline_gpu.py -q -x -s -k "loop20" 2>&1 | grep -v amdgpu.ids | grep -E "free-running|from the reference|passed|failed|Error" | cut -c1-900 ) > gpurun_out/${T}_pytest_loop20.log; cat gpurun_out/${T}_pytest_loop20.log
