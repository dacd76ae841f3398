cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
T=r04e
( timeout 300 python -m pytest tests/test_kernels_gpu.py -q -x -k "ffn" 2>&1 | tail -3 ) > gpurun_out/${T}_pytest_ffn.log; cat gpurun_out/${T}_pytest_ffn.log
( timeout 200 python tools/gpu_ffn_bench.py 2>&1 | grep -v amdgpu.ids ) > gpurun_out/${T}_ffn_bench.log; cat gpurun_out/${T}_ffn_bench.log
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-config4"
for tag in rot_on "rot_off:MUSEV_FFN_ROTATE=0" rot_on2 "rot_off2:MUSEV_FFN_ROTATE=0" "ffn_off:MUSEV_FFN_FUSED=0"; do
  name=${tag%%:*}; envs=""; [ "$tag" != "$name" ] && envs=${tag#*:}
  ( env $envs timeout 300 $B 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('config2 $name', d['ms_per_step'], d['value'])" ) >> gpurun_out/${T}_ffn_ab.log 2>&1
done
cat gpurun_out/${T}_ffn_ab.log
