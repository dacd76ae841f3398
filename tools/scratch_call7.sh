cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
T=r04t
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-config4"
for tag in table_r03 "table_r04t:MUSEV_HIP_LIBRARY=$GRAFT_REPO_ROOT/musev_amd/csrc/libmusev_hip_tune.so" table_r03_2 "table_r04t_2:MUSEV_HIP_LIBRARY=$GRAFT_REPO_ROOT/musev_amd/csrc/libmusev_hip_tune.so"; do
  name=${tag%%:*}; envs=""; [ "$tag" != "$name" ] && envs=${tag#*:}
  ( env $envs timeout 300 $B 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('config2 $name', d['ms_per_step'], d['value'])" ) >> gpurun_out/${T}_table_ab.log 2>&1
done
cat gpurun_out/${T}_table_ab.log
