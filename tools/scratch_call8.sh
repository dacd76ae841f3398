cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
( timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --rehearse-shared-gpu --steps 3 --warmup 1 --no-cpu-baseline --no-roofline 2>&1 | grep -v amdgpu.ids | tail -3 ) > gpurun_out/r04x_bench_rehearsal_n2_shared_gpu.log; cut -c1-3000 gpurun_out/r04x_bench_rehearsal_n2_shared_gpu.log
( timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-config4 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('config2', d['ms_per_step'], d['value'], 'traffic', r['traffic'], r['traffic_ratio'], 'frac', r['frac'])" ) > gpurun_out/r04x_bench_with_traffic.log; cat gpurun_out/r04x_bench_with_traffic.log
