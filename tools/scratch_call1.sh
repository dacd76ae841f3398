cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 420 python tools/gpu_error_attribution.py --only carry --no-hip --out gpurun_out/r04b_attribution_carry.json 2>&1 | grep -v amdgpu.ids ) > gpurun_out/r04b_attribution_carry.log
cat gpurun_out/r04b_attribution_carry.log
( timeout 420 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-config4 --gemm-by-problem gpurun_out/r04b_gemm_by_problem.json 2>&1 | tail -1 ) > gpurun_out/r04b_bench.json
cut -c1-300 gpurun_out/r04b_bench.json
