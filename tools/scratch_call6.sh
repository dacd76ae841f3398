cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 700 python tools/gpu_gemm_tune.py r04t_musev512 2>&1 | grep -v amdgpu.ids | tail -40 ) > gpurun_out/r04t_tune_musev512.log; tail -30 gpurun_out/r04t_tune_musev512.log
