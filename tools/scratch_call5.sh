cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 200 python tools/gpu_ffn_bench.py --ablate 2>&1 | grep -v amdgpu.ids ) > gpurun_out/r04f_ffn_ablate.log; cat gpurun_out/r04f_ffn_ablate.log
