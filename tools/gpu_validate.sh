#!/usr/bin/env bash
# Every GPU test + smoke() on the tree (through gpurun, ~10 GPU-minutes):  bash tools/gpu_validate.sh   -> gpurun_out/r04y_*
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 1500 python -m pytest tests -m gpu -q 2>&1 | grep -v amdgpu.ids | tail -25 ) > gpurun_out/r04y_pytest_gpu.log; cat gpurun_out/r04y_pytest_gpu.log
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -3 ) > gpurun_out/r04y_smoke.log; cat gpurun_out/r04y_smoke.log
( timeout 300 python __graft_entry__.py smoke 2>&1 | grep -v amdgpu.ids | tail -3 ) > gpurun_out/r04y_build_then_smoke.log; cat gpurun_out/r04y_build_then_smoke.log
