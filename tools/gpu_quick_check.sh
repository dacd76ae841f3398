#!/usr/bin/env bash
# A short GPU check after a host-side change (through gpurun, ~2 GPU-minutes): carry / statistics kernel cases, the small-net loops
# (graph replay bit-identity, 20-step drift, odd-unit lane), smoke().  MUSEV_QUICK_AT_SIZE=1 adds the at-size loop goldens (+3 min).
cd $GRAFT_REPO_ROOT
( timeout 300 python -m pytest tests/test_kernels_gpu.py -q -x -k "carry or colstats" 2>&1 | tail -3 )
K="twenty or first_steps or odd_unit or uniform_v2"
[ "${MUSEV_QUICK_AT_SIZE:-0}" = "1" ] && K="$K or at_size"
( timeout 500 python -m pytest tests/test_pipeline_gpu.py -q -x -s -k "$K" 2>&1 | grep -E "free-running|passed|failed|Error" | cut -c1-400 )
( timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 )
