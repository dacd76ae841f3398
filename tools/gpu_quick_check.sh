cd $GRAFT_REPO_ROOT
( timeout 300 python -m pytest tests/test_kernels_gpu.py -q -x -k "carry or colstats" 2>&1 | tail -3 )
( timeout 500 python -m pytest tests/test_pipeline_gpu.py -q -x -s -k "at_size or twenty" 2>&1 | grep -E "free-running|passed|failed|Error" | cut -c1-400 )
( timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 )
