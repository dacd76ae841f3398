#!/usr/bin/env bash
# GEMM variant A/B (parity + microbench), then tests / smoke / bench / rocprof with the best passing variant.
set -u
TAG=${1:-r01b}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
( timeout 900 python tools/gpu_gemm_ab.py $TAG ${VARIANTS:-2 10 11} 2>&1 | tail -80 ) > $OUT/${TAG}_gemm_ab.log
BEST=$(python - <<PY
import json
try:
    r=json.load(open("$OUT/${TAG}_gemm_ab.json"))["variants"]
    best,bt=2,None
    for v,rep in r.items():
        if not rep["all_ok"]: continue
        t=sum(b.get("ms",1e9) for b in rep["bench"].values())
        if bt is None or t<bt: best,bt=int(v),t
    print(best)
except Exception: print(1)
PY
)
BESTA=$(python -c "import json; print(json.load(open('$OUT/${TAG}_gemm_ab.json')).get('best_attn', 1))" 2>/dev/null || echo 1)
echo "best variant: gemm $BEST attn $BESTA" | tee $OUT/${TAG}_best.log
export MUSEV_GEMM_VARIANT=$BEST
export MUSEV_ATTN_VARIANT=$BESTA
( timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > $OUT/${TAG}_pytest_gpu.log
( timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5 ) > $OUT/${TAG}_smoke.log
( timeout 900 python bench.py --steps 6 --warmup 2 2>&1 | tail -2 ) > $OUT/${TAG}_bench.log
if [ "${NOPROF:-0}" = "1" ]; then tail -40 $OUT/${TAG}_gemm_ab.log; tail -3 $OUT/${TAG}_pytest_gpu.log; tail -2 $OUT/${TAG}_smoke.log; tail -1 $OUT/${TAG}_bench.log | cut -c1-1500; exit 0; fi
cd /tmp
( timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof -o bench -- python $ROOT/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-roofline 2>&1 | tail -3 ) > $OUT/${TAG}_rocprof.log
cd $ROOT
find $OUT/${TAG}_prof -name "*kernel_trace.csv" -size +30M -delete
tail -40 $OUT/${TAG}_gemm_ab.log
tail -3 $OUT/${TAG}_pytest_gpu.log; tail -2 $OUT/${TAG}_smoke.log; tail -1 $OUT/${TAG}_bench.log | cut -c1-1500
