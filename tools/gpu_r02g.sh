#!/usr/bin/env bash
# Round 2, GPU call 7: PMC counters of the attention kernel (SQ issue / wait / pipe busy), timing after the cross-lane-free common path
set -u
TAG=${1:-r02g}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
( timeout 300 python tools/gpu_attn_bench.py 2>&1 | grep -E "attn_|FAIL" ) > $OUT/${TAG}_attn_bench.log; cat $OUT/${TAG}_attn_bench.log
cd /tmp
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM"; do
  n=$(echo $set | cut -d' ' -f1)
  ( timeout 200 rocprofv3 --pmc $set --output-format csv -d $OUT/${TAG}_pmc_$n -o pmc -- python $ROOT/tools/gpu_attn_pmc.py 2>&1 | tail -2 ) > $OUT/${TAG}_pmc_$n.log
done
cd $ROOT
python - <<'PY'
import csv, glob, collections, os
out=os.environ.get("GRAFT_REPO_ROOT", os.getcwd())+"/gpurun_out"
for f in glob.glob(out+"/r02g_pmc_*/**/*counter_collection.csv", recursive=True):
    agg=collections.defaultdict(lambda:[0,0.0])
    for row in csv.DictReader(open(f)):
        if "attn3" in row.get("Kernel_Name",""):
            a=agg[row["Counter_Name"]]; a[0]+=1; a[1]+=float(row["Counter_Value"])
    for k,(n,v) in sorted(agg.items()): print(f"{k:28s} per launch {v/n:16.0f}")
PY
