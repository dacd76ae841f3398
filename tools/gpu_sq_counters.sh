#!/usr/bin/env bash
# SQ counter passes over ONE eager step of the benchmark (every kernel of the step, per-dispatch rows):
#     bash tools/gpu_sq_counters.sh <tag>            (through gpurun; ~2 min per pass)
# Two passes of 8 SQ counters (+ GRBM_GUI_ACTIVE, an independent block), counters only -- no trace domain next to --pmc.
#   pass A  where the waves' time goes: SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS
#           SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS        (quad-cycle units, MI355X_MICROARCH.md "rocprofv3 PMC slots")
#   pass B  what was issued: SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_VALU_MFMA_BUSY_CYCLES (cycles)
#           SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
# tools/sq_summary.py folds the rows per kernel (template instantiation = tile configuration) into gpurun_out/<tag>_sq_counters.json:
# MFMA-busy share of the SIMD-cycles of the launch, VALU-active share of the wave cycles, LDS bank-conflict share, wait shares.
set -u
TAG=${1:-r04a}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
A="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE"
B="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"
i=0
for set in "$A" "$B"; do
  i=$((i+1))
  ( MUSEV_NO_GRAPH=1 MUSEV_HALF_STREAMS=${SQ_HALF_STREAMS:-1} timeout 600 rocprofv3 --pmc $set --output-format csv -d $OUT/${TAG}_sq_$i -o pmc -- \
      python $ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-config4 2>&1 | tail -2 ) > $OUT/${TAG}_sq_$i.log
done
cd $ROOT
python tools/sq_summary.py $TAG > $OUT/${TAG}_sq_counters.log 2>&1
find $OUT/${TAG}_sq_1 $OUT/${TAG}_sq_2 -name "*counter_collection.csv" -delete 2>/dev/null
tail -40 $OUT/${TAG}_sq_counters.log
