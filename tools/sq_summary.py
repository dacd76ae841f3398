"""Folds the per-dispatch SQ counter rows of tools/gpu_sq_counters.sh into one table per kernel (template instantiation).

    python tools/sq_summary.py <tag>        reads gpurun_out/<tag>_sq_{1,2}/**/counter_collection.csv, writes gpurun_out/<tag>_sq_counters.json

Units (MI355X_MICROARCH.md, "rocprofv3 PMC slots" / constants table): SQ_WAVE_CYCLES, SQ_WAIT_*, SQ_ACTIVE_INST_* count quad-cycles
summed over waves; SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over SIMDs; GRBM_GUI_ACTIVE = GPU-busy cycles of the dispatch as
rocprofv3 reports it on this part: SUMMED OVER THE 8 XCDs (a 749 us launch reads 7.3 M at ~1.9 GHz under the profiler), hence / 8.
Derived per kernel, over all its dispatches of the step:
  mfma_busy      = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024 SIMDs)        share of the chip's matrix-pipe cycles in use
  valu_pipe      = SQ_ACTIVE_INST_VALU * 4 / (GRBM_GUI_ACTIVE / 8 * 1024 SIMDs)          share of the SIMDs' cycles with a VALU (incl. MFMA)
                                                                                         instruction issuing (quad-cycles -> cycles)
  lds_pipe       = SQ_LDS_IDX_ACTIVE / (GRBM_GUI_ACTIVE / 8 * 256 CUs)                   share of the CUs' LDS-array cycles in use
  valu_active    = SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES                                  share of wave time issuing VALU (incl. MFMA issue)
  lds_active     = SQ_ACTIVE_INST_LDS / SQ_WAVE_CYCLES
  wait_any       = SQ_WAIT_ANY / SQ_WAVE_CYCLES          parked on s_waitcnt / barrier
  wait_inst      = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES     issue stalls (pipe busy / dependency);  wait_inst_lds its LDS sub-bucket
  lds_conflict   = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE                               share of LDS-array cycles lost to bank conflicts
  valu_per_mfma  = SQ_INSTS_VALU / SQ_INSTS_MFMA,  lds_per_mfma = SQ_INSTS_LDS / SQ_INSTS_MFMA   (instruction mix, per wave instruction)"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N_SIMD = 256 * 4
N_XCC = 8


def short(name: str) -> str:
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    m = re.match(r"([A-Za-z0-9_]+(?:<[^>]*>)?)", name)
    return m.group(1) if m else name[:60]


def fold(tag: str) -> dict:
    agg = defaultdict(lambda: defaultdict(float))
    calls = defaultdict(lambda: defaultdict(int))
    for i in (1, 2):
        for f in glob.glob(os.path.join(ROOT, "gpurun_out", f"{tag}_sq_{i}", "**", "*counter_collection.csv"), recursive=True):
            with open(f) as fh:
                for row in csv.DictReader(fh):
                    k = short(row.get("Kernel_Name", ""))
                    c = row.get("Counter_Name", "")
                    if c == "GRBM_GUI_ACTIVE":
                        c = f"GRBM_GUI_ACTIVE_{i}"
                    agg[k][c] += float(row.get("Counter_Value", 0.0))
                    calls[k][c] += 1
    return derive({k: dict(v) for k, v in agg.items()}, {k: dict(c) for k, c in calls.items()})


def derive(agg: dict, calls: dict) -> dict:
    out = {}
    for k, v in agg.items():
        g = lambda n: v.get(n, 0.0)  # noqa: E731
        wc = max(g("SQ_WAVE_CYCLES"), 1.0)
        mf = max(g("SQ_INSTS_MFMA"), 1.0)
        e = {"dispatches": calls[k].get("SQ_WAVE_CYCLES", 0) or calls[k].get("SQ_INSTS_VALU", 0),
             "gui_active_cycles": (g("GRBM_GUI_ACTIVE_2") or g("GRBM_GUI_ACTIVE_1")) / N_XCC,
             "mfma_busy": g("SQ_VALU_MFMA_BUSY_CYCLES") / max(g("GRBM_GUI_ACTIVE_2") / N_XCC * N_SIMD, 1.0),
             "valu_pipe": g("SQ_ACTIVE_INST_VALU") * 4.0 / max(g("GRBM_GUI_ACTIVE_1") / N_XCC * N_SIMD, 1.0),
             "lds_pipe": g("SQ_LDS_IDX_ACTIVE") / max(g("GRBM_GUI_ACTIVE_2") / N_XCC * 256, 1.0),
             "valu_active": g("SQ_ACTIVE_INST_VALU") / wc, "lds_active": g("SQ_ACTIVE_INST_LDS") / wc,
             "inst_active": g("SQ_ACTIVE_INST_ANY") / wc, "wait_any": g("SQ_WAIT_ANY") / wc, "wait_inst": g("SQ_WAIT_INST_ANY") / wc,
             "wait_inst_lds": g("SQ_WAIT_INST_LDS") / wc,
             "lds_conflict": g("SQ_LDS_BANK_CONFLICT") / max(g("SQ_LDS_IDX_ACTIVE"), 1.0),
             "valu_per_mfma": g("SQ_INSTS_VALU") / mf if g("SQ_INSTS_MFMA") else None,
             "lds_per_mfma": g("SQ_INSTS_LDS") / mf if g("SQ_INSTS_MFMA") else None,
             "raw": {n: x for n, x in v.items()}}
        out[k] = e
    return out


def main():
    tag = sys.argv[1]
    path = os.path.join(ROOT, "gpurun_out", f"{tag}_sq_counters.json")
    if len(sys.argv) > 2 and sys.argv[2] == "--from-json":  # re-derive from the raw sums of an existing summary (the csv rows are not kept)
        old = json.load(open(path))["kernels"]
        res = derive({k: e["raw"] for k, e in old.items()}, {k: {"SQ_WAVE_CYCLES": e["dispatches"]} for k, e in old.items()})
    else:
        res = fold(tag)
    order = sorted(res, key=lambda k: -res[k]["gui_active_cycles"])
    print(f"{'kernel':44s} {'n':>5s} {'Mcyc':>8s} {'mfma':>6s} {'vpipe':>6s} {'lpipe':>6s} {'valu':>6s} {'lds':>6s} {'wait':>6s} {'stall':>6s} {'st_lds':>6s} {'confl':>6s} {'V/M':>6s} {'L/M':>6s}")
    for k in order[:40]:
        e = res[k]
        f = lambda x: "   -  " if x is None else f"{x:6.3f}"  # noqa: E731
        print(f"{k[:44]:44s} {e['dispatches']:5d} {e['gui_active_cycles'] / 1e6:8.2f} {f(e['mfma_busy'])} {f(e['valu_pipe'])} {f(e['lds_pipe'])} {f(e['valu_active'])} {f(e['lds_active'])} "
              f"{f(e['wait_any'])} {f(e['wait_inst'])} {f(e['wait_inst_lds'])} {f(e['lds_conflict'])} {f(e['valu_per_mfma'])} {f(e['lds_per_mfma'])}")
    sys.path.insert(0, ROOT)
    import bench
    with open(path, "w") as fh:
        json.dump({"kernel_source_hash": bench.kernel_source_hash(), "units": __doc__, "kernels": res}, fh, indent=1)


if __name__ == "__main__":
    main()
