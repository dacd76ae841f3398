#!/usr/bin/env python3
"""Per-problem roofline of the step's matrix launches, from a committed bench.py --gemm-by-problem file (no GPU needed):

    python tools/gemm_roofline_by_problem.py [profiles/r04z_gemm_by_problem.json] [--out profiles/<tag>_gemm_roofline_by_problem.json]

Every distinct batch-1 problem of a config-2 step was timed alone (bench.py --gemm-by-problem: HIP events around back-to-back launches
on its real operands).  For each one: the time the MFMA peak allows (2 M N K / 2.5 PFLOP/s), the time HBM allows (algorithmic bytes
-- every operand read once, the output written once -- / 8 TB/s, and / 6.3 TB/s "achievable", MI355X_MICROARCH.md), the larger of
the two = the problem's roofline, and measured / roofline.  The step-level `roofline.frac` of the bench line divides ALL the family's
FLOPs by the MFMA peak; this table says how much of the family's time sits in launches that the memory system, not the matrix
pipe, bounds -- and what the family could reach at best."""
from __future__ import annotations

import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PEAK_TFLOPS, HBM_TBPS, HBM_ACHIEVABLE_TBPS = 2500.0, 8.0, 6.3


def main():
    argv = sys.argv[1:]
    out = None
    if "--out" in argv:
        i = argv.index("--out")
        out = argv[i + 1]
        del argv[i:i + 2]
    path = argv[0] if argv else os.path.join(ROOT, "profiles", "r04z_gemm_by_problem.json")
    rows = []
    for p in json.load(open(path)):
        us = p["us"]
        flops = p["tflops"] * 1e12 * us * 1e-6
        bytes_ = p["algorithmic_GBps"] * 1e9 * us * 1e-6
        t_mfma = flops / (PEAK_TFLOPS * 1e12) * 1e6
        t_hbm = bytes_ / (HBM_TBPS * 1e12) * 1e6
        t_hbm_a = bytes_ / (HBM_ACHIEVABLE_TBPS * 1e12) * 1e6
        bound = "hbm" if t_hbm_a > t_mfma else "mfma"
        roof = max(t_mfma, t_hbm_a)
        rows.append(dict(p, mfma_us=t_mfma, hbm_us=t_hbm, hbm_achievable_us=t_hbm_a, bound=bound, roofline_us=roof, frac_of_roofline=roof / us,
                         flop_per_byte=flops / bytes_))
    rows.sort(key=lambda r: -r["ms_per_step"])
    tot = sum(r["ms_per_step"] for r in rows)
    tot_roof = sum(r["roofline_us"] * r["launches_per_step"] for r in rows) / 1e3
    tot_mfma = sum(r["mfma_us"] * r["launches_per_step"] for r in rows) / 1e3
    hbm_rows = [r for r in rows if r["bound"] == "hbm"]
    t_hbm_bound = sum(r["ms_per_step"] for r in hbm_rows)
    print(f"{os.path.basename(path)}: {len(rows)} problems, {sum(r['launches_per_step'] for r in rows)} launches, {tot:.2f} ms per step alone on the GPU")
    print(f"{'mode':10s} {'M':>7s} {'N':>6s} {'K':>6s} {'epilogue':16s} {'x/step':>6s} {'us':>7s} {'TFLOP/s':>8s} {'F/B':>6s} {'mfma us':>8s} {'hbm us':>7s} bound  of-roofline  ms/step")
    for r in rows:
        print(f"{r['mode']:10s} {r['M']:7d} {r['N']:6d} {r['K']:6d} {r['epilogue'][:16]:16s} {r['launches_per_step']:6d} {r['us']:7.1f} {r['tflops']:8.0f} "
              f"{r['flop_per_byte']:6.0f} {r['mfma_us']:8.1f} {r['hbm_achievable_us']:7.1f} {r['bound']:5s}  {r['frac_of_roofline']:10.2f}  {r['ms_per_step']:7.3f}")
    print(f"family alone: {tot:.2f} ms per step; at the MFMA peak {tot_mfma:.2f} ms ({tot_mfma / tot:.3f}); at each problem's own roofline "
          f"(max of MFMA peak and {HBM_ACHIEVABLE_TBPS} TB/s) {tot_roof:.2f} ms ({tot_roof / tot:.3f})")
    print(f"HBM-bound problems (ridge at {PEAK_TFLOPS / HBM_ACHIEVABLE_TBPS:.0f} FLOP/B): {len(hbm_rows)} of {len(rows)}, {t_hbm_bound:.2f} ms = {t_hbm_bound / tot:.1%} of the family's time, "
          f"running at {sum(r['roofline_us'] * r['launches_per_step'] for r in hbm_rows) / 1e3 / max(t_hbm_bound, 1e-9):.2f} of their (memory) roofline; "
          f"MFMA-bound problems at {sum(r['roofline_us'] * r['launches_per_step'] for r in rows if r['bound'] == 'mfma') / 1e3 / max(tot - t_hbm_bound, 1e-9):.2f} of theirs")
    if out:
        with open(out, "w") as f:
            json.dump(dict(source=os.path.basename(path), peak_tflops=PEAK_TFLOPS, hbm_tbps=HBM_TBPS, hbm_achievable_tbps=HBM_ACHIEVABLE_TBPS,
                           family_ms_per_step=tot, at_mfma_peak_ms=tot_mfma, at_own_roofline_ms=tot_roof, hbm_bound_share_of_time=t_hbm_bound / tot, problems=rows), f, indent=1)


if __name__ == "__main__":
    main()
