#!/usr/bin/env python3
"""Per-family busy time of the denoise step from a rocprofv3 kernel trace WITH timestamps.

    rocprofv3 --kernel-trace --output-format csv -d <dir> -o trace -- python bench.py --steps K --warmup W --no-cpu-baseline --no-roofline --no-config4
    python tools/trace_overlap.py <dir or *_kernel_trace.csv> --steps K [--family-tflop F] [--json out.json]

The step runs the two CFG halves on two HIP streams, so kernels overlap and "sum of durations" says nothing about the step.  From
the begin / end timestamps of every dispatch this tool computes, per denoise step (a step ends with its `cfg_ddim_step_kernel`
dispatch; the LAST K steps of the trace are the timed ones):
  wall        end of the previous step's cfg_ddim_step -> end of this one
  busy        time with at least one kernel running (union of all intervals)
  overlap     share of the busy time with two or more kernels running
  and per kernel family (matrix = gemm2_kernel + splitk_reduce + ffn_geglu_kernel + tsa_kernel: bench.py's `roofline` family)
  union       time with at least one kernel of the family running
  attributed  the family's share of the busy time: every instant is split equally among the kernels running at it -- the
              families' attributed times ADD UP to `busy`, so a family can never exceed the step (the cross-check the isolated
              sum of durations failed in round 5)
  serial      plain sum of the family's dispatch durations (what `--stats` reports)
`--family-tflop F` (algorithmic TFLOP of the matrix family per step: bench.py's roofline.algorithmic_flops_per_launch x
launches_per_step) adds  frac = F / attributed_matrix / 2500  -- the in-step roofline fraction, reproducible from the trace alone.
If the profiler serialises the streams (overlap ~ 0 and wall far above the untraced step) the numbers say so."""
from __future__ import annotations

import argparse
import csv
import glob
import json
import os
import sys

FAMILIES = (
    ("matrix", ("gemm2_kernel", "splitk_reduce", "ffn_geglu_kernel", "tsa_kernel")),
    ("attention", ("attn3_kernel", "xattn_kernel", "tattn3_kernel", "tattn_kernel", "attn_kernel")),
    ("norm", ("gn_", "layernorm_kernel", "groupnorm")),
    ("glue", ("window_", "cfg_ddim_step", "cfg_affine_step", "im2col", "conv3x3_cout_small", "timestep_embedding", "silu", "add_kernel",
              "zero_rows", "bcthw", "bthwc")),
)
PEAK_TFLOPS = 2500.0


def family_of(name: str) -> str:
    for fam, keys in FAMILIES:
        if any(k in name for k in keys):
            return fam
    return "other"


def find_trace(path: str) -> str:
    if os.path.isfile(path):
        return path
    hits = sorted(glob.glob(os.path.join(path, "**", "*kernel_trace.csv"), recursive=True))
    if not hits:
        raise SystemExit(f"no *kernel_trace.csv under {path}")
    return max(hits, key=os.path.getsize)


def load(path: str):
    rows = []
    with open(path, newline="") as f:
        rd = csv.DictReader(f)
        for r in rd:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", ""), r.get("Stream_Id", "")))
    rows.sort()
    return rows


def sweep(rows, t0: int, t1: int):
    """rows clipped to [t0, t1): busy, overlap, per-family union / attributed / serial / launches (ns)"""
    ev = []
    fams = {}
    for s, e, name, *_ in rows:
        if e <= t0 or s >= t1:
            continue
        fam = family_of(name)
        d = fams.setdefault(fam, {"union": 0, "attributed": 0.0, "serial": 0, "launches": 0})
        d["serial"] += e - s
        d["launches"] += 1
        ev.append((max(s, t0), 1, fam))
        ev.append((min(e, t1), -1, fam))
    ev.sort(key=lambda x: (x[0], x[1]))
    active = {}
    n = 0
    busy = over = 0
    prev = t0
    for t, sign, fam in ev:
        dt = t - prev
        if dt > 0 and n > 0:
            busy += dt
            if n > 1:
                over += dt
            for f_, c in active.items():
                if c > 0:
                    fams[f_]["union"] += dt
                    fams[f_]["attributed"] += dt * c / n
        prev = t
        active[fam] = active.get(fam, 0) + sign
        n += sign
    return busy, over, fams


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("trace")
    ap.add_argument("--steps", type=int, required=True, help="timed steps = the last K denoise steps of the trace")
    ap.add_argument("--family-tflop", type=float, default=None, help="algorithmic TFLOP of the matrix family per step")
    ap.add_argument("--step-tflop", type=float, default=None, help="algorithmic TFLOP of the whole step (SURVEY 8d: 36.95 at config 2)")
    ap.add_argument("--json", default=None)
    args = ap.parse_args()
    path = find_trace(args.trace)
    rows = load(path)
    ends = [e for s, e, name, *_ in rows if "cfg_ddim_step" in name or "cfg_affine_step" in name]
    if len(ends) < args.steps + 1:
        raise SystemExit(f"{len(ends)} step-closing dispatches in the trace, need > {args.steps}")
    ends.sort()
    marks = ends[-(args.steps + 1):]
    per_step = []
    for a, b in zip(marks[:-1], marks[1:]):
        busy, over, fams = sweep(rows, a, b)
        per_step.append({"wall": b - a, "busy": busy, "overlap": over, "fams": fams})
    k = len(per_step)

    def mean(f):
        return sum(f(p) for p in per_step) / k
    out = {"trace": os.path.basename(path), "steps": k, "dispatches_per_step": mean(lambda p: sum(d["launches"] for d in p["fams"].values())),
           "wall_ms_per_step": mean(lambda p: p["wall"]) / 1e6, "busy_ms_per_step": mean(lambda p: p["busy"]) / 1e6,
           "overlap_frac_of_busy": mean(lambda p: p["overlap"] / max(p["busy"], 1)),
           "queues": sorted({q for *_x, q, _s in rows}), "streams": sorted({s for *_x, s in rows}), "families": {}}
    names = sorted({f for p in per_step for f in p["fams"]})
    for f in names:
        g = lambda p, key: p["fams"].get(f, {}).get(key, 0)  # noqa: E731
        out["families"][f] = {"launches_per_step": mean(lambda p: g(p, "launches")), "serial_ms": mean(lambda p: g(p, "serial")) / 1e6,
                              "union_ms": mean(lambda p: g(p, "union")) / 1e6, "attributed_ms": mean(lambda p: g(p, "attributed")) / 1e6}
    m = out["families"].get("matrix")
    if m and args.family_tflop:
        out["matrix_roofline"] = {
            "formula": "frac = family_tflop / attributed_ms(matrix) / 2500 TFLOP/s; attributed = the family's share of the step's busy time "
                       "(every instant split equally among the kernels running at it)",
            "family_tflop_per_step": args.family_tflop,
            "in_step": {"tflops": args.family_tflop / (m["attributed_ms"] * 1e-3), "frac": args.family_tflop / (m["attributed_ms"] * 1e-3) / PEAK_TFLOPS},
            "union": {"tflops": args.family_tflop / (m["union_ms"] * 1e-3), "frac": args.family_tflop / (m["union_ms"] * 1e-3) / PEAK_TFLOPS},
            "isolated_sum_of_durations": {"tflops": args.family_tflop / (m["serial_ms"] * 1e-3), "frac": args.family_tflop / (m["serial_ms"] * 1e-3) / PEAK_TFLOPS},
        }
    if args.step_tflop:
        out["whole_step"] = {"tflop": args.step_tflop, "frac_of_mfma_peak_on_traced_wall": args.step_tflop / (out["wall_ms_per_step"] * 1e-3) / PEAK_TFLOPS}
    txt = json.dumps(out, indent=1)
    print(txt)
    if args.json:
        with open(args.json, "w") as fh:
            fh.write(txt + "\n")


if __name__ == "__main__":
    sys.exit(main())
