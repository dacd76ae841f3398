"""HBM traffic of the implicit-GEMM launches PER PROBLEM: joins the per-dispatch PMC rows tools/gpu_profile.sh keeps
(gpurun_out/<tag>_pmc_{FETCH_SIZE,WRITE_SIZE}_gemm_rows.csv; eager launches, so the dispatches of a step appear in host issue order)
with the launch list `bench.py --dump-gemm-launches` wrote for the same build (gpurun_out/<tag>_gemm_launches.json), and prints, per
distinct (mode, M, N, K, epilogue), launches per step, measured bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950 read correction,
see pmc_summary.py) against the algorithmic bytes -- where the 2 x of the family comes from.   python tools/pmc_by_problem.py <tag>"""
import csv
import json
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def dispatch_values(path):
    """[(dispatch id, kernel name, counter value)] in dispatch order"""
    rows = []
    with open(path) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r.get("Dispatch_Id", len(rows))), r.get("Kernel_Name", ""), float(r.get("Counter_Value", 0.0))))
    rows.sort(key=lambda t: t[0])
    return rows


def join(launches, fetch, write):
    """per launch: the gemm2_kernel dispatch (+ the splitk_reduce dispatch behind it when the launch was split).  The counter passes
    hold several identical steps (warm-up + timed + the first eager pass): the LAST complete step is used."""
    have_ffn = any("ffn_geglu" in r[1] for r in fetch)
    if not have_ffn:  # rows kept before the fused feed-forward's dispatches were part of the filter: its launches cannot be joined
        launches = [l for l in launches if l["mode"] != 3]
    if not any("xab_kernel" in r[1] for r in fetch):  # (likewise: rows kept before the fused cross-attention sub-block was part of the filter)
        launches = [l for l in launches if l["mode"] != 5]
    per_step = sum(1 + (1 if l["nsplit"] > 1 else 0) for l in launches)
    out = []
    for rows in (fetch, write):
        if len(rows) < per_step:
            raise SystemExit(f"only {len(rows)} GEMM dispatches in the counter pass, a step has {per_step}")
        out.append(rows[len(rows) - per_step:])
    f, w = out
    res, i = [], 0
    for l in launches:
        n = 1 + (1 if l["nsplit"] > 1 else 0)
        want = "ffn_geglu" if l["mode"] == 3 else "tsa_kernel" if l["mode"] == 4 else "xab_kernel" if l["mode"] == 5 else "gemm2_kernel"
        if not (want in f[i][1] and (n == 1 or "splitk_reduce" in f[i + 1][1])):
            raise SystemExit(f"dispatch order does not match the launch list at launch {len(res)}: {f[i][1][:60]}")
        fb = sum(v for _d, _n, v in f[i:i + n])
        wb = sum(v for _d, _n, v in w[i:i + n])
        res.append((l, (2.0 * fb + wb) * 1024.0))
        i += n
    return res


def main():
    tag = sys.argv[1]
    out = os.path.join(ROOT, "gpurun_out")
    launches = json.load(open(os.path.join(out, f"{tag}_gemm_launches.json")))
    fetch = dispatch_values(os.path.join(out, f"{tag}_pmc_FETCH_SIZE_gemm_rows.csv"))
    write = dispatch_values(os.path.join(out, f"{tag}_pmc_WRITE_SIZE_gemm_rows.csv"))
    agg = defaultdict(lambda: [0, 0.0, 0.0])
    for l, measured in join(launches, fetch, write):
        key = (("linear", "conv3x3", "tconv3", "ffn_fused", "tsa_fused", "xab_fused")[l["mode"]], l["M"], l["N"], l["K"], "geglu" if l["geglu"] else "ln" if l["ln"] else "res" if l["residual"] else "-",
               l["cfg"], l["nsplit"])
        a = agg[key]
        a[0] += 1
        a[1] += measured
        a[2] += l["algorithmic_bytes"]
    tot_m = sum(a[1] for a in agg.values())
    tot_a = sum(a[2] for a in agg.values())
    launches = [l for l, _m in join(launches, fetch, write)]
    print(f"{len(launches)} launches per step: measured {tot_m / 1e9:.1f} GB, algorithmic {tot_a / 1e9:.1f} GB, ratio {tot_m / tot_a:.2f}")
    print("excess GB | launches | measured MB | algorithmic MB | ratio | problem (mode, M, N, K, epilogue, cfg, K slices)")
    for key, a in sorted(agg.items(), key=lambda kv: -(kv[1][1] - kv[1][2]))[:40]:
        print(f"{(a[1] - a[2]) / 1e9:8.2f} | {a[0]:4d} | {a[1] / a[0] / 1e6:9.1f} | {a[2] / a[0] / 1e6:9.1f} | {a[1] / a[2]:5.2f} | {key}")


if __name__ == "__main__":
    main()
