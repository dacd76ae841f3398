#!/usr/bin/env python3
"""Fabric-traffic model of the implicit-GEMM launches per workgroup -> tile order, checked against the per-problem PMC bytes of a
committed profile, then used to PREDICT what the weight-stationary order (mv_gemm_desc.tile_order = 1) does to the step's traffic
(no GPU needed; VERDICT r3 item 5b asks for the measured ratio -- this is the model-side half, the measurement is next round's):

    python tools/gemm_traffic_model.py [--out profiles/<tag>_gemm_traffic_model.json]

Model: workgroup b of a launch runs on XCD (b + y * tiles) % 8 (y = its K slice) and produces the tile the library's own map gives
it (mv_gemm_tile_order: the same inline function the kernel uses, m-major groups or -- group -1 -- n-major).  An XCD fetches every
DISTINCT A row-block and every DISTINCT weight column-block its workgroups touch once per K slice (they are co-resident and walk K
together: the L2 serves the repeats), so a launch moves
    sum over XCDs and K slices of (distinct m-tiles x A tile bytes + distinct n-tiles x W tile bytes) / slices
    + the output (+ the residual) + for a split launch the fp32 slabs written and read back.
Inputs: profiles/r04z_pmc_by_problem.log (measured MB per problem with the tile and split it ran with), the catalogue's block shapes."""
from __future__ import annotations

import json
import math
import os
import re
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
TAPS = {"linear": 1, "conv3x3": 9, "tconv3": 3}


def tile_maps(lib, tiles_m, tiles_n, group):
    n = tiles_m * tiles_n
    tm = np.full(n, -1, dtype=np.int32)
    tn = np.full(n, -1, dtype=np.int32)
    assert lib.mv_gemm_tile_order(tiles_m, tiles_n, group, tm.ctypes.data, tn.ctypes.data) == 0
    return tm, tn


def launch_bytes(lib, mode, M, N, K, epilogue, bm, bn, ns, group):
    cin = K // TAPS[mode]
    tiles_m, tiles_n = -(-M // bm), -(-N // bn)
    nwg = tiles_m * tiles_n
    tm, tn = tile_maps(lib, tiles_m, tiles_n, group)
    a_tile = min(bm, M) * cin * 2          # the rows of an m-tile (a convolution re-reads them per tap out of the L2)
    w_tile = min(bn, N) * K * 2
    ids = np.arange(nwg)
    total = 0.0
    for y in range(ns):
        xcd = (ids + y * nwg) % 8
        for x in range(8):
            mine = xcd == x
            if mine.any():
                total += (len(np.unique(tm[mine])) * a_tile + len(np.unique(tn[mine])) * w_tile) / ns
    n_out = N // 2 if "geglu" in epilogue else N
    total += M * n_out * 2
    if "res" in epilogue:
        total += M * n_out * 2
    if ns > 1:
        total += 2.0 * ns * M * N * 4
    return total


def main():
    out = sys.argv[sys.argv.index("--out") + 1] if "--out" in sys.argv else None
    from musev_amd import _lib
    lib = _lib.load()
    configs = json.load(open(os.path.join(ROOT, "profiles", "r04t_musev512_gemm_tune.json")))["configs"]
    rows = []
    for ln in open(os.path.join(ROOT, "profiles", "r04z_pmc_by_problem.log")):
        m = re.match(r"\s*([\d.]+) \|\s*(\d+) \|\s*([\d.]+) \|\s*([\d.]+) \|\s*([\d.]+) \| \('(\w+)', (\d+), (\d+), (\d+), '([^']*)', (-?\d+), (\d+)\)", ln)
        if m and m.group(6) in TAPS:
            _, launches, meas, alg, _, mode, M, N, K, epi, cfg, ns = m.groups()
            rows.append(dict(mode=mode, M=int(M), N=int(N), K=int(K), epilogue=epi, cfg=int(cfg), nsplit=int(ns), launches=int(launches),
                             measured_mb=float(meas), algorithmic_mb=float(alg)))
    tot = dict(measured=0.0, algorithmic=0.0, model_now=0.0, model_ws=0.0)
    n_ws = 0
    print(f"{'problem':44s} {'x':>3s} {'tile':>8s} {'ns':>2s} | {'measured':>8s} {'model':>8s} {'ratio':>5s} | weight-stationary {'model':>8s}  (MB per launch)")
    for r in rows:
        bm, bn = configs[r["cfg"]][0], configs[r["cfg"]][1]
        now = launch_bytes(lib, r["mode"], r["M"], r["N"], r["K"], r["epilogue"], bm, bn, r["nsplit"], 8) / 1e6
        takes = ws_applies(lib, r)
        ws = launch_bytes(lib, r["mode"], r["M"], r["N"], r["K"], r["epilogue"], bm, bn, r["nsplit"], -1) / 1e6 if takes else now
        n_ws += takes
        r.update(model_now_mb=now, model_ws_mb=ws, ws_taken=bool(takes))
        for k, v in (("measured", r["measured_mb"]), ("algorithmic", r["algorithmic_mb"]), ("model_now", now), ("model_ws", ws)):
            tot[k] += v * r["launches"] / 1e3
        name = f"{r['mode']} {r['M']}x{r['N']}x{r['K']} {r['epilogue']}"
        print(f"{name:44s} {r['launches']:3d} {bm:4d}x{bn:<3d} {r['nsplit']:2d} | {r['measured_mb']:8.1f} {now:8.1f} {now / r['measured_mb']:5.2f} | "
              f"{'yes' if takes else 'no ':3s}              {ws:8.1f}")
    err = [abs(math.log(r["model_now_mb"] / r["measured_mb"])) for r in rows]
    print(f"{len(rows)} problems (the GEMM launches of a config-2 step; the fused feed-forward is not a tiled launch): measured {tot['measured']:.1f} GB per step, "
          f"model {tot['model_now']:.1f} GB ({tot['model_now'] / tot['measured']:.2f} of measured; median |log ratio| per problem {sorted(err)[len(err) // 2]:.2f}), "
          f"algorithmic {tot['algorithmic']:.1f} GB")
    print(f"weight-stationary order where the model says it fetches less ({n_ws} problems): model {tot['model_ws']:.1f} GB per step = "
          f"{tot['model_ws'] / tot['algorithmic']:.2f} x algorithmic (now: model {tot['model_now'] / tot['algorithmic']:.2f} x, measured {tot['measured'] / tot['algorithmic']:.2f} x); "
          f"what stays is the split-K slabs")
    if out:
        with open(out, "w") as f:
            json.dump(dict(totals_gb_per_step=tot, problems=rows), f, indent=1)


def ws_applies(lib, r):
    """the LIBRARY's decision for this problem under its tile (mv_gemm_weight_stationary: csrc/gemm.hip:ws_model_prefers -- the n-major
    order is taken where the same fetch model, for one K slice, says the XCDs fetch at least 5 % less)"""
    import ctypes as C
    from musev_amd import _lib
    d = _lib.GemmDesc()
    d.a, d.w, d.c = 0x10000, 0x20000, 0x30000
    mode = {"linear": 0, "conv3x3": 1, "tconv3": 2}[r["mode"]]
    M, N, K = r["M"], r["N"], r["K"]
    geglu = int("geglu" in r["epilogue"])
    d.M, d.N, d.K, d.mode, d.geglu, d.cfg, d.splitk, d.tile_order = M, N, K, mode, geglu, r["cfg"], r["nsplit"], 1
    cin = K // TAPS[r["mode"]]
    d.lda, d.ldc, d.c1 = cin, (N // 2 if geglu else N), cin
    if mode == 1:
        side = next(s_ for s_ in (64, 32, 16, 8) if M % (s_ * s_) == 0 and M // (s_ * s_) in (13, 26))
        d.hin = d.win = d.hout = d.wout = side
        d.stride, d.upsample = 1, 0
    elif mode == 2:
        d.t, d.hw = 13, M // 13
    got = lib.mv_gemm_weight_stationary(C.byref(d))
    assert got in (0, 1), lib.mv_last_error().decode()
    return bool(got)


if __name__ == "__main__":
    main()
