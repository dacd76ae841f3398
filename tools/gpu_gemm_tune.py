#!/usr/bin/env python3
"""Per-shape tile-configuration tuner for the implicit-GEMM kernel (run on the MI355X):

    python tools/gpu_gemm_tune.py [tag] [--size 512] [--reps 3] [--min-gain 0.03]

Records every implicit-GEMM launch of ONE window forward of the benchmark model (config 2: B = 1 and B = 2, T = 13, 64x64 latents;
--size 768 gives the config-5 problems) through musev_amd.ops.GEMM_RECORD and times every distinct problem on its real operands --
`reps` launches back to back between one HIP event pair -- under every catalogue configuration / split-K factor
(mv_gemm_desc.cfg / .splitk), plus once under the built-in rules.  For every distinct problem
(mode, M, N, K, geglu) it keeps the fastest configuration if that beats the rules by more than --min-gain, and writes

    gpurun_out/<tag>_gemm_tuned.h      -> copy to musev_amd/csrc/gemm_tuned.h, rebuild (exact-match table in front of the rules)
    gpurun_out/<tag>_gemm_tune.json    -> per problem: launches per forward, ms under the rules and under every configuration

Nothing here touches results: every configuration reduces over K in the same order (tests/test_kernel_cpu_sim.py runs all of
them against torch on the host simulator; tests/test_kernels_gpu.py on the GPU)."""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("tag", nargs="?", default="tune")
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--min-gain", type=float, default=0.03)
    ap.add_argument("--flavour", default="musev")
    ap.add_argument("--no-split", action="store_true", help="skip the forced split-K sweeps")
    ap.add_argument("--max-cfg", type=int, default=-1, help="only catalogue ids <= this (default: all)")
    args = ap.parse_args()
    import bench
    from musev_amd import _lib, ops
    lib = _lib.load()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    unet = bench.build_unet(args.flavour, dev)
    h = w = args.size // 8
    t = 13
    t_dev = torch.full((1,), 601.0, device=dev)
    # the loop runs the CFG halves as two batch-1 forwards on two streams by default and as one batch-2 forward otherwise
    # (MUSEV_HALF_STREAMS=0, bench.py's instrumented pass): both sets of problem sizes are tuned
    inputs = {}
    for b in (1, 2):
        g = torch.Generator().manual_seed(b)
        kw = dict(sample_index=torch.arange(1, t, device=dev), vision_conditon_frames_sample_index=[0], sample_frame_rate=8.0)
        if args.flavour == "musev_referencenet":
            shapes, mid = bench.refer_shapes(h, w)
            kw["down_block_refer_embs"] = [torch.randn(b, c, 1, a, b_, generator=g).to(dev) for c, a, b_ in shapes]
            kw["mid_block_refer_emb"] = torch.randn(b, mid[0], 1, mid[1], mid[2], generator=g).to(dev)
            kw["vision_clip_emb"] = torch.randn(b, 4, 768, generator=g).to(dev)
        inputs[b] = (torch.randn(b * t * h * w, 4, generator=g).half().to(dev), torch.randn(b, 77, 768, generator=g).to(dev), kw)

    def forward():
        outs = []
        for b, (x, ehs, kw) in inputs.items():
            outs.append(unet.forward_rows(x, b, t, h, w, t_dev, ehs, **kw))
        return torch.cat(outs, dim=0)

    n_cfg = lib.mv_gemm_num_configs()
    descs = []
    for i in range(n_cfg):
        d = (C.c_int32 * 5)()
        lib.mv_gemm_config_desc(i, d)
        descs.append(list(d))
    names = {0: "linear", 1: "conv3x3", 2: "tconv3"}
    # (cfg, splitk) combinations: the rules (incl. the split-K rule), every catalogue entry unsplit, and 2 / 4 / 8 K slices on the
    # tiles that make sense for the small-M, long-K problems (a forced split is clamped by the library's workspace cap, so on the
    # large problems these runs repeat the unsplit kernel)
    split_cfgs = [c for c in range(n_cfg) if descs[c][0] <= 128 and descs[c][1] in (128, 160) and descs[c][2] >= 4]
    combos = [(-2, 0)] + [(c, 1) for c in range(n_cfg)] + [(c, s) for c in split_cfgs for s in (2, 4, 8)]
    if args.no_split:
        combos = [cs for cs in combos if cs[1] <= 1]
    if args.max_cfg >= 0:
        combos = [cs for cs in combos if cs[0] <= args.max_cfg]
    # ONE eager forward is recorded (musev_amd.ops.GEMM_RECORD: descriptor copies in launch order, tensors kept alive); every
    # DISTINCT problem is then re-issued from its first recorded descriptor under every configuration / split factor -- `reps`
    # launches back to back between one HIP event pair (ops.replay_gemms), i.e. device time without host launch gaps.  The choice
    # travels in the descriptor (cfg / splitk); a forced split gets its workspace here.
    forward()   # warm-up (packed weights, caches, code objects)
    torch.cuda.synchronize()
    ops.GEMM_RECORD = []
    forward()
    torch.cuda.synchronize()
    rec, ops.GEMM_RECORD = ops.GEMM_RECORD, None
    table = {}   # key -> {"n": launches per forward pair, "desc": first descriptor, "keep": tensors, "ms": {(cfg, splitk): ms per launch}}
    for d, keep, _nb in rec:
        key = (int(d.mode), int(d.M), int(d.N), int(d.K), int(d.geglu), int(bool(d.ln_colsum)))
        ent = table.setdefault(key, {"n": 0, "desc": d, "keep": keep, "ms": {}})
        ent["n"] += 1
    ws_cache = {}

    def time_problem(ent, cfg, splitk):
        d = _lib.GemmDesc.from_buffer_copy(ent["desc"])
        d.cfg, d.splitk = cfg, splitk
        d.workspace, d.workspace_bytes = None, 0
        need = lib.mv_gemm_workspace_bytes(C.byref(d))
        if need < 0:
            return None
        if need > 0:
            ws = ws_cache.get(need)
            if ws is None:
                ws = ws_cache[need] = torch.empty(need, dtype=torch.uint8, device=dev)
            d.workspace, d.workspace_bytes = ws.data_ptr(), need
        one = [(d, (), 0)]
        ops.replay_gemms(one, 1)
        return ops.replay_gemms(one, args.reps) / args.reps

    for cfg, splitk in combos:
        tot = 0.0
        for key, ent in table.items():
            ms = time_problem(ent, cfg, splitk)
            if ms is None:   # the configuration cannot run this problem (GEGLU needs an even TN): not a candidate
                continue
            ent["ms"][(cfg, splitk)] = ms
            tot += ms * ent["n"]
        label = "rules" if cfg == -2 else f"cfg {cfg:2d} {descs[cfg][0]}x{descs[cfg][1]} {descs[cfg][2]}w x{descs[cfg][4]} splitk {splitk}"
        print(f"{label:40s} GEMM family {tot:7.2f} ms / forward pair", flush=True)
    del rec

    rules_total = tuned_total = 0.0
    entries, report = [], []
    for key, ent in sorted(table.items(), key=lambda kv: -kv[1]["ms"].get((-2, 0), 0.0) * kv[1]["n"]):
        base = ent["ms"][(-2, 0)]
        # ties (a clamped split repeats the unsplit kernel) go to the smaller split factor
        best = min((c for c in ent["ms"] if c[0] >= 0), key=lambda c: (ent["ms"][c], c[1]))
        pick = best if ent["ms"][best] < (1.0 - args.min_gain) * base else None
        rules_total += base * ent["n"]
        tuned_total += (ent["ms"][pick] if pick is not None else base) * ent["n"]
        mode, M, N, K, geglu, ln = key
        flops = 2.0 * M * N * K
        report.append({"mode": names[mode], "M": M, "N": N, "K": K, "geglu": geglu, "ln": ln, "launches": ent["n"], "rules_ms": base,
                       "rules_tflops": flops / base / 1e9, "best_cfg": best[0], "best_splitk": best[1], "best_ms": ent["ms"][best],
                       "best_tflops": flops / ent["ms"][best] / 1e9, "picked": list(pick) if pick else None,
                       "ms": {f"{c}/{s}": v for (c, s), v in ent["ms"].items()}})
        # every problem gets an entry (the lookup inherits choices across sizes by nearest M, so "the rules measured best" must be
        # recorded too: cfg -2)
        entries.append((mode, M, N, K, geglu, ln) + ((pick[0], pick[1], base, ent["ms"][pick]) if pick is not None else (-2, 0, base, base)))
    out_dir = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out_dir, exist_ok=True)
    with open(os.path.join(out_dir, f"{args.tag}_gemm_tune.json"), "w") as f:
        json.dump({"device": torch.cuda.get_device_name(0), "size": args.size, "flavour": args.flavour, "configs": descs,
                   "rules_ms_per_forward": rules_total, "tuned_ms_per_forward": tuned_total, "problems": report}, f, indent=1)
    lines = ["// gemm_tuned.h -- per-shape tile configurations of the implicit-GEMM kernel, measured on the MI355X.",
             "// GENERATED by tools/gpu_gemm_tune.py (do not edit by hand); ids refer to MV_GEMM_CFGS in gemm.hip.",
             f"// {torch.cuda.get_device_name(0)}; {args.flavour}, {args.size}x{args.size}, T = 13, one batch-1 + one batch-2 forward; rules "
             f"{rules_total:.2f} ms -> table {tuned_total:.2f} ms of GEMM time.",
             "// {mode, M, N, K, geglu, ln, cfg, nsplit}   ln: LayerNorm-folded launch; nsplit: K slices (clamped by the workspace cap); 0 = split-K rule; "
             "cfg -2: the rules' own choice measured best",
             "static const GemmTuned kGemmTuned[] = {"]
    for mode, M, N, K, geglu, ln, pick, split, base, best in entries:
        lines.append(f"    {{{mode}, {M}, {N}, {K}, {geglu}, {ln}, {pick}, {split}}},  // {names[mode]}{' +LN' if ln else ''}: {base * 1e3:.0f} -> {best * 1e3:.0f} us")
    lines += ["    {-1, 0, 0, 0, 0, 0, -1, 0},  // sentinel (never matches)", "};", f"static const int kNumGemmTuned = {len(entries)};", ""]
    with open(os.path.join(out_dir, f"{args.tag}_gemm_tuned.h"), "w") as f:
        f.write("\n".join(lines))
    print(f"rules {rules_total:.2f} ms -> tuned {tuned_total:.2f} ms per forward over {len(table)} problems, {sum(1 for e in entries if e[6] >= 0)} non-rule entries")
    for r in report[:24]:
        print(f"{r['mode']:8s} M{r['M']:<7d} N{r['N']:<6d} K{r['K']:<6d} g{r['geglu']} x{r['launches']:<3d} rules {r['rules_ms'] * 1e3:6.0f} us "
              f"{r['rules_tflops']:5.0f} TF | best cfg {r['best_cfg']:2d}/{r['best_splitk']} {r['best_ms'] * 1e3:6.0f} us {r['best_tflops']:5.0f} TF")


if __name__ == "__main__":
    main()
