"""Is the loop-parity margin a property of the FIXTURE?  (CPU, oracle only -- the GPU half is tools/gpu_fixture_sweep.py.)

    python tools/cpu_fixture_sweep.py [--threads 3] [--out profiles/r05_fixture_sweep_cpu.json]

The loop tests run on seeded weights turned into a noise predictor by oracle.unet3d.calibrate_as_denoiser: eps = 0.9 x (group-normalised
latent, carried conv_in centre tap -> skip 0 -> last conv_shortcut -> conv_norm_out -> conv_out) + random_gain x (the random network).
This sweeps what that construction fixes -- the weight seed, the share of the random network (random_gain 0.18 / 0.35 / 0.5) and the
carrier's layout (the latent channels mixed by a random orthogonal matrix; the carrier taken from skip 1, one whole level-0 stage
behind conv_in, instead of skip 0) -- on the net / schedule of the 20-step drift test (2-level SD-1.5-width `musev`, 16 x 16 latents,
10 frames, window 6 overlap 2, 20 DDIM steps, guidance 3.5).  Per variant: the fp32 oracle loop (saved under tools/scratch/fixture_sweep/
for the GPU half), and the same loop with the oracle's rounding hook set to "every stored value fp16" (the floor of ANY fp16 UNet)
and to "fp16 with the two-fp16 carry at level 0" (what the HIP forward implements); free-running |delta latent|max per step."""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

from cpu_loop_rounding_experiment import ARCH, make_hook, rtn16  # noqa: E402

SCRATCH = os.path.join(ROOT, "tools", "scratch", "fixture_sweep")
T, H, W = 10, 16, 16
LOOP_KW = dict(num_inference_steps=20, guidance_scale=3.5, context_frames=6, context_overlap=2, motion_speed=8.0)


def variants():
    out = []
    for wseed in (3, 4, 5):
        for gain in (0.18, 0.35, 0.5):
            out.append((f"w{wseed}_g{gain}_skip0", dict(weight_seed=wseed, random_gain=gain)))
        out.append((f"w{wseed}_g0.18_mix", dict(weight_seed=wseed, random_gain=0.18, carrier_mix_seed=100 + wseed)))
        out.append((f"w{wseed}_g0.18_skip1", dict(weight_seed=wseed, random_gain=0.18, carrier_route="skip1")))
        out.append((f"w{wseed}_g0.35_skip1", dict(weight_seed=wseed, random_gain=0.35, carrier_route="skip1")))
    return out


def build(spec: dict):
    from oracle import unet3d
    cfg = unet3d.flavour_config("musev", **ARCH)
    kw = {k: v for k, v in spec.items() if k != "weight_seed"}
    return cfg, unet3d.calibrate_as_denoiser(unet3d.init_state_dict(cfg, spec["weight_seed"]), cfg, **kw)


def inputs(seed: int = 7):
    g = torch.Generator().manual_seed(seed)
    latents = torch.randn(1, 4, T, H, W, generator=g)
    cond = 0.18215 * torch.randn(1, 4, 1, H, W, generator=g)
    prompt = torch.randn(2, 77, 768, generator=g)
    return latents, cond, prompt


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", type=int, default=3)
    ap.add_argument("--only", default=None, help="comma-separated variant names")
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r05_fixture_sweep_cpu.json"))
    args = ap.parse_args()
    torch.set_num_threads(args.threads)
    from oracle import pipeline as opipe
    from oracle import unet3d
    os.makedirs(SCRATCH, exist_ok=True)
    report = {"net": "musev 2-level (320, 640), 16x16 latents, 10 frames, window 6 overlap 2, 20 DDIM steps, guidance 3.5, input seed 7",
              "modes": {"fp16_all": "every stored value rounded to fp16 (fp32 accumulation)", "carry_l0": "the same with the two-fp16 carry on the identity path at level 0"},
              "variants": {}}
    if os.path.exists(args.out):
        report["variants"] = json.load(open(args.out)).get("variants", {})
    latents, cond, prompt = inputs()
    t0 = time.time()
    for name, spec in variants():
        if args.only and name not in args.only.split(","):
            continue
        if name in report["variants"] and os.path.exists(os.path.join(SCRATCH, name + ".npz")):
            continue
        cfg, sd = build(spec)
        sd16 = {k: rtn16(v) for k, v in sd.items()}
        recs = {}
        for mode in ("fp32", "fp16_all", "carry_l0"):
            rec = []
            unet3d.HOOK = make_hook(mode)
            weights = sd if mode == "fp32" else sd16
            try:
                with torch.no_grad():
                    opipe.denoise_loop(lambda x, t, ehs, **k: unet3d.unet3d_forward(weights, cfg, x if mode == "fp32" else rtn16(x), t,
                                                                                   ehs if mode == "fp32" else rtn16(ehs), **k),
                                       latents, prompt, condition_latents=cond, record_latents=rec, **LOOP_KW)
            finally:
                unet3d.HOOK = None
            recs[mode] = rec
        np.savez_compressed(os.path.join(SCRATCH, name + ".npz"), **{f"latents_step{i + 1}": r.numpy().astype(np.float32) for i, r in enumerate(recs["fp32"])})
        ent = {"spec": spec, "latent_absmax": max(r.abs().max().item() for r in recs["fp32"])}
        for mode in ("fp16_all", "carry_l0"):
            ent[mode] = [(a - b).abs().max().item() for a, b in zip(recs[mode], recs["fp32"])]
        report["variants"][name] = ent
        print(f"{name:22s} ({time.time() - t0:5.0f} s) |latent|max {ent['latent_absmax']:.2f}  fp16_all peak {max(ent['fp16_all']):.2e}  carry_l0 peak {max(ent['carry_l0']):.2e}", flush=True)
        with open(args.out, "w") as f:
            json.dump(report, f, indent=1)


if __name__ == "__main__":
    main()
