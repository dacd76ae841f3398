"""Which fp16 roundings does the FREE-RUNNING 20-step loop drift come from?  (runs on the CPU, oracle only.)

    python tools/cpu_loop_rounding_experiment.py [--seeds 5] [--threads 3] [--out profiles/r04b_loop_rounding_ensemble.json]

The net / schedule of tests/test_pipeline_gpu.py::test_twenty_step_drift_against_fp16_torch_floor (2-level SD-1.5-width `musev`,
noise-predictor weights, 16x16 latents, 10 frames, window 6 overlap 2, 20 DDIM steps, guidance 3.5).  The fp32 oracle loop is the
reference; the same loop is re-run with the oracle's rounding hook (oracle.unet3d.HOOK) set to

  fp16_all   every value an fp16 implementation stores is rounded to fp16 (fp32 accumulation): the floor of ANY fp16 UNet
  carry_all  the same, but the identity path of the residual stream keeps its unrounded value while every layer READS the
             fp16-rounded value ("stream_read"): a two-fp16 (hi + lo) carry on the outer residual adds
  carry_l0   the carry only where the stream is 320 channels wide (level 0)

for several latent / prompt seeds; per-step |delta latent|max against the fp32 loop of the same seed.  Test infrastructure."""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import torch  # noqa: E402

ARCH = dict(block_out_channels=(320, 640), layers_per_block=1,
            down_block_types=("CrossAttnDownBlock3D", "DownBlock3D"), up_block_types=("UpBlock3D", "CrossAttnUpBlock3D"))
ALL = ["emb", "conv_in", "gemm", "gn", "gn_out", "ln", "stream_outer", "stream_read", "stream_inner", "attn_q", "attn_p", "attn_o"]


def rtn16(x):
    return x.half().float()


def make_hook(mode):
    if mode == "fp32":
        return None
    cl = set(ALL)

    def hook(kind, v):
        if kind == "stream_outer":
            if mode == "carry_all":
                return v
            if mode == "carry_l0" and 320 in (v.shape[1], v.shape[-1]):
                return v
        return rtn16(v) if kind in cl else v

    return hook


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=5)
    ap.add_argument("--threads", type=int, default=3)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r04b_loop_rounding_ensemble.json"))
    args = ap.parse_args()
    torch.set_num_threads(args.threads)
    from oracle import pipeline as opipe
    from oracle import unet3d
    cfg = unet3d.flavour_config("musev", **ARCH)
    sd = unet3d.calibrate_as_denoiser(unet3d.init_state_dict(cfg, 3), cfg)
    sd16 = {k: rtn16(v) for k, v in sd.items()}
    T, h, w = 10, 16, 16
    modes = ["fp32", "fp16_all", "carry_all", "carry_l0"]
    report = {"net": "musev 2-level (320, 640), noise-predictor weights, 16x16 latents, 10 frames, window 6 overlap 2, 20 DDIM steps, guidance 3.5",
              "seeds": []}
    t0 = time.time()
    for seed in range(7, 7 + args.seeds):
        g = torch.Generator().manual_seed(seed)
        latents = torch.randn(1, 4, T, h, w, generator=g)
        cond = 0.18215 * torch.randn(1, 4, 1, h, w, generator=g)
        prompt = torch.randn(2, 77, 768, generator=g)
        kw = dict(num_inference_steps=20, max_steps=args.steps, guidance_scale=3.5, condition_latents=cond, context_frames=6, context_overlap=2,
                  motion_speed=8.0)
        recs = {}
        for mode in modes:
            rec = []
            unet3d.HOOK = make_hook(mode)
            weights = sd if mode == "fp32" else sd16
            try:
                with torch.no_grad():
                    opipe.denoise_loop(lambda x, t, ehs, **k: unet3d.unet3d_forward(weights, cfg, x if mode == "fp32" else rtn16(x), t,
                                                                                   ehs if mode == "fp32" else rtn16(ehs), **k),
                                       latents, prompt, record_latents=rec, **kw)
            finally:
                unet3d.HOOK = None
            recs[mode] = rec
        entry = {"seed": seed, "latent_absmax": max(r.abs().max().item() for r in recs["fp32"])}
        for mode in modes[1:]:
            entry[mode] = [(a - b).abs().max().item() for a, b in zip(recs[mode], recs["fp32"])]
        report["seeds"].append(entry)
        print(f"seed {seed} ({time.time() - t0:.0f} s): " + "  ".join(f"{m} peak {max(entry[m]):.2e} final {entry[m][-1]:.2e}" for m in modes[1:]), flush=True)
        with open(args.out, "w") as f:
            json.dump(report, f, indent=1)


if __name__ == "__main__":
    main()
