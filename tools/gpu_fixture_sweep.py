"""GPU half of the fixture-sensitivity sweep: the HIP denoise loop on every variant tools/cpu_fixture_sweep.py
left under tools/scratch/fixture_sweep/ (the fp32 oracle loop's per-step latents), free-running over all 20 steps, next to the
torch-fp16 floor of the same run (the oracle loop with the UNet evaluated by plain torch in fp16 on the GPU -- what the reference
itself runs).  The two-fp16 carry on / off is a second invocation with MUSEV_OPS="CARRY=0".
    python tools/gpu_fixture_sweep.py [--tag r05x] [--floor]"""
from __future__ import annotations

import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))

from cpu_fixture_sweep import H, LOOP_KW, SCRATCH, T, W, build, inputs, variants  # noqa: E402
from cpu_loop_rounding_experiment import ARCH  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tag", default="r05")
    ap.add_argument("--floor", action="store_true", help="also run the torch-fp16 floor (oracle loop, UNet by torch fp16 on the GPU)")
    args = ap.parse_args()
    from musev_amd import ops
    from musev_amd.models.unet_loader import load_unet_by_name
    from musev_amd.pipelines.parallel_denoise import ParallelDenoiser
    from oracle import pipeline as opipe
    from oracle import unet3d
    dev = torch.device("cuda", 0)
    latents, cond, prompt = inputs()
    out = {"carry": bool(ops.CARRY), "variants": {}}
    for name, spec in variants():
        path = os.path.join(SCRATCH, name + ".npz")
        if not os.path.exists(path):
            continue
        with np.load(path) as z:
            rec32 = [torch.from_numpy(z[f"latents_step{i + 1}"]) for i in range(20)]
        cfg, sd = build(spec)
        unet = load_unet_by_name("musev", sd_unet_model=sd, dtype=torch.float16, **ARCH).to(dev)
        den = ParallelDenoiser(unet, context_frames=6, context_overlap=2)
        rech = []
        den(latents.to(dev), prompt.to(dev), num_inference_steps=20, guidance_scale=3.5, condition_latents=cond.to(dev), motion_speed=8.0,
            callback=lambda i, t, lat: rech.append(lat.clone().view(1, 4, T, H, W)))
        torch.cuda.synchronize()
        d_hip = [(a.float().cpu() - b).abs().max().item() for a, b in zip(rech, rec32)]
        ent = {"spec": spec, "hip_free_running": d_hip, "hip_peak": max(d_hip)}
        if args.floor:
            rec16 = []
            sdh = {k: v.to(dev, torch.float16) for k, v in sd.items()}

            def unet16(x, t, ehs, **k):
                k = {n: (v.to(dev) if torch.is_tensor(v) else v) for n, v in k.items()}
                return unet3d.unet3d_forward(sdh, cfg, x.to(dev, torch.float16), t, ehs.to(dev, torch.float16), **k).float().cpu()

            with torch.no_grad():
                opipe.denoise_loop(unet16, latents, prompt, condition_latents=cond, record_latents=rec16, **LOOP_KW)
            d16 = [(a - b).abs().max().item() for a, b in zip(rec16, rec32)]
            ent["torch_fp16_floor"] = d16
            ent["torch_fp16_peak"] = max(d16)
        out["variants"][name] = ent
        print(f"{name:22s} carry {int(ops.CARRY)}  HIP free-running peak {max(d_hip):.2e} (final {d_hip[-1]:.2e})" +
              (f"   torch-fp16 floor peak {ent['torch_fp16_peak']:.2e}" if args.floor else ""), flush=True)
        del unet, den
    od = os.path.join(os.environ.get("GRAFT_REPO_ROOT", ROOT), "gpurun_out")
    os.makedirs(od, exist_ok=True)
    with open(os.path.join(od, f"{args.tag}_fixture_sweep_gpu_carry{int(ops.CARRY)}.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
