#!/usr/bin/env python3
"""Generates tests/golden/reference_loop_<case>.npz: per-step latents of the denoise loop at BASELINE config-2 size.

    python tests/golden/make_loop_goldens.py [--case NAME ...] [--threads N]   (needs /root/reference; minutes per step, ~30 GB)

The UNet inside the loop is the REFERENCE'S OWN ``UNet3DConditionModel`` (/root/reference/musev, third-party packages replaced
by tests/golden/refshim.py as in make_reference_goldens.py); the loop around it is oracle/pipeline.py:denoise_loop (the reference's
``MusevControlNetPipeline.__call__`` sits on the un-vendored diffusers pipeline base, so the loop body itself cannot be executed
here -- its restatement is pinned piecewise: window schedule, DDIM step and index helpers against the reference's own functions).
Only seeds and OUTPUTS are stored; tests/test_pipeline_gpu.py regenerates weights and inputs from the seeds (golden_cases.py)."""
from __future__ import annotations

import logging
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

import refshim  # noqa: E402

refshim.install("/root/reference")
logging.disable(logging.CRITICAL)

from golden_cases import (FLAVOUR_CTOR_KWARGS, LOOP_CASES_AT_SIZE, loop_case_inputs, loop_case_state_dict,  # noqa: E402
                          loop_case_unet_kwargs)
from oracle import pipeline as opipe  # noqa: E402


def main():
    from musev.models.unet_3d_condition import UNet3DConditionModel
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--case", action="append", default=None, help="case name(s) of golden_cases.LOOP_CASES_AT_SIZE (default: all)")
    ap.add_argument("--threads", type=int, default=os.cpu_count() or 1)
    ap.add_argument("--resume", action="store_true", help="continue from reference_loop_<case>.npz.part.npz (DDIM: stateless steps)")
    args = ap.parse_args()
    torch.set_num_threads(args.threads)
    for name, case in LOOP_CASES_AT_SIZE.items():
        if args.case and name not in args.case:
            continue
        cfg, sd = loop_case_state_dict(case)
        ctor = dict(FLAVOUR_CTOR_KWARGS[case["flavour"]])
        ctor.update(block_out_channels=tuple(cfg["block_out_channels"]), layers_per_block=cfg["layers_per_block"],
                    down_block_types=tuple(cfg["down_block_types"]), up_block_types=tuple(cfg["up_block_types"]),
                    cross_attention_dim=cfg["cross_attention_dim"], attention_head_dim=cfg["attention_head_dim"])
        model = UNet3DConditionModel(**ctor).eval()
        model.load_state_dict(sd, strict=True)
        del sd
        latents, cond, prompt = loop_case_inputs(case)
        side = loop_case_unet_kwargs(case, cfg)
        t0 = [time.time()]

        def unet_fn(x, t, ehs, **kw):
            out = model(x, t, encoder_hidden_states=ehs, return_dict=False, **kw)[0]
            print(f"  forward {tuple(x.shape)} t={int(t)} {time.time() - t0[0]:.0f} s", flush=True)
            return out

        path = os.path.join(HERE, f"reference_loop_{name}.npz")

        class _Rec(list):  # a 20-step run is hours of CPU: the file is rewritten after every step
            def append(self, r):
                super().append(r)
                np.savez_compressed(path + ".part.npz", **{f"latents_step{i + 1}": v.numpy().astype(np.float32) for i, v in enumerate(self)})
                print(f"  step {len(self)} |latent|max {float(r.abs().max()):.3f} {time.time() - t0[0]:.0f} s", flush=True)

        rec = _Rec()
        start = 0
        if args.resume and os.path.exists(path + ".part.npz"):
            with np.load(path + ".part.npz") as z:
                while f"latents_step{start + 1}" in z:
                    list.append(rec, torch.from_numpy(z[f"latents_step{start + 1}"]))
                    start += 1
            if start:
                latents = rec[-1].clone()
            print(f"  resuming {name} after step {start}", flush=True)
        with torch.no_grad():
            opipe.denoise_loop(unet_fn, latents, prompt, num_inference_steps=case["num_inference_steps"], max_steps=case["steps"],
                               guidance_scale=case["guidance_scale"], condition_latents=cond, context_frames=case["context_frames"],
                               context_overlap=case["context_overlap"], motion_speed=8.0, record_latents=rec, unet_kwargs=side, start_step=start,
                               vision_condition_latent_index=case.get("vision_condition_latent_index"))
            out = {f"latents_step{i + 1}": r.numpy().astype(np.float32) for i, r in enumerate(rec)}
        np.savez_compressed(path, **out)
        os.remove(path + ".part.npz")
        print("loop", name, [f"{float(r.abs().max()):.3f}" for r in rec], f"{time.time() - t0[0]:.0f} s", flush=True)


if __name__ == "__main__":
    main()
