"""Model-level parity cases: musev_amd.UNet3DConditionModel (HIP kernels through the C ABI, fp16) against the oracle
(plain torch fp32 on the CPU) on identical seeded weights and inputs.  Tolerance: the north-star bound
|delta|_max < 1e-2 on the predicted noise (outputs are O(1))."""
from __future__ import annotations

import time
from typing import Dict, Optional

import torch

TOL = 1e-2

ARCHS = {
    # 2-level variant of the SD-1.5 topology: exercises head dims 40 and 80, every block type, both samplers
    "small": dict(block_out_channels=(320, 640), layers_per_block=1,
                  down_block_types=("CrossAttnDownBlock3D", "DownBlock3D"), up_block_types=("UpBlock3D", "CrossAttnUpBlock3D")),
    # 3-level variant for the referencenet flavours: the reference's refer-emb slice of the LAST down block
    # (unet_3d_condition.py:1090-1095) is only channel-compatible when the last two levels have equal width
    "small3": dict(block_out_channels=(320, 640, 640), layers_per_block=1,
                   down_block_types=("CrossAttnDownBlock3D", "CrossAttnDownBlock3D", "DownBlock3D"),
                   up_block_types=("UpBlock3D", "CrossAttnUpBlock3D", "CrossAttnUpBlock3D")),
    # the real SD-1.5 widths (1.4 B parameters)
    "full": dict(),
}


def refer_shapes(cfg, h, w):
    """ReferenceNet feature shapes for a latent of h x w: down_block_res_samples of a UNet2D with the same widths."""
    ch = cfg["block_out_channels"]
    L = cfg["layers_per_block"]
    out = [(ch[0], h, w)]
    hh, ww = h, w
    for i, c in enumerate(ch):
        for _ in range(L):
            out.append((c, hh, ww))
        if i != len(ch) - 1:
            hh, ww = hh // 2, ww // 2
            out.append((c, hh, ww))
    return out, (ch[-1], hh, ww)


def make_inputs(cfg, b=2, t=5, h=16, w=16, seed=0, n_cond=1, text_len=77):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(b, cfg["in_channels"], t, h, w, generator=g)
    ehs = torch.randn(b, text_len, cfg["cross_attention_dim"], generator=g)
    kw: Dict[str, object] = dict(sample_frame_rate=8)
    if n_cond:
        kw["vision_conditon_frames_sample_index"] = torch.arange(n_cond)
        kw["sample_index"] = torch.arange(n_cond, t)
    if cfg["need_refer_emb"]:
        shapes, mid = refer_shapes(cfg, h, w)
        # the CFG halves share the same reference features (pipeline_controlnet.py:844-858)
        kw["down_block_refer_embs"] = [torch.randn(1, c, 1, hh, ww, generator=g).repeat(b, 1, 1, 1, 1) for c, hh, ww in shapes]
        kw["mid_block_refer_emb"] = torch.randn(1, mid[0], 1, mid[1], mid[2], generator=g).repeat(b, 1, 1, 1, 1)
    if cfg["ip_adapter_cross_attn"]:
        kw["vision_clip_emb"] = torch.randn(b, 4, cfg["cross_attention_dim"], generator=g)
        kw["ip_adapter_scale"] = 0.8
    return x, ehs, kw


def to_dev(v, dev):
    if torch.is_tensor(v):
        return v.to(dev)
    if isinstance(v, (list, tuple)):
        return [to_dev(u, dev) for u in v]
    return v


def run_case(flavour: str, arch: str, *, b=2, t=5, h=16, w=16, timestep=601, n_cond=1, seed=3, skip_temporal=None,
             threads: Optional[int] = None) -> Dict:
    from oracle import unet3d
    from musev_amd.models.unet_loader import load_unet_by_name
    if threads:
        torch.set_num_threads(threads)
    over = ARCHS[arch]
    cfg = unet3d.flavour_config(flavour, **over)
    sd = unet3d.init_state_dict(cfg, seed)
    x, ehs, kw = make_inputs(cfg, b, t, h, w, seed=seed + 100, n_cond=n_cond)
    t0 = time.time()
    okw = dict(kw)
    if skip_temporal is not None:
        okw["skip_temporal_layers"] = skip_temporal
    ref = unet3d.unet3d_forward(sd, cfg, x, torch.tensor(timestep), ehs, **okw)
    t_cpu = time.time() - t0
    model = load_unet_by_name(flavour, sd_unet_model=sd, dtype=torch.float16, **over).to("cuda")
    hkw = {k: to_dev(v, "cuda") for k, v in kw.items()}
    if skip_temporal is not None:
        hkw["skip_temporal_layers"] = skip_temporal
    t0 = time.time()
    got = model(x.to("cuda"), torch.tensor(timestep, device="cuda"), encoder_hidden_states=ehs.to("cuda"), return_dict=False, **hkw)[0]
    torch.cuda.synchronize()
    t_gpu_first = time.time() - t0
    t0 = time.time()
    got2 = model(x.to("cuda"), torch.tensor(timestep, device="cuda"), encoder_hidden_states=ehs.to("cuda"), return_dict=False, **hkw)[0]
    torch.cuda.synchronize()
    t_gpu = time.time() - t0
    err = (got.float().cpu() - ref).abs()
    return {
        "name": f"unet {flavour}/{arch} b{b} t{t} {h}x{w} n_cond{n_cond} skip_temporal={skip_temporal}",
        "max_abs_err": err.max().item(), "mean_abs_err": err.mean().item(), "ref_absmax": ref.abs().max().item(),
        "ref_absmean": ref.abs().mean().item(), "tol": TOL,
        "deterministic": bool(torch.equal(got, got2)),
        "finite": bool(torch.isfinite(got).all()),
        "ok": bool(err.max().item() < TOL) and bool(torch.isfinite(got).all()),
        "cpu_oracle_s": t_cpu, "gpu_first_s": t_gpu_first, "gpu_second_s": t_gpu,
    }


MODEL_CASES = [
    ("musev_small", lambda: run_case("musev", "small")),
    ("musev_small_nocond", lambda: run_case("musev", "small", n_cond=0, t=4)),
    ("musev_small_2d", lambda: run_case("musev", "small", t=1, n_cond=0, skip_temporal=True)),
    ("referencenet_small", lambda: run_case("musev_referencenet", "small3")),
    ("musev_full", lambda: run_case("musev", "full", t=5, h=16, w=16)),
    ("referencenet_full", lambda: run_case("musev_referencenet", "full", t=4, h=16, w=16)),
]
