"""ORACLE -- test infrastructure only: plain-torch fp32 restatement of the ControlNet forward the reference calls once per
window and step (musev/pipelines/pipeline_controlnet.py:1251-1260, wrapper musev/models/controlnet.py:290-306).

PINNING STATUS -- read before trusting: the class is diffusers' ``ControlNetModel`` from the un-vendored fork
(git+https://github.com/TMElyralab/diffusers.git@tme, a branch; API level upstream v0.24-0.25).  Its source is not
under /root/reference, so the TOP-LEVEL composition below (conv_in + conditioning embedding, zero convolutions,
conditioning / guess-mode scales) restates the published upstream algorithm and is **parity unpinned**.  What it is composed
of IS pinned by executing reference source in this container:
  * the SD-1.5 UNet2D encoder walk (conv_in, CrossAttnDownBlock2D x3, DownBlock2D, UNetMidBlock2DCrossAttn) is
    oracle/referencenet.py, pinned against the reference's own ReferenceNet2D (a clone of that encoder);
  * the conditioning embedding has the layer structure of the reference's PoseGuider (which was modelled on upstream's
    ControlNetConditioningEmbedding: conv_in, [conv, conv stride 2] per level, zero conv_out, SiLU between), pinned in
    oracle/poseguider.py.
The fork-only ``controlnet_cond_latents`` argument (CHANGES:5) cannot be restated and is not accepted."""
from __future__ import annotations

from collections import OrderedDict
from typing import Dict, List, Tuple

import torch
import torch.nn.functional as F

from . import poseguider as pg
from . import referencenet as rn

Tensor = torch.Tensor


def controlnet_config(**overrides) -> dict:
    cfg = rn.referencenet_config()
    cfg.update(conditioning_channels=3, conditioning_embedding_out_channels=(16, 32, 96, 256))
    cfg.update(overrides)
    return cfg


def n_residuals(cfg: dict) -> int:
    ch, L = cfg["block_out_channels"], cfg["layers_per_block"]
    return 1 + sum(L + (0 if i == len(ch) - 1 else 1) for i in range(len(ch)))


def param_shapes(cfg: dict) -> "OrderedDict[str, Tuple[int, ...]]":
    d = rn.param_shapes(cfg)
    ch, L = cfg["block_out_channels"], cfg["layers_per_block"]
    for k, v in pg.param_shapes(ch[0], cfg["conditioning_channels"], cfg["conditioning_embedding_out_channels"]).items():
        d[f"controlnet_cond_embedding.{k}"] = v
    widths = [ch[0]]
    for i, c in enumerate(ch):
        widths += [c] * (L + (0 if i == len(ch) - 1 else 1))
    for i, c in enumerate(widths):
        d[f"controlnet_down_blocks.{i}.weight"], d[f"controlnet_down_blocks.{i}.bias"] = (c, c, 1, 1), (c,)
    d["controlnet_mid_block.weight"], d["controlnet_mid_block.bias"] = (ch[-1], ch[-1], 1, 1), (ch[-1],)
    return d


def init_state_dict(cfg: dict, seed: int = 5) -> "OrderedDict[str, Tensor]":
    """seeded weights; the zero convolutions are randomised (upstream zero-initialises them: every residual would be 0)"""
    sd = rn.init_state_dict(cfg, seed)
    g = torch.Generator().manual_seed(seed + 1000)
    for k, shp in param_shapes(cfg).items():
        if k in sd:
            continue
        if k.endswith(".bias"):
            sd[k] = 0.05 * torch.randn(shp, generator=g)
        else:
            fan_in = shp[1] * shp[2] * shp[3]
            sd[k] = torch.randn(shp, generator=g) * ((1.6 if k.startswith("controlnet_cond_embedding") else 1.0) / fan_in ** 0.5)
    return sd


def controlnet_forward(sd: Dict[str, Tensor], cfg: dict, sample: Tensor, timestep, encoder_hidden_states: Tensor,
                       controlnet_cond: Tensor, conditioning_scale: float = 1.0, guess_mode: bool = False) -> Tuple[List[Tensor], Tensor]:
    """published ControlNetModel.forward: sample [(b t), 4, h, w], controlnet_cond [(b t), 3, 8h, 8w] -> 13 residuals + mid"""
    emb_sd = {k[len("controlnet_cond_embedding."):]: v for k, v in sd.items() if k.startswith("controlnet_cond_embedding.")}
    cond = pg.poseguider_forward(emb_sd, controlnet_cond[:, :, None])[:, :, 0]          # controlnet_cond_embedding
    res, mid = rn.referencenet_forward(sd, cfg, sample, timestep, encoder_hidden_states, num_frames=1, return_ndim=4,
                                       conv_in_add=cond)                                  # sample = conv_in(sample) + cond; encoder
    down = [F.conv2d(r, sd[f"controlnet_down_blocks.{i}.weight"], sd[f"controlnet_down_blocks.{i}.bias"]) for i, r in enumerate(res)]
    mid = F.conv2d(mid, sd["controlnet_mid_block.weight"], sd["controlnet_mid_block.bias"])
    if guess_mode:
        scales = torch.logspace(-1, 0, len(down) + 1) * conditioning_scale                # 0.1 ... 1.0
        down = [d * s for d, s in zip(down, scales)]
        mid = mid * scales[-1]
    else:
        down = [d * conditioning_scale for d in down]
        mid = mid * conditioning_scale
    return down, mid
