"""TEST ORACLE (not product code): CPU fp32 restatement of the two side models either side of the denoise loop

  * the VAE *decoder* the pipeline calls after the loop -- ``decode_latents`` (musev/pipelines/pipeline_controlnet.py:233-238 ->
    diffusers ``StableDiffusionPipeline.decode_latents`` -> ``AutoencoderKL.decode``), in temporal chunks of ``decoder_t_segment``
    frames (:2157-2171);
  * the IP-Adapter image projection ``ImageProjModel`` (ip_adapter package, instantiated at musev/models/ip_adapter_loader.py:89-93),
    whose tokens are the ``vision_clip_emb`` input of the UNet (pipeline_controlnet.py:736-774).

PARITY UNPINNED: ``AutoencoderKL`` lives in the un-vendored diffusers fork (requirements.txt:1, a branch) and ``ImageProjModel`` in
the un-vendored ip_adapter package (requirements.txt:2); nothing of either is under /root/reference and the reference holds no
test vectors for them.  What follows restates the PUBLISHED upstream semantics (diffusers v0.24-0.25 ``models/vae.py`` Decoder,
``UNetMidBlock2D``, ``UpDecoderBlock2D``, ``ResnetBlock2D(temb_channels=None)``, ``Attention`` with one 512-wide head; tencent-ailab
IP-Adapter ``ImageProjModel``), functional over the upstream state-dict keys so that real checkpoints load:
  post_quant_conv, decoder.conv_in, decoder.mid_block.{resnets.{0,1}, attentions.0.{group_norm,to_q,to_k,to_v,to_out.0}},
  decoder.up_blocks.i.{resnets.j.{norm1,conv1,norm2,conv2,conv_shortcut}, upsamplers.0.conv}, decoder.conv_norm_out, decoder.conv_out
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module."""
from __future__ import annotations

import math
from collections import OrderedDict
from typing import Dict, List, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor

SCALING_FACTOR = 0.18215  # SD-1.5 vae config.json "scaling_factor"


def vae_config(block_out_channels=(128, 256, 512, 512), layers_per_block: int = 2, latent_channels: int = 4, out_channels: int = 3,
               norm_num_groups: int = 32) -> dict:
    return dict(block_out_channels=tuple(block_out_channels), layers_per_block=layers_per_block, latent_channels=latent_channels,
                out_channels=out_channels, norm_num_groups=norm_num_groups)


def _resnet_shapes(p: str, cin: int, cout: int) -> Dict[str, Tuple[int, ...]]:
    d = {f"{p}.norm1.weight": (cin,), f"{p}.norm1.bias": (cin,), f"{p}.conv1.weight": (cout, cin, 3, 3), f"{p}.conv1.bias": (cout,),
         f"{p}.norm2.weight": (cout,), f"{p}.norm2.bias": (cout,), f"{p}.conv2.weight": (cout, cout, 3, 3), f"{p}.conv2.bias": (cout,)}
    if cin != cout:
        d[f"{p}.conv_shortcut.weight"] = (cout, cin, 1, 1)
        d[f"{p}.conv_shortcut.bias"] = (cout,)
    return d


def decoder_param_shapes(cfg: dict) -> "OrderedDict[str, Tuple[int, ...]]":
    ch = cfg["block_out_channels"]
    lc = cfg["latent_channels"]
    top = ch[-1]
    d: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    d["post_quant_conv.weight"] = (lc, lc, 1, 1)
    d["post_quant_conv.bias"] = (lc,)
    d["decoder.conv_in.weight"] = (top, lc, 3, 3)
    d["decoder.conv_in.bias"] = (top,)
    d.update(_resnet_shapes("decoder.mid_block.resnets.0", top, top))
    a = "decoder.mid_block.attentions.0"
    d[f"{a}.group_norm.weight"] = (top,)
    d[f"{a}.group_norm.bias"] = (top,)
    for n in ("to_q", "to_k", "to_v", "to_out.0"):
        d[f"{a}.{n}.weight"] = (top, top)
        d[f"{a}.{n}.bias"] = (top,)
    d.update(_resnet_shapes("decoder.mid_block.resnets.1", top, top))
    rev = list(reversed(ch))
    out = rev[0]
    for i, c in enumerate(rev):
        prev, out = out, c
        for j in range(cfg["layers_per_block"] + 1):
            d.update(_resnet_shapes(f"decoder.up_blocks.{i}.resnets.{j}", prev if j == 0 else out, out))
        if i != len(rev) - 1:
            d[f"decoder.up_blocks.{i}.upsamplers.0.conv.weight"] = (out, out, 3, 3)
            d[f"decoder.up_blocks.{i}.upsamplers.0.conv.bias"] = (out,)
    d["decoder.conv_norm_out.weight"] = (ch[0],)
    d["decoder.conv_norm_out.bias"] = (ch[0],)
    d["decoder.conv_out.weight"] = (cfg["out_channels"], ch[0], 3, 3)
    d["decoder.conv_out.bias"] = (cfg["out_channels"],)
    return d


def init_state_dict(shapes: Dict[str, Tuple[int, ...]], seed: int, residual_gain: float = 0.3) -> "OrderedDict[str, Tensor]":
    """seeded random weights: N(0, 1/fan_in), the last projection of every residual branch scaled by ``residual_gain`` (keeps the
    ~30-layer random network's activations O(1), as oracle/unet3d.init_state_dict does)"""
    g = torch.Generator().manual_seed(seed)
    sd: "OrderedDict[str, Tensor]" = OrderedDict()
    for k, shp in shapes.items():
        if k.endswith(".weight") and len(shp) == 1:
            v = 1.0 + 0.1 * torch.randn(shp, generator=g)
        elif k.endswith(".bias"):
            v = 0.05 * torch.randn(shp, generator=g)
        else:
            fan_in = 1
            for s in shp[1:]:
                fan_in *= s
            gain = residual_gain if k.endswith(("conv2.weight", "to_out.0.weight")) else 1.0
            v = torch.randn(shp, generator=g) * (gain / math.sqrt(fan_in))
        sd[k] = v
    return sd


def _resnet(sd, p: str, x: Tensor, groups: int) -> Tensor:
    """diffusers ResnetBlock2D(temb_channels=None, eps=1e-6, output_scale_factor=1)"""
    h = F.silu(F.group_norm(x, groups, sd[f"{p}.norm1.weight"], sd[f"{p}.norm1.bias"], eps=1e-6))
    h = F.conv2d(h, sd[f"{p}.conv1.weight"], sd[f"{p}.conv1.bias"], padding=1)
    h = F.silu(F.group_norm(h, groups, sd[f"{p}.norm2.weight"], sd[f"{p}.norm2.bias"], eps=1e-6))
    h = F.conv2d(h, sd[f"{p}.conv2.weight"], sd[f"{p}.conv2.bias"], padding=1)
    if f"{p}.conv_shortcut.weight" in sd:
        x = F.conv2d(x, sd[f"{p}.conv_shortcut.weight"], sd[f"{p}.conv_shortcut.bias"])
    return x + h


def _mid_attention(sd, p: str, x: Tensor, groups: int) -> Tensor:
    """diffusers Attention(C, heads=1, dim_head=C, norm_num_groups, residual_connection=True, bias=True) on [B, C, H, W]"""
    b, c, h, w = x.shape
    n = F.group_norm(x, groups, sd[f"{p}.group_norm.weight"], sd[f"{p}.group_norm.bias"], eps=1e-6)
    t = n.reshape(b, c, h * w).transpose(1, 2)  # [B, HW, C]
    q = F.linear(t, sd[f"{p}.to_q.weight"], sd[f"{p}.to_q.bias"])
    k = F.linear(t, sd[f"{p}.to_k.weight"], sd[f"{p}.to_k.bias"])
    v = F.linear(t, sd[f"{p}.to_v.weight"], sd[f"{p}.to_v.bias"])
    s = torch.softmax(torch.matmul(q, k.transpose(1, 2)) * (c ** -0.5), dim=-1)
    o = F.linear(torch.matmul(s, v), sd[f"{p}.to_out.0.weight"], sd[f"{p}.to_out.0.bias"])
    return x + o.transpose(1, 2).reshape(b, c, h, w)


def vae_decode(sd, cfg: dict, z: Tensor) -> Tensor:
    """AutoencoderKL.decode: post_quant_conv + Decoder.forward; z [N, 4, h, w] (already divided by the scaling factor) -> [N, 3, 8h, 8w]"""
    g = cfg["norm_num_groups"]
    x = F.conv2d(z, sd["post_quant_conv.weight"], sd["post_quant_conv.bias"])
    x = F.conv2d(x, sd["decoder.conv_in.weight"], sd["decoder.conv_in.bias"], padding=1)
    x = _resnet(sd, "decoder.mid_block.resnets.0", x, g)
    x = _mid_attention(sd, "decoder.mid_block.attentions.0", x, g)
    x = _resnet(sd, "decoder.mid_block.resnets.1", x, g)
    n_up = len(cfg["block_out_channels"])
    for i in range(n_up):
        for j in range(cfg["layers_per_block"] + 1):
            x = _resnet(sd, f"decoder.up_blocks.{i}.resnets.{j}", x, g)
        if i != n_up - 1:
            x = F.interpolate(x, scale_factor=2.0, mode="nearest")
            x = F.conv2d(x, sd[f"decoder.up_blocks.{i}.upsamplers.0.conv.weight"], sd[f"decoder.up_blocks.{i}.upsamplers.0.conv.bias"], padding=1)
    x = F.silu(F.group_norm(x, g, sd["decoder.conv_norm_out.weight"], sd["decoder.conv_norm_out.bias"], eps=1e-6))
    return F.conv2d(x, sd["decoder.conv_out.weight"], sd["decoder.conv_out.bias"], padding=1)


def decode_latents(sd, cfg: dict, latents: Tensor, decoder_t_segment: int = 200, scaling_factor: float = SCALING_FACTOR) -> Tensor:
    """pipeline_controlnet.py:2157-2171 + :233-238 + diffusers decode_latents: latents [b, c, t, h, w] -> video [b, 3, t, 8h, 8w] in
    [0, 1], decoded in slices of ``decoder_t_segment`` frames along t (the slicing is a memory measure: results do not depend on it)"""
    b, c, t, h, w = latents.shape
    outs: List[Tensor] = []
    for s in range(0, t, decoder_t_segment):
        seg = latents[:, :, s:s + decoder_t_segment]
        f = seg.shape[2]
        z = seg.permute(0, 2, 1, 3, 4).reshape(b * f, c, h, w) / scaling_factor            # "b c f h w -> (b f) c h w"
        img = (vae_decode(sd, cfg, z) / 2 + 0.5).clamp(0, 1)
        outs.append(img.reshape(b, f, img.shape[1], img.shape[2], img.shape[3]).permute(0, 2, 1, 3, 4))  # "(b f) c h w -> b c f h w"
    return torch.cat(outs, dim=2)


# ---- IP-Adapter ImageProjModel ---------------------------------------------------------------------------------------------------
def image_proj_shapes(cross_attention_dim: int = 768, clip_embeddings_dim: int = 1024, clip_extra_context_tokens: int = 4):
    n = clip_extra_context_tokens * cross_attention_dim
    return OrderedDict([("proj.weight", (n, clip_embeddings_dim)), ("proj.bias", (n,)), ("norm.weight", (cross_attention_dim,)),
                        ("norm.bias", (cross_attention_dim,))])


def image_proj(sd, image_embeds: Tensor, cross_attention_dim: int = 768, clip_extra_context_tokens: int = 4) -> Tensor:
    """ip_adapter.ip_adapter.ImageProjModel.forward: Linear(clip_dim -> tokens * dim) -> [B, tokens, dim] -> LayerNorm(dim)"""
    x = F.linear(image_embeds, sd["proj.weight"], sd["proj.bias"]).reshape(-1, clip_extra_context_tokens, cross_attention_dim)
    return F.layer_norm(x, (cross_attention_dim,), sd["norm.weight"], sd["norm.bias"], eps=1e-5)
