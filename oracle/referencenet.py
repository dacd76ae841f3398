"""ORACLE -- test infrastructure only (see oracle/unet3d.py): never imported by the product path.

Plain-PyTorch fp32 CPU restatement of ``ReferenceNet2D.forward`` (musev/models/referencenet.py:640-1143) in the
configuration ``load_referencenet_by_name("musev_referencenet")`` builds (referencenet_loader.py:111-119:
``need_block_embs=True, need_self_attn_block_embs=False``): the SD-1.5 UNet2D ENCODER -- conv_in, three
CrossAttnDownBlock2D, one DownBlock2D, UNetMidBlock2DCrossAttn (up blocks are not even constructed) -- returning the 12
down-path residuals and the mid-block output as ``b c t h w`` feature maps, i.e. the ``down_block_refer_embs`` /
``mid_block_refer_emb`` inputs of the UNet3D.  The blocks (vendored in musev/models/unet_2d_blocks.py:812-925,
1006-1075, 646-760) are the layer sequences below; their ResnetBlock2D / Attention / GEGLU arithmetic is the
un-vendored diffusers one already restated in oracle/unet3d.py (unpinned there, unpinned here).

Pinning: tests/golden/reference_referencenet_*.npz are outputs of the reference's own ReferenceNet2D executed in this
container (tests/golden/make_reference_goldens.py); tests/test_oracle_golden.py replays them."""
from __future__ import annotations

from collections import OrderedDict
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F
from einops import rearrange

from . import unet3d as u

Tensor = torch.Tensor


def referencenet_config(**overrides) -> dict:
    """SD-1.5 UNet2D encoder widths; ResnetBlock2D applies its own SiLU to temb (resnet_2d_skip_time_act False)."""
    cfg = dict(in_channels=4, block_out_channels=(320, 640, 1280, 1280), layers_per_block=2,
               down_block_types=("CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "DownBlock2D"),
               attention_head_dim=8, cross_attention_dim=768, norm_num_groups=32, norm_eps=1e-5,
               resnet_2d_skip_time_act=False)
    cfg.update(overrides)
    return cfg


def param_shapes(cfg: dict) -> "OrderedDict[str, Tuple[int, ...]]":
    d: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    ch, L, xd = cfg["block_out_channels"], cfg["layers_per_block"], cfg["cross_attention_dim"]
    temb = ch[0] * 4
    d["conv_in.weight"] = (ch[0], cfg["in_channels"], 3, 3)
    d["conv_in.bias"] = (ch[0],)
    u._lin(d, "time_embedding.linear_1", temb, ch[0])
    u._lin(d, "time_embedding.linear_2", temb, temb)
    cin = ch[0]
    for i, bt in enumerate(cfg["down_block_types"]):
        cout, final, p = ch[i], i == len(ch) - 1, f"down_blocks.{i}"
        for j in range(L):
            u._resnet(d, f"{p}.resnets.{j}", cin if j == 0 else cout, cout, temb)
        if bt == "CrossAttnDownBlock2D":
            for j in range(L):
                u._transformer2d(d, f"{p}.attentions.{j}", cout, xd, False)
        if not final:
            d[f"{p}.downsamplers.0.conv.weight"] = (cout, cout, 3, 3)
            d[f"{p}.downsamplers.0.conv.bias"] = (cout,)
        cin = cout
    c = ch[-1]
    u._transformer2d(d, "mid_block.attentions.0", c, xd, False)
    u._resnet(d, "mid_block.resnets.0", c, c, temb)
    u._resnet(d, "mid_block.resnets.1", c, c, temb)
    return d


def init_state_dict(cfg: dict, seed: int = 3, residual_gain: float = 0.3) -> "OrderedDict[str, Tensor]":
    import math
    g = torch.Generator().manual_seed(seed)
    sd: "OrderedDict[str, Tensor]" = OrderedDict()
    for k, shp in param_shapes(cfg).items():
        if k.endswith(".weight") and len(shp) == 1:
            v = 1.0 + 0.1 * torch.randn(shp, generator=g)
        elif k.endswith(".bias"):
            v = 0.05 * torch.randn(shp, generator=g)
        else:
            fan_in = 1
            for s in shp[1:]:
                fan_in *= s
            v = torch.randn(shp, generator=g) * ((residual_gain if k.endswith(u._RESIDUAL_OUT) else 1.0) / math.sqrt(fan_in))
        sd[k] = v
    return sd


def referencenet_forward(sd: Dict[str, Tensor], cfg: dict, sample: Tensor, timestep, encoder_hidden_states: Tensor,
                         num_frames: int, return_ndim: int = 5, conv_in_add: Optional[Tensor] = None) -> Tuple[List[Tensor], Tensor]:
    """sample [(b t), c, h, w]; encoder_hidden_states [(b t), L, D] -> (12 down features, mid feature), each
    [b, c, t, h, w] (return_ndim 5, referencenet.py:1018-1033) or [(b t), c, h, w] (4)."""
    ch, heads, L = cfg["block_out_channels"], cfg["attention_head_dim"], cfg["layers_per_block"]
    t = timestep if torch.is_tensor(timestep) else torch.tensor([timestep])
    t = t.reshape(-1).expand(sample.shape[0])                                       # referencenet.py:779-793
    emb = u.timestep_embedding_mlp(sd, "time_embedding", u.timesteps_sincos(t, ch[0]))  # :795-803
    ctx = dict(num_frames=1, vis_idx=None, vision_clip_emb=None, ip_adapter_scale=0.0, use_ip=False)
    x = F.conv2d(sample, sd["conv_in.weight"], sd["conv_in.bias"], padding=1)      # :914
    if conv_in_add is not None:  # not part of ReferenceNet2D: lets oracle/controlnet.py reuse this encoder walk
        x = x + conv_in_add
    res: List[Tensor] = [x]                                                          # :962
    for i, bt in enumerate(cfg["down_block_types"]):                                 # :963-1003
        p = f"down_blocks.{i}"
        for j in range(L):
            x = u.resnet_block_2d(sd, f"{p}.resnets.{j}", x, emb, cfg)               # unet_2d_blocks.py:1054-1062 / :884-915
            if bt == "CrossAttnDownBlock2D":
                x = u.transformer_2d(sd, f"{p}.attentions.{j}", x, encoder_hidden_states, heads, ctx)
            res.append(x)
        if i != len(ch) - 1:
            x = F.conv2d(x, sd[f"{p}.downsamplers.0.conv.weight"], sd[f"{p}.downsamplers.0.conv.bias"], stride=2, padding=1)
            res.append(x)
    x = u.resnet_block_2d(sd, "mid_block.resnets.0", x, emb, cfg)                   # unet_2d_blocks.py:724-760
    x = u.transformer_2d(sd, "mid_block.attentions.0", x, encoder_hidden_states, heads, ctx)
    x = u.resnet_block_2d(sd, "mid_block.resnets.1", x, emb, cfg)

    def shape(e: Tensor) -> Tensor:
        return e if return_ndim == 4 else rearrange(e, "(b t) c h w -> b c t h w", t=num_frames)

    return [shape(e) for e in res], shape(x)
