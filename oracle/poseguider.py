"""ORACLE (test infrastructure, not product): plain-torch fp32 restatement of the reference's PoseGuider
(musev/models/controlnet.py:326-373) on an explicit state dict.  Pinned against the reference's own class executed under
tests/golden/refshim.py (tests/golden/reference_poseguider_*.npz).  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this package."""
from __future__ import annotations

from typing import Dict, Sequence, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor


def param_shapes(conditioning_embedding_channels: int, conditioning_channels: int = 3,
                 block_out_channels: Sequence[int] = (16, 32, 64, 128)) -> Dict[str, Tuple[int, ...]]:
    """state-dict inventory: conv_in, blocks.<2i> (C_i -> C_i), blocks.<2i+1> (C_i -> C_{i+1}, stride 2), conv_out (:334-360)"""
    ch = list(block_out_channels)
    out = {"conv_in.weight": (ch[0], conditioning_channels, 3, 3), "conv_in.bias": (ch[0],)}
    for i, (a, b) in enumerate(zip(ch[:-1], ch[1:])):
        out[f"blocks.{2 * i}.weight"], out[f"blocks.{2 * i}.bias"] = (a, a, 3, 3), (a,)
        out[f"blocks.{2 * i + 1}.weight"], out[f"blocks.{2 * i + 1}.bias"] = (b, a, 3, 3), (b,)
    out["conv_out.weight"], out["conv_out.bias"] = (conditioning_embedding_channels, ch[-1], 3, 3), (conditioning_embedding_channels,)
    return out


def init_state_dict(shapes: Dict[str, Tuple[int, ...]], seed: int) -> Dict[str, Tensor]:
    """seeded weights (conv_out included: the reference zero-initialises it, which would make every output zero)"""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for k, s in shapes.items():
        if k.endswith(".weight"):
            sd[k] = torch.randn(s, generator=g) * (1.6 / (s[1] * 9) ** 0.5)
        else:
            sd[k] = torch.randn(s, generator=g) * 0.1
    return sd


def poseguider_forward(sd: Dict[str, Tensor], conditioning: Tensor) -> Tensor:
    """controlnet.py:363-373 with InflatedConv3d (:308-316) = Conv2d applied per frame: conditioning [b, c, f, h, w]"""
    b, c, f, h, w = conditioning.shape
    x = conditioning.permute(0, 2, 1, 3, 4).reshape(b * f, c, h, w).float()     # "b c f h w -> (b f) c h w"
    x = F.silu(F.conv2d(x, sd["conv_in.weight"], sd["conv_in.bias"], padding=1))             # :364-365
    n_blocks = len([k for k in sd if k.startswith("blocks.") and k.endswith(".weight")])
    for i in range(n_blocks):                                                                 # :367-369
        x = F.silu(F.conv2d(x, sd[f"blocks.{i}.weight"], sd[f"blocks.{i}.bias"], padding=1, stride=2 if i % 2 else 1))
    x = F.conv2d(x, sd["conv_out.weight"], sd["conv_out.bias"], padding=1)                   # :371
    return x.reshape(b, f, *x.shape[1:]).permute(0, 2, 1, 3, 4).contiguous()                  # "(b f) c h w -> b c f h w"
