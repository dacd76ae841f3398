"""TemporalConvLayer (musev/models/resnet.py:33-135): 4 x [GroupNorm(32) over (C/32, T, H, W) -> SiLU ->
Conv3d (3,1,1)], identity + |temporal_weight| * h.  The Conv3d is an implicit GEMM with K = 3*C that gathers frames
t-1, t, t+1 in place (zero halo in t); the two full permute copies of the reference (:106-108, :133) vanish."""
from __future__ import annotations

from typing import Optional

import torch
from torch import nn

from .. import ops
from . import Model_Register
from .layers import HipModule, w16
from .runtime import Ctx, Geo


@Model_Register.register
class TemporalConvLayer(HipModule):
    def __init__(self, in_dim: int, out_dim: Optional[int] = None, dropout: float = 0.0,
                 keep_content_condition: bool = False, femb_channels: Optional[int] = None,
                 need_temporal_weight: bool = True):
        super().__init__()
        out_dim = out_dim or in_dim
        if keep_content_condition or not need_temporal_weight or out_dim != in_dim:
            raise NotImplementedError("TemporalConvLayer: only the shipped configuration (keep_content_condition=False, "
                                      "need_temporal_weight=True, out_dim == in_dim)")
        self.in_dim, self.out_dim = in_dim, out_dim
        # same Sequential indices as the reference so state_dict keys match (conv1.0/conv1.2, convN.0/convN.3)
        self.conv1 = nn.Sequential(nn.GroupNorm(32, in_dim), nn.SiLU(), nn.Conv3d(in_dim, out_dim, (3, 1, 1), padding=(1, 0, 0)))
        self.conv2 = nn.Sequential(nn.GroupNorm(32, out_dim), nn.SiLU(), nn.Dropout(dropout),
                                   nn.Conv3d(out_dim, in_dim, (3, 1, 1), padding=(1, 0, 0)))
        self.conv3 = nn.Sequential(nn.GroupNorm(32, out_dim), nn.SiLU(), nn.Dropout(dropout),
                                   nn.Conv3d(out_dim, in_dim, (3, 1, 1), padding=(1, 0, 0)))
        self.conv4 = nn.Sequential(nn.GroupNorm(32, out_dim), nn.SiLU(), nn.Dropout(dropout),
                                   nn.Conv3d(out_dim, in_dim, (3, 1, 1), padding=(1, 0, 0)))
        self.temporal_weight = nn.Parameter(torch.tensor([1e-5]))
        nn.init.zeros_(self.conv4[-1].weight)  # resnet.py:91-92
        nn.init.zeros_(self.conv4[-1].bias)
        self.skip_temporal_layers = False

    def alpha(self) -> torch.Tensor:
        return self.packed("alpha", lambda: self.temporal_weight.detach().float().reshape(1).contiguous())

    def hip_forward(self, x: torch.Tensor, ctx: Ctx, geo: Geo) -> torch.Tensor:
        if self.skip_temporal_layers or ctx.skip_temporal:
            return x
        h = x
        seqs = (self.conv1, self.conv2, self.conv3, self.conv4)
        for i, seq in enumerate(seqs):
            gn, conv = seq[0], seq[-1]
            h = ops.groupnorm(h, w16(gn.weight), w16(gn.bias), geo.b, geo.t * geo.hw, eps=gn.eps, silu=True, groups=gn.num_groups)
            w = self.packed(f"conv{i + 1}", lambda conv=conv: ops.pack_conv_weight(conv.weight.detach()))
            last = i == len(seqs) - 1
            h = ops.tconv3(h, w, geo.b, geo.t, geo.hw, bias=w16(conv.bias), residual=x if last else None,
                           alpha=self.alpha() if last else None, carry=last)
        return h
