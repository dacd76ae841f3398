"""Transformer2DModel (musev/models/transformer_2d.py:55-445, continuous-input branch :257-271,365-389):
GroupNorm(32, eps 1e-6) -> 1x1 conv -> BasicTransformerBlock -> 1x1 conv -> + residual.  In the channels-last row
layout the 1x1 convolutions are plain GEMMs and the "b c h w -> b (h w) c" permutes disappear."""
from __future__ import annotations

from typing import Optional

import torch
from torch import nn

from .. import ops
from .attention import BasicTransformerBlock
from .layers import HipModule, lin_b, lin_w, w16
from .runtime import Ctx, Geo


class Transformer2DModel(HipModule):
    def __init__(self, num_attention_heads: int = 16, attention_head_dim: int = 88, in_channels: Optional[int] = None,
                 num_layers: int = 1, norm_num_groups: int = 32, cross_attention_dim: Optional[int] = None,
                 use_linear_projection: bool = False, cross_attn_temporal_cond: bool = False,
                 ip_adapter_cross_attn: bool = False, need_t2i_facein: bool = False,
                 need_t2i_ip_adapter_face: bool = False, **_unused):
        super().__init__()
        if use_linear_projection:
            raise NotImplementedError("use_linear_projection=True is not part of the SD-1.5 MuseV configs")
        inner = num_attention_heads * attention_head_dim
        if inner != in_channels:
            raise ValueError("Transformer2DModel: heads * head_dim must equal in_channels")
        self.norm = nn.GroupNorm(norm_num_groups, in_channels, eps=1e-6, affine=True)
        self.proj_in = nn.Conv2d(in_channels, inner, 1)
        self.transformer_blocks = nn.ModuleList([
            BasicTransformerBlock(inner, num_attention_heads, attention_head_dim, cross_attention_dim=cross_attention_dim,
                                  cross_attn_temporal_cond=cross_attn_temporal_cond,
                                  ip_adapter_cross_attn=ip_adapter_cross_attn, need_t2i_facein=need_t2i_facein,
                                  need_t2i_ip_adapter_face=need_t2i_ip_adapter_face)
            for _ in range(num_layers)])
        self.proj_out = nn.Conv2d(inner, in_channels, 1)
        self.ip_adapter_cross_attn = ip_adapter_cross_attn
        self.reference_only = True  # set by UNet3DConditionModel from t2i_ip_adapter_attn_processor

    def hip_forward(self, x: torch.Tensor, ctx: Ctx, geo: Geo) -> torch.Tensor:
        def stem():
            # norm -> proj_in (transformer_2d.py:260-271, 365-368): where the per-frame weight copies are small next to the tensor the
            # GroupNorm is folded into them and its apply pass never runs (ops.groupnorm_fold_linear)
            h0 = ops.groupnorm_fold_linear(x, w16(self.norm.weight), w16(self.norm.bias), geo.n, geo.hw, eps=self.norm.eps,
                                           groups=self.norm.num_groups, w=lin_w(self.proj_in), bias=lin_b(self.proj_in))
            if h0 is not None:
                return h0
            g = ops.groupnorm(x, w16(self.norm.weight), w16(self.norm.bias), geo.n, geo.hw, eps=self.norm.eps, silu=False,
                              groups=self.norm.num_groups)
            return ops.gemm(g, lin_w(self.proj_in), bias=lin_b(self.proj_in))
        h = ctx.shared((id(self), "stem"), stem)
        for blk in self.transformer_blocks:
            h = blk.hip_forward_spatial(h, ctx, geo, self.reference_only, self.ip_adapter_cross_attn)
        return ops.gemm(h, lin_w(self.proj_out), bias=lin_b(self.proj_out), residual=x, colstats=True, carry=True)
