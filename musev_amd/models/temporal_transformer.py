"""TransformerTemporalModel (musev/models/temporal_transformer.py:57-308): GroupNorm over (C/32, T, H, W) -> Linear ->
+ frame-embedding projection -> BasicTransformerBlock (two self-attentions over T) -> Linear (zero-init in the
reference) -> residual + |temporal_weight| * h.  Rows never leave (b, t, p) order: the reference's
"(b t) c h w -> b c t h w -> (b h w) t c" permute copies (:234-241, 277-279) are index arithmetic here."""
from __future__ import annotations

from typing import Optional

import torch
from torch import nn

from .. import ops
from . import Model_Register
from .attention import BasicTransformerBlock
from .layers import HipModule, lin_b, lin_w, w16
from .runtime import Ctx, Geo


@Model_Register.register
class TransformerTemporalModel(HipModule):
    def __init__(self, num_attention_heads: int = 16, attention_head_dim: int = 88, in_channels: Optional[int] = None,
                 num_layers: int = 1, femb_channels: Optional[int] = None, norm_num_groups: int = 32,
                 cross_attention_dim: Optional[int] = None, double_self_attention: bool = True,
                 need_temporal_weight: bool = True, need_spatial_position_emb: bool = False, **_unused):
        super().__init__()
        if need_spatial_position_emb:
            raise NotImplementedError("need_spatial_position_emb=True is unused by all shipped configs (unet_loader.py:236)")
        if not double_self_attention or not need_temporal_weight:
            raise NotImplementedError("only double_self_attention=True / need_temporal_weight=True (the shipped configuration)")
        inner = num_attention_heads * attention_head_dim
        self.in_channels = in_channels
        self.norm = nn.GroupNorm(norm_num_groups, in_channels, eps=1e-6, affine=True)
        self.proj_in = nn.Linear(in_channels, inner)
        self.frame_emb_proj = nn.Linear(femb_channels, inner)
        self.transformer_blocks = nn.ModuleList([
            BasicTransformerBlock(inner, num_attention_heads, attention_head_dim, cross_attention_dim=cross_attention_dim,
                                  double_self_attention=True) for _ in range(num_layers)])
        self.proj_out = nn.Linear(inner, in_channels)
        self.temporal_weight = nn.Parameter(torch.tensor([1e-5]))
        nn.init.zeros_(self.proj_out.weight)  # temporal_transformer.py:186-187
        nn.init.zeros_(self.proj_out.bias)
        self.skip_temporal_layers = False

    def alpha(self) -> torch.Tensor:
        return self.packed("alpha", lambda: self.temporal_weight.detach().float().reshape(1).contiguous())

    def hip_forward(self, x: torch.Tensor, ctx: Ctx, geo: Geo) -> torch.Tensor:
        if self.skip_temporal_layers or ctx.skip_temporal:
            return x
        fproj = ctx.proj_for(self)  # [B*T, C] column slice of the batched embedding projection
        if fproj is None:
            fproj = ops.gemm(ctx.femb_act, lin_w(self.frame_emb_proj), bias=lin_b(self.frame_emb_proj))
        # norm -> proj_in + frame embedding (temporal_transformer.py:239-251): the statistics span (C / 32, T, H, W) of a batch item, so the
        # folded form needs ONE scaled copy of proj_in's weights per item; the frame embedding rides in the folded row bias
        h = ops.groupnorm_fold_linear(x, w16(self.norm.weight), w16(self.norm.bias), geo.b, geo.t * geo.hw, eps=self.norm.eps,
                                      groups=self.norm.num_groups, w=lin_w(self.proj_in), bias=lin_b(self.proj_in),
                                      rowbias=fproj, rb_per_item=geo.t)
        if h is None:
            h = ops.groupnorm(x, w16(self.norm.weight), w16(self.norm.bias), geo.b, geo.t * geo.hw, eps=self.norm.eps,
                              silu=False, groups=self.norm.num_groups)
            h = ops.gemm(h, lin_w(self.proj_in), bias=lin_b(self.proj_in), rowbias=fproj, rows_per_group=geo.hw)
        for blk in self.transformer_blocks:
            h = blk.hip_forward_temporal(h, geo)
        return ops.gemm(h, lin_w(self.proj_out), bias=lin_b(self.proj_out), residual=x, alpha=self.alpha(), colstats=True, carry=True)
