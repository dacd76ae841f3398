"""Attention processors of the reference's registry, by name (musev/models/attention_processor.py:153-554), and
ReferEmbFuseAttention (:558-750).

In the reference a processor object carries the attention *algorithm* and is swapped into diffusers' Attention by
``hack_t2i_sd_layer_attn_with_ip`` (unet_3d_condition.py:116-137).  Here the algorithms are fixed HIP kernels; the
processor classes survive as registry names that select which key/value segments a block feeds to
``ops.attention`` (see BasicTransformerBlock.hip_forward), so the same constructor strings keep working."""
from __future__ import annotations

from typing import Optional

import torch
from torch import nn

from .. import ops
from . import Model_Register
from .layers import HipModule, IPAttention, lin_b, lin_w
from .runtime import Ctx, Geo, SourceCache


@Model_Register.register
class BaseIPAttnProcessor(nn.Module):
    """marker base class (attention_processor.py:153-159)"""
    mode = "plain"


@Model_Register.register
class T2IReferencenetIPAdapterXFormersAttnProcessor(BaseIPAttnProcessor):
    """text cross-attention + IP-Adapter image-prompt attention sharing the query (attention_processor.py:162-359)"""
    mode = "cross_ip"


@Model_Register.register
class NonParamT2ISelfReferenceXFormersAttnProcessor(BaseIPAttnProcessor):
    """reference-only self-attention: K/V = [self | vision-condition frame | referencenet tokens] (:363-546)"""
    mode = "self_reference"


@Model_Register.register
class NonParamReferenceIPXFormersAttnProcessor(NonParamT2ISelfReferenceXFormersAttnProcessor):
    """alias used by the shipped flavours (:549-554)"""


class ReferEmbFuseAttention(IPAttention):
    """Fuses one ReferenceNet feature map into the UNet latents with attention over [ref tokens | own tokens] and a
    residual connection (attention_processor.py:558-750).  The reference zero-initialises to_out (:626-627)."""

    def __init__(self, query_dim: int, heads: int = 8, dim_head: int = 64, **_unused):
        super().__init__(query_dim=query_dim, cross_attention_dim=None, heads=heads, dim_head=dim_head, bias=False)
        self.processor = None
        nn.init.zeros_(self.to_out[0].weight)
        nn.init.zeros_(self.to_out[0].bias)

    def ref_kv(self, ref: torch.Tensor) -> tuple:
        """K/V projection of the reference tokens, cached across denoise steps (the features are computed once per
        pipeline call, pipeline_controlnet.py:1883-1899).  ref: [B, C, t2, h2, w2]."""
        cache = self._cache().setdefault("ref_kv", SourceCache())
        kv = cache.get(ref, lambda r: ops.gemm(ops.bcthw_to_bthwc(r), self.w_kv()))  # rows [(b t2 h2 w2), C] -> [., 2C]
        return kv, ref.shape[2] * ref.shape[3] * ref.shape[4]

    def hip_forward(self, x: torch.Tensor, ref: torch.Tensor, geo: Geo) -> torch.Tensor:
        c = self.heads * self.dim_head
        if ref.shape[0] != geo.b or ref.shape[1] != c:
            raise ValueError(f"ReferEmbFuseAttention: refer emb {tuple(ref.shape)} does not match b={geo.b}, c={c}")
        qkv = ops.gemm(x, self.w_qkv())
        kv_ref, ref_len = self.ref_kv(ref)
        segs = [(kv_ref[:, :c], kv_ref[:, c:], ref_len, geo.t, 1, 0),          # ref tokens of batch item n // T
                (qkv[:, c:2 * c], qkv[:, 2 * c:], geo.hw, 1, 1, 0)]            # own tokens
        a = ops.attention(qkv[:, :c], segs, geo.n, geo.hw, self.heads, self.dim_head, self.scale)
        return self.project_out(a, residual=x)
