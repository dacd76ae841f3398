"""BasicTransformerBlock of the reference (musev/models/attention.py:52-431) on HIP kernels.

Spatial instance: attn1 = reference-only self-attention, attn2 = text cross-attention (+ IP-Adapter), GEGLU FF.
Temporal instance (double_self_attention=True): attn1 and attn2 are self-attention over the T frames of each pixel.
Not restated: the classifier-free-guidance recompute at attention.py:319-334 -- its result is overwritten before it
is read (dead code)."""
from __future__ import annotations

from typing import Optional

import torch
from torch import nn

from .. import ops
from .layers import FeedForward, HipModule, IPAttention, lin_b, lin_w, ln_linear, w16
from .runtime import Ctx, Geo, SourceCache


class BasicTransformerBlock(HipModule):
    def __init__(self, dim: int, num_attention_heads: int, attention_head_dim: int, cross_attention_dim: Optional[int] = None,
                 double_self_attention: bool = False, only_cross_attention: bool = False, attention_bias: bool = False,
                 cross_attn_temporal_cond: bool = False, ip_adapter_cross_attn: bool = False,
                 need_t2i_facein: bool = False, need_t2i_ip_adapter_face: bool = False, processor=None, **_unused):
        super().__init__()
        if only_cross_attention:
            raise NotImplementedError("only_cross_attention is not used by any shipped MuseV flavour")
        if not only_cross_attention and double_self_attention:
            cross_attention_dim = None  # attention.py:80-81
        self.double_self_attention = double_self_attention
        self.norm1 = nn.LayerNorm(dim)
        self.attn1 = IPAttention(dim, None, num_attention_heads, attention_head_dim, bias=attention_bias,
                                 cross_attn_temporal_cond=cross_attn_temporal_cond, ip_adapter_dim=attention_head_dim,
                                 processor=processor)
        self.norm2 = nn.LayerNorm(dim)
        self.attn2 = IPAttention(dim, cross_attention_dim if not double_self_attention else None, num_attention_heads,
                                 attention_head_dim, bias=attention_bias, cross_attn_temporal_cond=ip_adapter_cross_attn,
                                 ip_adapter_dim=cross_attention_dim if not double_self_attention else attention_head_dim,
                                 need_t2i_facein=need_t2i_facein, need_t2i_ip_adapter_face=need_t2i_ip_adapter_face,
                                 ip_adapter_face_dim=cross_attention_dim if not double_self_attention else attention_head_dim,
                                 processor=processor)
        self.norm3 = nn.LayerNorm(dim)
        self.ff = FeedForward(dim)
        self.heads, self.dim_head = num_attention_heads, attention_head_dim

    # ---- spatial: rows = (frame n, pixel p), sequences = the HW pixels of one frame ----
    def hip_forward_spatial(self, x: torch.Tensor, ctx: Ctx, geo: Geo, reference_only: bool, use_ip: bool) -> torch.Tensor:
        c, h, d = self.heads * self.dim_head, self.heads, self.dim_head
        if ctx.refer_self is not None or ctx.refer_self_write is not None:
            ctx.split()  # refer_self_attn_emb is handed over (or collected) per CFG half
        x_in = x
        x, q = ctx.shared((id(self), "attn1+q"), lambda: self._self_attention_and_query(x_in, ctx, geo, reference_only, use_ip))
        ctx.split()  # the text (and the image-prompt tokens) differ between the CFG halves from here on
        return self._cross_attention_and_ff(x, q, ctx, geo, use_ip)

    def _self_attention_and_query(self, x: torch.Tensor, ctx: Ctx, geo: Geo, reference_only: bool, use_ip: bool = False):
        """norm1 -> reference-only self-attention -> + x, then norm2 -> to_q of the cross-attention: the part of the block that
        does not see the text (shared by the CFG halves in the first block of the network, runtime.PrefixMemo)"""
        c, h, d = self.heads * self.dim_head, self.heads, self.dim_head
        a1 = self.attn1
        if ctx.refer_self_write is not None:
            # attention.py:240-259 + transformer_2d.py:340-359 ("write"): the self-attention's INPUT (norm1's output) of this block goes
            # into the caller's list as [(b t), c, h, w] -- the one place the normalised rows are materialised (elsewhere norm1 is
            # folded into the q / k / v projection)
            nrm = ops.layernorm(x, self.norm1.weight.detach().to(torch.float16), self.norm1.bias.detach().to(torch.float16), self.norm1.eps)
            ctx.refer_self_write[self.spatial_self_attn_idx] = nrm.view(geo.n, geo.h, geo.w, c).permute(0, 3, 1, 2).contiguous()
        qkv = ln_linear(a1, "qkv", x, self.norm1, a1.build_qkv)  # norm1 folded into the fused q/k/v projection where it pays
        k, v = qkv[:, c:2 * c], qkv[:, 2 * c:]
        segs = [(k, v, geo.hw, 1, 1, 0)]
        if reference_only and ctx.vis_idx is not None and geo.t > 1:
            # attention_processor.py:431-468: append the vision-condition frame(s) of the same batch item; their K/V
            # projections are the rows already computed for that frame
            for ci in ctx.vis_idx:
                segs.append((k, v, geo.hw, geo.t, geo.t, int(ci)))
        if reference_only and ctx.refer_self is not None:
            # attention_processor.py:476-491: the tokens of refer_self_attn_emb[block] ("b c t h w -> b 1 (t h w) c", repeated over
            # the frames) join the keys / values; they are constant over the denoise loop -> projected once per source tensor
            ref = ctx.refer_self[self.spatial_self_attn_idx]
            if ref.shape[0] != geo.b or ref.shape[1] != c:
                raise ValueError(f"refer_self_attn_emb[{self.spatial_self_attn_idx}] must be [b = {geo.b}, c = {c}, t, h, w]")
            n_ref = ref.shape[2] * ref.shape[3] * ref.shape[4]
            rkv = a1._cache().setdefault("refer_self_kv", SourceCache()).get(
                ref, lambda src: ops.gemm(ops.bcthw_to_bthwc(src), a1.w_kv()))
            segs.append((rkv[:, :c], rkv[:, c:], n_ref, geo.t, 1, 0))
        att = ops.attention(qkv[:, :c], segs, geo.n, geo.hw, h, d, a1.scale)
        x = a1.project_out(att, residual=x)
        a2 = self.attn2
        if self._xab_applies(ctx, geo, use_ip):
            return x, None   # level 0, text only: norm2 -> to_q -> attention -> to_out + x is one launch (_cross_attention_and_ff)
        return x, ln_linear(a2, "q", x, self.norm2, lambda: lin_w(a2.to_q).contiguous())

    def _xab_applies(self, ctx: Ctx, geo: Geo, use_ip: bool) -> bool:
        """the text cross-attention sub-block as one launch (ops.xattn_block): one softmax group of <= 80 keys at C = 320 = 8 x 40"""
        a2 = self.attn2
        if a2.to_q.bias is not None or (use_ip and a2.cross_attn_temporal_cond and ctx.clip is not None and ctx.ip_scale > 0) or \
                (a2.need_t2i_ip_adapter_face and ctx.face is not None and ctx.face_scale > 0):
            return False
        return ops.xab_fused_applies(self.heads * self.dim_head, self.heads, self.dim_head, int(ctx.text_len), geo.t * geo.hw)

    def _cross_attention_and_ff(self, x: torch.Tensor, q: torch.Tensor, ctx: Ctx, geo: Geo, use_ip: bool) -> torch.Tensor:
        c, h, d = self.heads * self.dim_head, self.heads, self.dim_head
        a2 = self.attn2
        cache = a2._cache()
        # K/V of the prompt: constant over the denoise loop -> projected once per (prompt tensor, weights)
        tkv = cache.setdefault("text_kv", SourceCache()).get(ctx.text_src, lambda _s: ops.gemm(ctx.text, a2.w_kv()))
        # text cross-attention + ip_adapter_scale * image-prompt attention (+ face_scale * FaceID attention, attention_processor.py:
        # 258-300, 308-338): every term its own softmax of the same queries.  Head dims 40 / 80: ONE launch with softmax groups (the
        # queries read once, the output written once); d = 160: one launch per term, accumulated into the output.
        if q is None:
            x = ops.xattn_block(x, w16(self.norm2.weight), w16(self.norm2.bias), self.norm2.eps,
                                a2.packed("xab_q", lambda: ops.pack_xab_q(lin_w(a2.to_q), h, d)), tkv[:, :c], tkv[:, c:], int(ctx.text_len),
                                geo.t * geo.hw, a2.packed("tsa_out", lambda: ops.pack_tsa_out(lin_w(a2.to_out[0]), h, d)), lin_b(a2.to_out[0]),
                                h, d, a2.scale)
            return self.ff.hip_forward(x, residual=x, norm=self.norm3)
        terms = [((tkv[:, :c], tkv[:, c:], ctx.text_len, geo.t, 1, 0), 1.0)]
        if use_ip and a2.cross_attn_temporal_cond and ctx.clip is not None and ctx.ip_scale > 0:
            ikv = cache.setdefault("clip_kv", SourceCache()).get(ctx.clip_src, lambda _s: ops.gemm(ctx.clip, a2.w_kv_ip()))
            terms.append(((ikv[:, :c], ikv[:, c:], ctx.clip_len, geo.t, 1, 0), float(ctx.ip_scale)))
        if a2.need_t2i_ip_adapter_face and ctx.face is not None and ctx.face_scale > 0:
            fkv = cache.setdefault("face_kv", SourceCache()).get(ctx.face_src, lambda _s: ops.gemm(ctx.face, a2.w_kv_face()))
            terms.append(((fkv[:, :c], fkv[:, c:], ctx.face_len, geo.t, 1, 0), float(ctx.face_scale)))
        if len(terms) == 1 or (d in (40, 80) and getattr(ops, "ATTN_GROUPS", True)):
            att = ops.attention(q, [sg for sg, _w in terms], geo.n, geo.hw, h, d, a2.scale,
                                group_scales=[w for _sg, w in terms] if len(terms) > 1 else None)
        else:
            att = ops.attention(q, [terms[0][0]], geo.n, geo.hw, h, d, a2.scale)
            for sg, w in terms[1:]:
                ops.attention(q, [sg], geo.n, geo.hw, h, d, a2.scale, out=att, accumulate=True, out_scale=w)
        x = a2.project_out(att, residual=x)
        return self.ff.hip_forward(x, residual=x, norm=self.norm3)

    # ---- temporal: rows stay in (b, t, p) order; sequences = the T frames of one pixel ----
    def hip_forward_temporal(self, x: torch.Tensor, geo: Geo) -> torch.Tensor:
        c, h, d = self.heads * self.dim_head, self.heads, self.dim_head
        for norm, attn in ((self.norm1, self.attn1), (self.norm2, self.attn2)):
            if ops.tsa_fused_applies(c, h, d, geo.t, geo.hw) and attn.to_q.bias is None:
                # level 0: LayerNorm -> q / k / v -> T x T attention -> to_out + x in one launch (mv_temporal_attn_block_f16)
                x = ops.temporal_attn_block(
                    x, w16(norm.weight), w16(norm.bias), norm.eps,
                    attn.packed("tsa_qkv", lambda a=attn: ops.pack_tsa_qkv(lin_w(a.to_q), lin_w(a.to_k), lin_w(a.to_v), h, d)),
                    attn.packed("tsa_out", lambda a=attn: ops.pack_tsa_out(lin_w(a.to_out[0]), h, d)),
                    lin_b(attn.to_out[0]), geo.b, geo.t, geo.hw, h, d, attn.scale)
                continue
            qkv = ln_linear(attn, "qkv", x, norm, attn.build_qkv)
            att = ops.temporal_attention(qkv[:, :c], qkv[:, c:2 * c], qkv[:, 2 * c:], geo.b, geo.t, geo.hw, h, d, attn.scale)
            x = attn.project_out(att, residual=x)
        return self.ff.hip_forward(x, residual=x, norm=self.norm3)
