"""HIP-backed building blocks that the reference takes from the (un-vendored) diffusers fork: ResnetBlock2D,
Downsample2D / Upsample2D, TimestepEmbedding, FeedForward(GEGLU), and the attention module (IPAttention,
musev/models/attention_processor.py:54-150).

Every class keeps torch-native parameters under the reference's attribute names (so real checkpoints load and LoRA
merges can edit ``.weight.data`` in place) and adds a ``hip_forward`` that consumes / produces fp16
channels-last row matrices through musev_amd.ops.  Weights are re-packed lazily into the kernel layouts (fp16,
conv taps-major, fused QKV, GEGLU-interleaved) and re-packed again when the owning model notices a parameter version
bump."""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import torch
from torch import nn

from .. import ops
from .runtime import Ctx, Geo

_PACK_EPOCH = [0]


def bump_pack_epoch() -> None:
    _PACK_EPOCH[0] += 1


class HipModule(nn.Module):
    """nn.Module with a cache of packed (kernel-layout, fp16) weights."""

    # the top-level forwards refuse non-HIP tensors (there is no CPU path); the CPU test suite switches this off on an
    # instance only together with tests/emu_ops.py, to exercise the host-side wiring against the oracle
    _device_check = True

    def _cache(self) -> Dict[str, object]:
        if getattr(self, "_pk_epoch", -1) != _PACK_EPOCH[0]:
            object.__setattr__(self, "_pk", {})
            object.__setattr__(self, "_pk_epoch", _PACK_EPOCH[0])
        return self._pk

    def packed(self, name: str, builder):
        c = self._cache()
        if name not in c:
            c[name] = builder()
        return c[name]


def w16(p: torch.Tensor) -> torch.Tensor:
    """fp16 contiguous view/copy of a parameter (no copy when the model already is fp16)."""
    d = p.detach()
    if d.dtype == torch.float16 and d.is_contiguous():
        return d
    return d.to(torch.float16).contiguous()


def lin_w(m: nn.Module) -> torch.Tensor:
    """Linear [O, I] or 1x1 Conv2d [O, I, 1, 1] weight as a [O, I] fp16 matrix."""
    w = w16(m.weight)
    return w.view(w.shape[0], -1)


def lin_b(m: nn.Module) -> Optional[torch.Tensor]:
    return None if m.bias is None else w16(m.bias)


def ln_linear(mod: HipModule, key: str, x: torch.Tensor, norm: nn.LayerNorm, w_builder, bias_builder=None, *, geglu: bool = False,
              residual: Optional[torch.Tensor] = None) -> torch.Tensor:
    """LayerNorm(x) @ W.T (+ bias) -- nn.LayerNorm followed by a Linear (norm1 -> to_q/k/v, norm2 -> to_q, norm3 -> ff.net.0.proj;
    musev/models/attention.py:293-308,345-362,398-429).  Where the projection runs as one K slice the normalisation is folded
    into it (ops.gemm(ln=): no normalised tensor, no LayerNorm launch); elsewhere mv_layernorm_f16 + the plain projection.
    ``w_builder()`` -> packed [N, K] fp16 weight, ``bias_builder()`` -> [N] fp16 bias or None (both cached on ``mod``)."""
    w = mod.packed(key, w_builder)
    bias = mod.packed(key + "/bias", bias_builder) if bias_builder is not None else None
    M, K = x.shape
    if ops.ln_fold_applies(M, w.shape[0], K, geglu):
        wf, cs, cb = mod.packed(key + "/ln", lambda: ops.fold_layernorm(w, bias, w16(norm.weight), w16(norm.bias)))
        return ops.gemm(x, wf, ln=(cs, cb, norm.eps), geglu=geglu, residual=residual)
    h = ops.layernorm(x, w16(norm.weight), w16(norm.bias), norm.eps)
    return ops.gemm(h, w, bias=bias, geglu=geglu, residual=residual)


class TimestepEmbedding(HipModule):
    """diffusers TimestepEmbedding(in, dim, act_fn="silu"): linear_2(SiLU(linear_1(x)))."""

    def __init__(self, in_channels: int, time_embed_dim: int):
        super().__init__()
        self.linear_1 = nn.Linear(in_channels, time_embed_dim)
        self.linear_2 = nn.Linear(time_embed_dim, time_embed_dim)

    def hip_forward(self, x: torch.Tensor, final_silu: bool) -> torch.Tensor:
        h = ops.gemm(x, lin_w(self.linear_1), bias=lin_b(self.linear_1), act=ops.MV_ACT_SILU)
        return ops.gemm(h, lin_w(self.linear_2), bias=lin_b(self.linear_2),
                        act=ops.MV_ACT_SILU if final_silu else ops.MV_ACT_NONE)


class ResnetBlock2D(HipModule):
    """diffusers ResnetBlock2D(time_embedding_norm="default", pre_norm, output_scale_factor=1), applied per frame.
    Constructed by the reference at musev/models/unet_3d_blocks.py:272-285,486-499,811-824,1035-1048,1286-1299."""

    def __init__(self, in_channels: int, out_channels: int, temb_channels: int, eps: float = 1e-5, groups: int = 32,
                 skip_time_act: bool = False):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.norm1 = nn.GroupNorm(groups, in_channels, eps=eps)
        self.conv1 = nn.Conv2d(in_channels, out_channels, 3, padding=1)
        self.time_emb_proj = nn.Linear(temb_channels, out_channels)
        self.norm2 = nn.GroupNorm(groups, out_channels, eps=eps)
        self.conv2 = nn.Conv2d(out_channels, out_channels, 3, padding=1)
        self.conv_shortcut = nn.Conv2d(in_channels, out_channels, 1) if in_channels != out_channels else None
        self.skip_time_act = skip_time_act

    def hip_forward(self, x: torch.Tensor, x2: Optional[torch.Tensor], ctx: Ctx, geo: Geo) -> torch.Tensor:
        """x (+ x2: the skip tensor the reference concatenates with torch.cat, unet_3d_blocks.py:1130,1342)."""
        g = self.norm1.num_groups
        h = ops.groupnorm(x, w16(self.norm1.weight), w16(self.norm1.bias), geo.n, geo.hw, eps=self.norm1.eps, silu=True,
                          x2=x2, groups=g)
        tproj = ctx.proj_for(self)  # [N, Cout] column slice of the batched embedding projection
        if tproj is None:
            tproj = ops.gemm(ctx.temb_act, lin_w(self.time_emb_proj), bias=lin_b(self.time_emb_proj))
        w1 = self.packed("conv1", lambda: ops.pack_conv_weight(self.conv1.weight.detach()))
        h = ops.conv3x3(h, w1, geo.n, geo.h, geo.w, bias=w16(self.conv1.bias), rowbias=tproj, rows_per_group=geo.hw)
        h = ops.groupnorm(h, w16(self.norm2.weight), w16(self.norm2.bias), geo.n, geo.hw, eps=self.norm2.eps, silu=True,
                          groups=g)
        if self.conv_shortcut is not None:
            sc = ops.gemm(x, lin_w(self.conv_shortcut), a2=x2, bias=lin_b(self.conv_shortcut))
        else:
            if x2 is not None:
                raise ValueError("ResnetBlock2D: concatenated input needs a conv_shortcut")
            sc = x
        w2 = self.packed("conv2", lambda: ops.pack_conv_weight(self.conv2.weight.detach()))
        return ops.conv3x3(h, w2, geo.n, geo.h, geo.w, bias=w16(self.conv2.bias), residual=sc, carry=True)


class Downsample2D(HipModule):
    """diffusers Downsample2D(use_conv=True, padding=1, name="op"): conv3x3 stride 2 (parameter name ``conv``)."""

    def __init__(self, channels: int, padding: int = 1):
        super().__init__()
        if padding != 1:
            raise ValueError("Downsample2D: only downsample_padding=1 is supported (SD-1.5 config)")
        self.conv = nn.Conv2d(channels, channels, 3, stride=2, padding=1)

    def hip_forward(self, x: torch.Tensor, geo: Geo) -> torch.Tensor:
        w = self.packed("conv", lambda: ops.pack_conv_weight(self.conv.weight.detach()))
        return ops.conv3x3(x, w, geo.n, geo.h, geo.w, stride=2, bias=w16(self.conv.bias))


class Upsample2D(HipModule):
    """diffusers Upsample2D(use_conv=True): nearest x2 (fused into the conv's gather) + conv3x3."""

    def __init__(self, channels: int):
        super().__init__()
        self.conv = nn.Conv2d(channels, channels, 3, padding=1)

    def hip_forward(self, x: torch.Tensor, geo: Geo, output_size=None):
        """-> (rows, geometry of the result).  ``output_size`` = (h, w): the explicit size the reference forwards for latents that are not
        multiples of 2^(number of upsamplers) (``upsampler(hidden_states, upsample_size)``, unet_3d_blocks.py:1235,1397 -> diffusers
        Upsample2D: F.interpolate(size=output_size, mode="nearest")); an exact x 2 stays fused into the convolution's gather."""
        w = self.packed("conv", lambda: ops.pack_conv_weight(self.conv.weight.detach()))
        if output_size is None or tuple(int(v) for v in output_size) == (2 * geo.h, 2 * geo.w):
            return ops.conv3x3(x, w, geo.n, geo.h, geo.w, upsample=True, bias=w16(self.conv.bias)), geo.up()
        ho, wo = (int(v) for v in output_size)
        up = ops.upsample_nearest(x, geo.n, geo.h, geo.w, ho, wo)
        return ops.conv3x3(up, w, geo.n, ho, wo, bias=w16(self.conv.bias)), Geo(geo.b, geo.t, ho, wo)


class GEGLU(nn.Module):
    def __init__(self, dim_in: int, dim_out: int):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out * 2)


class FeedForward(HipModule):
    """diffusers FeedForward(dim, mult=4, activation_fn="geglu"): net = [GEGLU, Dropout, Linear]."""

    def __init__(self, dim: int, mult: int = 4):
        super().__init__()
        inner = dim * mult
        self.net = nn.ModuleList([GEGLU(dim, inner), nn.Dropout(0.0), nn.Linear(inner, dim)])

    def _geglu_packed(self):
        return self.packed("geglu", lambda: ops.pack_geglu(lin_w(self.net[0].proj), lin_b(self.net[0].proj)))

    def hip_forward(self, x: torch.Tensor, residual: torch.Tensor, norm: Optional[nn.LayerNorm] = None) -> torch.Tensor:
        """FF(norm(x)) + residual when ``norm`` is given (the block's norm3, folded into the GEGLU projection), else FF(x) + residual"""
        if norm is not None and ops.ffn_fused_applies(x.shape[1], self.net[2].in_features):
            # level 0: norm3 -> GEGLU projection -> output projection -> + residual as ONE launch (csrc/ffn.hip)
            wp, bp = self._geglu_packed()
            return ops.ffn_geglu(x, w16(norm.weight), w16(norm.bias), norm.eps, wp, bp, lin_w(self.net[2]), lin_b(self.net[2]), residual)
        if norm is not None:
            h = ln_linear(self, "geglu_w", x, norm, lambda: self._geglu_packed()[0], lambda: self._geglu_packed()[1], geglu=True)
        else:
            wp, bp = self._geglu_packed()
            h = ops.gemm(x, wp, bias=bp, geglu=True)  # value * gelu(gate) applied in the GEMM epilogue (K7)
        return ops.gemm(h, lin_w(self.net[2]), bias=lin_b(self.net[2]), residual=residual)


class IPAttention(HipModule):
    """Parameter container with the reference's attribute names (diffusers Attention + IPAttention extras,
    musev/models/attention_processor.py:54-150): bias-free to_q/to_k/to_v, biased to_out[0], optional
    to_k_ip / to_v_ip.  The attention math itself lives in the owning block's hip_forward."""

    def __init__(self, query_dim: int, cross_attention_dim: Optional[int] = None, heads: int = 8, dim_head: int = 64,
                 bias: bool = False, cross_attn_temporal_cond: bool = False, ip_adapter_dim: Optional[int] = None,
                 need_t2i_facein: bool = False, need_t2i_ip_adapter_face: bool = False, ip_adapter_face_dim: Optional[int] = None,
                 processor=None):
        super().__init__()
        if need_t2i_facein:
            raise NotImplementedError("facein")  # attention_processor.py:123-124
        inner = heads * dim_head
        kv = cross_attention_dim if cross_attention_dim is not None else query_dim
        self.heads, self.dim_head, self.scale = heads, dim_head, dim_head ** -0.5
        self.to_q = nn.Linear(query_dim, inner, bias=bias)
        self.to_k = nn.Linear(kv, inner, bias=bias)
        self.to_v = nn.Linear(kv, inner, bias=bias)
        self.to_out = nn.ModuleList([nn.Linear(inner, query_dim), nn.Dropout(0.0)])
        self.cross_attn_temporal_cond = cross_attn_temporal_cond
        if cross_attn_temporal_cond:
            self.to_k_ip = nn.Linear(ip_adapter_dim, query_dim, bias=False)
            self.to_v_ip = nn.Linear(ip_adapter_dim, query_dim, bias=False)
        # IP-Adapter-FaceID: a second pair of image-prompt projections (attention_processor.py:127-135)
        self.need_t2i_ip_adapter_face = need_t2i_ip_adapter_face
        if need_t2i_ip_adapter_face:
            self.ip_adapter_face_to_k_ip = nn.Linear(ip_adapter_face_dim, query_dim, bias=False)
            self.ip_adapter_face_to_v_ip = nn.Linear(ip_adapter_face_dim, query_dim, bias=False)
        self.processor = processor

    def set_use_memory_efficient_attention_xformers(self, *args, **kwargs):  # pipeline compatibility no-op
        return None

    # ---- packed weights ----
    def build_qkv(self) -> torch.Tensor:
        return torch.cat([lin_w(self.to_q), lin_w(self.to_k), lin_w(self.to_v)], 0).contiguous()

    def w_qkv(self) -> torch.Tensor:
        return self.packed("qkv", self.build_qkv)

    def w_kv(self) -> torch.Tensor:
        return self.packed("kv", lambda: torch.cat([lin_w(self.to_k), lin_w(self.to_v)], 0).contiguous())

    def w_kv_ip(self) -> torch.Tensor:
        return self.packed("kv_ip", lambda: torch.cat([lin_w(self.to_k_ip), lin_w(self.to_v_ip)], 0).contiguous())

    def w_kv_face(self) -> torch.Tensor:
        return self.packed("kv_face", lambda: torch.cat([lin_w(self.ip_adapter_face_to_k_ip), lin_w(self.ip_adapter_face_to_v_ip)], 0).contiguous())

    def project_out(self, a: torch.Tensor, residual: torch.Tensor) -> torch.Tensor:
        return ops.gemm(a, lin_w(self.to_out[0]), bias=lin_b(self.to_out[0]), residual=residual)
