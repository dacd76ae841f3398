"""VAE decoder (diffusers ``AutoencoderKL.decode``) and IP-Adapter image projection (``ImageProjModel``) on the HIP kernels of the
UNet path -- the two side models either side of the denoise loop.

Call sites in the reference:
  * ``MusevControlNetPipeline.decode_latents`` (musev/pipelines/pipeline_controlnet.py:233-238) -> diffusers
    ``StableDiffusionPipeline.decode_latents``: ``image = vae.decode(latents / scaling_factor); (image / 2 + 0.5).clamp(0, 1)``,
    called per temporal chunk of ``decoder_t_segment`` frames (:2157-2171).  96 frames x 512x512 = 1.27 TFLOP per frame: the
    largest cost after the loop.
  * ``ImageProjModel`` (ip_adapter package; built at musev/models/ip_adapter_loader.py:89-93, applied at
    pipeline_controlnet.py:736-774): CLIP image embedding [B, 1024] -> 4 tokens of width 768 = the UNet's ``vision_clip_emb``.
Both classes live in un-vendored packages: the modules here keep the upstream parameter names (so ``vae/diffusion_pytorch_model.*``
and the ``image_proj`` dict of an IP-Adapter checkpoint load), the semantics are those restated in oracle/vae.py (parity
unpinned, see its header).  Only the decoder half of the VAE is on this path (``encode`` -- condition images -- is not).

Kernels: every convolution is the implicit-GEMM kernel (3x3; nearest-x2 upsample fused into the gather; 1x1 shortcuts as LINEAR;
conv_in 4 -> 512 through the im2col + one-K-step form of the UNet's conv_in; conv_out 128 -> 3 through the direct small-Cout kernel),
GroupNorm(32, eps 1e-6)(+SiLU) is mv_groupnorm_f16.  The mid block's single 512-wide attention head over H*W tokens runs as
GEMM (Q K^T, scale folded into Q) -> mv_softmax_rows_f16 -> GEMM (P V, with V^T produced directly by a GEMM with swapped operands),
per frame: d = 512 does not fit the fused attention kernel's register tiling and the block runs once per decoded frame."""
from __future__ import annotations

from types import SimpleNamespace
from typing import Optional

import torch
from torch import nn

from .. import ops
from .layers import HipModule, Upsample2D, bump_pack_epoch, lin_b, lin_w, w16
from .runtime import Geo

__all__ = ["AutoencoderKL", "ImageProjModel"]


class VaeResnetBlock2D(HipModule):
    """diffusers ResnetBlock2D(temb_channels=None, eps=1e-6, groups=32, output_scale_factor=1)"""

    def __init__(self, in_channels: int, out_channels: int, groups: int = 32, eps: float = 1e-6):
        super().__init__()
        self.norm1 = nn.GroupNorm(groups, in_channels, eps=eps)
        self.conv1 = nn.Conv2d(in_channels, out_channels, 3, padding=1)
        self.norm2 = nn.GroupNorm(groups, out_channels, eps=eps)
        self.conv2 = nn.Conv2d(out_channels, out_channels, 3, padding=1)
        self.conv_shortcut = nn.Conv2d(in_channels, out_channels, 1) if in_channels != out_channels else None

    def hip_forward(self, x: torch.Tensor, geo: Geo) -> torch.Tensor:
        g = self.norm1.num_groups
        h = ops.groupnorm(x, w16(self.norm1.weight), w16(self.norm1.bias), geo.n, geo.hw, eps=self.norm1.eps, silu=True, groups=g)
        w1 = self.packed("conv1", lambda: ops.pack_conv_weight(self.conv1.weight.detach()))
        h = ops.conv3x3(h, w1, geo.n, geo.h, geo.w, bias=w16(self.conv1.bias))
        h = ops.groupnorm(h, w16(self.norm2.weight), w16(self.norm2.bias), geo.n, geo.hw, eps=self.norm2.eps, silu=True, groups=g)
        sc = x if self.conv_shortcut is None else ops.gemm(x, lin_w(self.conv_shortcut), bias=lin_b(self.conv_shortcut))
        w2 = self.packed("conv2", lambda: ops.pack_conv_weight(self.conv2.weight.detach()))
        return ops.conv3x3(h, w2, geo.n, geo.h, geo.w, bias=w16(self.conv2.bias), residual=sc)


class VaeAttention(HipModule):
    """diffusers Attention(C, heads=1, dim_head=C, norm_num_groups=32, eps=1e-6, residual_connection=True, bias=True)"""

    def __init__(self, channels: int, groups: int = 32, eps: float = 1e-6):
        super().__init__()
        self.group_norm = nn.GroupNorm(groups, channels, eps=eps)
        self.to_q = nn.Linear(channels, channels)
        self.to_k = nn.Linear(channels, channels)
        self.to_v = nn.Linear(channels, channels)
        self.to_out = nn.ModuleList([nn.Linear(channels, channels), nn.Dropout(0.0)])
        self.channels = channels

    def hip_forward(self, x: torch.Tensor, geo: Geo) -> torch.Tensor:
        c, hw = self.channels, geo.hw
        n = ops.groupnorm(x, w16(self.group_norm.weight), w16(self.group_norm.bias), geo.n, hw, eps=self.group_norm.eps, silu=False,
                          groups=self.group_norm.num_groups)
        scale = c ** -0.5

        def qk_pack():  # [to_q * scale | to_k] rows: one GEMM for both, the softmax scale folded into Q (keeps fp16 scores small)
            return (torch.cat([lin_w(self.to_q).float() * scale, lin_w(self.to_k).float()], 0).half().contiguous(),
                    torch.cat([lin_b(self.to_q).float() * scale, lin_b(self.to_k).float()], 0).half().contiguous())
        wqk, bqk = self.packed("qk", qk_pack)
        qk = ops.gemm(n, wqk, bias=bqk)                                     # [N*HW, 2C]
        out = torch.empty_like(x)
        wv = lin_w(self.to_v)
        for f in range(geo.n):                                              # once per decoded frame
            rows = slice(f * hw, (f + 1) * hw)
            s = ops.gemm(qk[rows, :c], qk[rows, c:].contiguous())          # S = (Q / sqrt(C)) K^T            [HW, HW]
            ops.softmax_rows_(s)
            vt = ops.gemm(wv, n[rows])                                      # V^T = W_v X^T (swapped operands)  [C, HW]
            # P V + b_v (softmax rows sum to 1, so the bias of to_v passes through the average unchanged)
            o = ops.gemm(s, vt, bias=lin_b(self.to_v))                      # [HW, C]
            ops.gemm(o, lin_w(self.to_out[0]), bias=lin_b(self.to_out[0]), residual=x[rows], out=out[rows])
        return out


class UNetMidBlock2D(nn.Module):
    def __init__(self, channels: int, groups: int):
        super().__init__()
        self.resnets = nn.ModuleList([VaeResnetBlock2D(channels, channels, groups), VaeResnetBlock2D(channels, channels, groups)])
        self.attentions = nn.ModuleList([VaeAttention(channels, groups)])

    def hip_forward(self, x, geo):
        x = self.resnets[0].hip_forward(x, geo)
        x = self.attentions[0].hip_forward(x, geo)
        return self.resnets[1].hip_forward(x, geo)


class UpDecoderBlock2D(nn.Module):
    def __init__(self, in_channels: int, out_channels: int, num_layers: int, groups: int, add_upsample: bool):
        super().__init__()
        self.resnets = nn.ModuleList([VaeResnetBlock2D(in_channels if j == 0 else out_channels, out_channels, groups)
                                      for j in range(num_layers)])
        self.upsamplers = nn.ModuleList([Upsample2D(out_channels)]) if add_upsample else None

    def hip_forward(self, x, geo):
        for r in self.resnets:
            x = r.hip_forward(x, geo)
        if self.upsamplers is not None:
            x, geo = self.upsamplers[0].hip_forward(x, geo)
        return x, geo


class Decoder(HipModule):
    def __init__(self, latent_channels: int, out_channels: int, block_out_channels, layers_per_block: int, groups: int):
        super().__init__()
        top = block_out_channels[-1]
        self.conv_in = nn.Conv2d(latent_channels, top, 3, padding=1)
        self.mid_block = UNetMidBlock2D(top, groups)
        rev = list(reversed(block_out_channels))
        blocks, out = [], rev[0]
        for i, c in enumerate(rev):
            prev, out = out, c
            blocks.append(UpDecoderBlock2D(prev, out, layers_per_block + 1, groups, add_upsample=i != len(rev) - 1))
        self.up_blocks = nn.ModuleList(blocks)
        self.conv_norm_out = nn.GroupNorm(groups, block_out_channels[0], eps=1e-6)
        self.conv_out = nn.Conv2d(block_out_channels[0], out_channels, 3, padding=1)


# legacy (diffusers < 0.17 "AttentionBlock") names of the mid-block attention parameters in older VAE checkpoints
_LEGACY_ATTN = {"query": "to_q", "key": "to_k", "value": "to_v", "proj_attn": "to_out.0"}


class AutoencoderKL(HipModule):
    """Decoder half of diffusers' AutoencoderKL (SD-1.5 config: block_out_channels (128, 256, 512, 512), layers_per_block 2,
    latent_channels 4, scaling_factor 0.18215)."""

    def __init__(self, block_out_channels=(128, 256, 512, 512), layers_per_block: int = 2, latent_channels: int = 4,
                 out_channels: int = 3, norm_num_groups: int = 32, scaling_factor: float = 0.18215):
        super().__init__()
        self.config = SimpleNamespace(block_out_channels=tuple(block_out_channels), layers_per_block=layers_per_block,
                                      latent_channels=latent_channels, out_channels=out_channels, norm_num_groups=norm_num_groups,
                                      scaling_factor=scaling_factor)
        self.post_quant_conv = nn.Conv2d(latent_channels, latent_channels, 1)
        self.decoder = Decoder(latent_channels, out_channels, block_out_channels, layers_per_block, norm_num_groups)

    @property
    def dtype(self):
        return self.post_quant_conv.weight.dtype

    @property
    def device(self):
        return self.post_quant_conv.weight.device

    def _apply(self, fn, *args, **kwargs):
        out = super()._apply(fn, *args, **kwargs)
        bump_pack_epoch()
        return out

    def load_state_dict(self, state_dict, strict: bool = True, **kw):
        """accepts a full AutoencoderKL checkpoint: encoder.* / quant_conv.* (the encode half) are ignored, legacy attention
        names are mapped, [C, C, 1, 1] attention weights of very old checkpoints are flattened"""
        sd = {}
        for k, v in state_dict.items():
            if k.startswith(("encoder.", "quant_conv.")):
                continue
            parts = k.split(".")
            if "attentions" in parts:
                for old, new in _LEGACY_ATTN.items():
                    if old in parts:
                        k = k.replace("." + old + ".", "." + new + ".")
                if k.endswith("weight") and v.ndim == 4 and "group_norm" not in k:
                    v = v.reshape(v.shape[0], v.shape[1])
            sd[k] = v
        out = super().load_state_dict(sd, strict=strict, **kw)
        bump_pack_epoch()
        return out

    @torch.no_grad()
    def decode(self, z: torch.Tensor, return_dict: bool = False):
        """z [N, 4, h, w] (already divided by scaling_factor) -> image [N, 3, 8h, 8w] fp32, as AutoencoderKL.decode(...)[0]"""
        if z.ndim != 4 or z.shape[1] != self.config.latent_channels:
            raise ValueError(f"decode expects [N, {self.config.latent_channels}, h, w], got {tuple(z.shape)}")
        if self._device_check and not z.is_cuda:
            raise RuntimeError("AutoencoderKL.decode runs on the GPU only (musev_amd has no CPU path)")
        n, c, h, w = z.shape
        geo = Geo(1, n, h, w)
        d = self.decoder
        # post_quant_conv (1x1, 4 -> 4): the K = 4 contraction padded to the kernel's 8-element granule
        zr = torch.zeros((geo.rows, 8), dtype=torch.float16, device=z.device)
        zr[:, :c] = z.permute(0, 2, 3, 1).reshape(geo.rows, c)
        wpq = self.packed("post_quant", lambda: ops.pad_cols(lin_w(self.post_quant_conv), 8))
        x = ops.gemm(zr, wpq, bias=lin_b(self.post_quant_conv))                       # [rows, 4]
        w_in = self.packed("conv_in64", lambda: ops.pad_cols(ops.pack_conv_weight(d.conv_in.weight.detach()), 64))
        x = ops.conv3x3_cin_small_gemm(x.contiguous(), w_in, w16(d.conv_in.bias), geo.n, h, w)
        x = d.mid_block.hip_forward(x, geo)
        for blk in d.up_blocks:
            x, geo = blk.hip_forward(x, geo)
        x = ops.groupnorm(x, w16(d.conv_norm_out.weight), w16(d.conv_norm_out.bias), geo.n, geo.hw, eps=d.conv_norm_out.eps,
                          silu=True, groups=d.conv_norm_out.num_groups)
        w_out = self.packed("conv_out", lambda: ops.pack_conv_weight(d.conv_out.weight.detach()))
        y = ops.conv3x3_cout_small(x, w_out, w16(d.conv_out.bias), geo.n, geo.h, geo.w, out_dtype=torch.float32)  # [rows, 3]
        img = y.view(n, geo.h, geo.w, self.config.out_channels).permute(0, 3, 1, 2)
        if return_dict:
            return SimpleNamespace(sample=img)
        return (img,)


class ImageProjModel(HipModule):
    """ip_adapter.ip_adapter.ImageProjModel: proj = Linear(clip_embeddings_dim, tokens * cross_attention_dim), norm =
    LayerNorm(cross_attention_dim); forward: [B, clip_dim] -> [B, tokens, cross_attention_dim]"""

    def __init__(self, cross_attention_dim: int = 768, clip_embeddings_dim: int = 1024, clip_extra_context_tokens: int = 4):
        super().__init__()
        self.cross_attention_dim = cross_attention_dim
        self.clip_extra_context_tokens = clip_extra_context_tokens
        self.proj = nn.Linear(clip_embeddings_dim, clip_extra_context_tokens * cross_attention_dim)
        self.norm = nn.LayerNorm(cross_attention_dim)

    def _apply(self, fn, *args, **kwargs):
        out = super()._apply(fn, *args, **kwargs)
        bump_pack_epoch()
        return out

    @torch.no_grad()
    def forward(self, image_embeds: torch.Tensor) -> torch.Tensor:
        if self._device_check and not image_embeds.is_cuda:
            raise RuntimeError("ImageProjModel runs on the GPU only (musev_amd has no CPU path)")
        x = image_embeds.reshape(-1, image_embeds.shape[-1]).to(torch.float16).contiguous()
        t = ops.gemm(x, lin_w(self.proj), bias=lin_b(self.proj))                                     # [B, tokens * dim]
        t = t.view(-1, self.cross_attention_dim)                                                      # [B * tokens, dim]
        y = ops.layernorm(t, w16(self.norm.weight), w16(self.norm.bias), eps=self.norm.eps)
        return y.view(-1, self.clip_extra_context_tokens, self.cross_attention_dim)
