"""PoseGuider on HIP kernels (the part the shipped `musev_referencenet_pose` flavour uses): reference
musev/models/controlnet.py:326-402.  A conv stack -- conv_in, then per level (conv 3x3, conv 3x3 stride 2), each followed
by SiLU, and a zero-initialised conv_out -- that turns the pose images [b, c, f, H, W] into ``pose_guider_emb``
[b, 320, f, H/8, W/8], which the UNet adds right after conv_in (unet_3d_condition.py:1011-1016).  The pipeline runs it once
per call (pipeline_controlnet.py:1774-1781); with a PoseGuider installed the ControlNet is not run at all (:1217).

Parameters live under the reference's state-dict keys (``conv_in``, ``blocks.<i>``, ``conv_out``; InflatedConv3d is an
nn.Conv2d applied per frame, so the weights are plain [O, I, 3, 3]); every convolution is one launch of
mv_conv3x3_direct_f16 with the bias and the SiLU fused."""
from __future__ import annotations

import os
from typing import Tuple

import torch
import torch.nn as nn

from types import SimpleNamespace
from typing import List, Optional, Sequence, Union

from .. import ops
from .layers import HipModule, TimestepEmbedding, lin_b, lin_w, w16
from .runtime import Ctx, Geo

__all__ = ["PoseGuider", "ControlNetModel", "ControlNetOutput"]


class PoseGuider(HipModule):
    def __init__(self, conditioning_embedding_channels: int, conditioning_channels: int = 3,
                 block_out_channels: Tuple[int, ...] = (16, 32, 64, 128)):
        super().__init__()
        ch = tuple(block_out_channels)
        self.conv_in = nn.Conv2d(conditioning_channels, ch[0], kernel_size=3, padding=1)
        blocks = []
        for cin, cout in zip(ch[:-1], ch[1:]):
            blocks.append(nn.Conv2d(cin, cin, kernel_size=3, padding=1))
            blocks.append(nn.Conv2d(cin, cout, kernel_size=3, padding=1, stride=2))
        self.blocks = nn.ModuleList(blocks)
        self.conv_out = nn.Conv2d(ch[-1], conditioning_embedding_channels, kernel_size=3, padding=1)
        for p in self.conv_out.parameters():  # zero_module (controlnet.py:319-323, 353-360)
            nn.init.zeros_(p)

    @property
    def dtype(self):
        return self.conv_in.weight.dtype

    @property
    def device(self):
        return self.conv_in.weight.device

    def _conv(self, name: str, conv: nn.Conv2d, x: torch.Tensor, n: int, h: int, w: int, act: int):
        wp = self.packed(name, lambda: ops.pack_conv_weight(conv.weight.detach()))
        y = ops.conv3x3_direct(x, wp, w16(conv.bias), n, h, w, stride=conv.stride[0], act=act)
        s = conv.stride[0]
        return y, (h + 2 - 3) // s + 1, (w + 2 - 3) // s + 1

    def hip_rows(self, x: torch.Tensor, n: int, h: int, w: int):
        """the conv stack on channels-last rows [n*h*w, c] -> (rows [n*h'*w', emb], h', w')"""
        x, h, w = self._conv("conv_in", self.conv_in, x, n, h, w, ops.MV_ACT_SILU)
        for i, conv in enumerate(self.blocks):
            x, h, w = self._conv(f"blocks.{i}", conv, x, n, h, w, ops.MV_ACT_SILU)
        return self._conv("conv_out", self.conv_out, x, n, h, w, ops.MV_ACT_NONE)

    @torch.no_grad()
    def forward(self, conditioning: torch.Tensor) -> torch.Tensor:
        """conditioning [b, c, f, H, W] -> embedding [b, conditioning_embedding_channels, f, H/8, W/8] (for the default four
        levels), in the input's dtype when that is fp16 / fp32."""
        if self._device_check and not conditioning.is_cuda:
            raise RuntimeError("musev_amd.PoseGuider runs only on an MI355X (HIP) device; there is no CPU path")
        if conditioning.ndim != 5:
            raise ValueError(f"conditioning must be b c f h w, got ndim={conditioning.ndim}")
        b, c, f, h, w = conditioning.shape
        if c != self.conv_in.in_channels:
            raise ValueError(f"conditioning has {c} channels, conv_in expects {self.conv_in.in_channels}")
        # rows (b, f, y, x) x c: "b c f h w -> (b f) c h w" is index arithmetic on the channels-last rows
        x, h, w = self.hip_rows(ops.bcthw_to_bthwc(conditioning), b * f, h, w)
        out_dtype = conditioning.dtype if conditioning.dtype in (torch.float16, torch.float32) else torch.float32
        return ops.bthwc_to_bcthw(x, b, f, h, w, dtype=out_dtype)

    @classmethod
    def from_pretrained(cls, pretrained_model_path, conditioning_embedding_channels: int, conditioning_channels: int = 3,
                        block_out_channels: Tuple[int, ...] = (16, 32, 64, 128)):
        """reference :373-402: a plain ``torch.load`` state dict (or an in-memory dict), loaded with strict=False"""
        if isinstance(pretrained_model_path, dict):
            state = pretrained_model_path
        else:
            if not os.path.exists(pretrained_model_path):
                raise FileNotFoundError(f"There is no model file in {pretrained_model_path}")
            state = torch.load(pretrained_model_path, map_location="cpu")
        model = cls(conditioning_embedding_channels=conditioning_embedding_channels, conditioning_channels=conditioning_channels,
                    block_out_channels=block_out_channels)
        model.load_state_dict(state, strict=False)
        return model.eval()


class ControlNetOutput(SimpleNamespace):
    """down_block_res_samples (tuple), mid_block_res_sample -- the fields of diffusers' ControlNetOutput"""


class ControlNetModel(HipModule):
    """SD-1.5 ControlNet on HIP kernels: the per-window, per-step call of
    MusevControlNetPipeline.get_controlnet_emb (reference musev/pipelines/pipeline_controlnet.py:1202-1291; wrapper
    musev/models/controlnet.py:20-306).  The class itself is diffusers' ``ControlNetModel`` (un-vendored fork, >= v0.24;
    restated from the published algorithm, see oracle/controlnet.py for what is and is not pinned):

        emb = time_embedding(time_proj(t));  x = conv_in(sample) + controlnet_cond_embedding(controlnet_cond)
        12 down residuals + mid of the SD-1.5 UNet2D encoder (the ReferenceNet2D blocks of this package)
        residual_i = controlnet_down_blocks[i](res_i) * scale_i;  mid = controlnet_mid_block(x) * scale

    It is a composition of kernels that are parity-tested individually: the encoder is ReferenceNet2D's (pinned against the
    reference's own ReferenceNet2D), the conditioning embedding has PoseGuider's structure and state-dict keys (pinned
    against the reference's PoseGuider), the zero convolutions are 1x1 GEMMs with the conditioning scale folded into the
    packed weights.  Outputs are [(b t), c, h, w] channels-last VIEWS of the kernel rows: UNet3DConditionModel turns them
    back into rows without a copy.  The fork-only ``controlnet_cond_latents`` input is not restated (NotImplementedError)."""

    def __init__(self, in_channels: int = 4, conditioning_channels: int = 3,
                 down_block_types: Sequence[str] = ("CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "DownBlock2D"),
                 block_out_channels: Sequence[int] = (320, 640, 1280, 1280), layers_per_block: int = 2,
                 attention_head_dim: int = 8, cross_attention_dim: int = 768, norm_num_groups: int = 32, norm_eps: float = 1e-5,
                 conditioning_embedding_out_channels: Sequence[int] = (16, 32, 96, 256), global_pool_conditions: bool = False,
                 **_unused):
        super().__init__()
        from .referencenet import _DownBlock2D, _MidBlock2DCrossAttn
        if len(down_block_types) != len(block_out_channels):
            raise ValueError("Must provide the same number of `block_out_channels` as `down_block_types`.")
        if global_pool_conditions:
            raise NotImplementedError("global_pool_conditions (shuffle ControlNets) is outside this build")
        ch = tuple(block_out_channels)
        temb = ch[0] * 4
        self.config = SimpleNamespace(in_channels=in_channels, conditioning_channels=conditioning_channels,
                                      block_out_channels=ch, down_block_types=tuple(down_block_types),
                                      layers_per_block=layers_per_block, attention_head_dim=attention_head_dim,
                                      cross_attention_dim=cross_attention_dim, global_pool_conditions=False,
                                      conditioning_embedding_out_channels=tuple(conditioning_embedding_out_channels))
        self.block_out_channels = ch
        self.conv_in = nn.Conv2d(in_channels, ch[0], 3, padding=1)
        self.time_embedding = TimestepEmbedding(ch[0], temb)
        self.controlnet_cond_embedding = PoseGuider(ch[0], conditioning_channels, tuple(conditioning_embedding_out_channels))
        blocks, zero, cin = [], [nn.Conv2d(ch[0], ch[0], 1)], ch[0]
        for i, bt in enumerate(down_block_types):
            if bt not in ("CrossAttnDownBlock2D", "DownBlock2D"):
                raise ValueError(f"{bt} does not exist.")
            last = i == len(ch) - 1
            blocks.append(_DownBlock2D(cin, ch[i], temb, layers_per_block, attention_head_dim, cross_attention_dim,
                                       bt == "CrossAttnDownBlock2D", not last, norm_eps, norm_num_groups))
            zero += [nn.Conv2d(ch[i], ch[i], 1) for _ in range(layers_per_block + (0 if last else 1))]
            cin = ch[i]
        self.down_blocks = nn.ModuleList(blocks)
        self.mid_block = _MidBlock2DCrossAttn(ch[-1], temb, attention_head_dim, cross_attention_dim, norm_eps, norm_num_groups)
        self.controlnet_down_blocks = nn.ModuleList(zero)
        self.controlnet_mid_block = nn.Conv2d(ch[-1], ch[-1], 1)
        for m in list(self.controlnet_down_blocks) + [self.controlnet_mid_block]:  # zero_module, as upstream
            nn.init.zeros_(m.weight)
            nn.init.zeros_(m.bias)

    @property
    def dtype(self):
        return self.conv_in.weight.dtype

    @property
    def device(self):
        return self.conv_in.weight.device

    def _zero_conv(self, name: str, conv: nn.Conv2d, rows: torch.Tensor, scale: float) -> torch.Tensor:
        # (W x + b) * scale with the scale folded into the packed fp16 copies (one pair per distinct scale: the pipeline uses
        # conditioning_scale * controlnet_keep[i], i.e. a handful of values over a run)
        wp, bp = self.packed(f"{name}@{scale!r}", lambda: ((lin_w(conv).float() * scale).to(torch.float16).contiguous(),
                                                          (lin_b(conv).float() * scale).to(torch.float16).contiguous()))
        return ops.gemm(rows, wp, bias=bp)

    @torch.no_grad()
    def forward(self, sample: torch.Tensor, timestep, encoder_hidden_states: torch.Tensor, controlnet_cond: torch.Tensor,
                conditioning_scale: float = 1.0, guess_mode: bool = False, return_dict: bool = True,
                controlnet_cond_latents: Optional[torch.Tensor] = None, **_unused):
        """sample [(b t), 4, h, w]; encoder_hidden_states [(b t), L, D]; controlnet_cond [(b t), 3, 8h, 8w] ->
        (13 down residuals, mid residual), each [(b t), c, h_i, w_i]"""
        if self._device_check and not sample.is_cuda:
            raise RuntimeError("musev_amd.ControlNetModel runs only on an MI355X (HIP) device; there is no CPU path")
        if controlnet_cond_latents is not None:
            raise NotImplementedError("controlnet_cond_latents exists only in the un-vendored diffusers fork (CHANGES:5)")
        if sample.ndim != 4 or controlnet_cond.ndim != 4:
            raise ValueError("sample and controlnet_cond must be (b t) c h w")
        if isinstance(conditioning_scale, (list, tuple)):
            raise NotImplementedError("per-ControlNet scale lists belong to MultiControlNetModel")
        n, _, h, w = sample.shape
        if controlnet_cond.shape[0] != n or encoder_hidden_states.ndim != 3 or encoder_hidden_states.shape[0] != n:
            raise ValueError("controlnet_cond / encoder_hidden_states batch must equal the sample batch (b t)")
        dev = sample.device
        ch0 = self.block_out_channels[0]
        tt = timestep.to(device=dev, dtype=torch.float32).reshape(-1) if torch.is_tensor(timestep) else \
            torch.tensor([float(timestep)], dtype=torch.float32, device=dev)
        emb = self.time_embedding.hip_forward(ops.timestep_embedding(tt.expand(n).contiguous(), ch0), final_silu=False)
        text = encoder_hidden_states.to(dtype=torch.float16).reshape(-1, encoder_hidden_states.shape[-1]).contiguous()
        ctx = Ctx(temb_act=ops.silu(emb), femb_act=None, text=text, text_len=encoder_hidden_states.shape[1], vis_idx=None,
                  clip=None, clip_len=0, ip_scale=0.0, skip_temporal=True, text_src=encoder_hidden_states, clip_src=None)
        geo = Geo(n, 1, h, w)

        def rows_of(x: torch.Tensor) -> torch.Tensor:  # (b t) c h w -> channels-last rows
            return x.to(torch.float16).permute(0, 2, 3, 1).reshape(-1, x.shape[1]).contiguous()

        cond, ch_, cw_ = self.controlnet_cond_embedding.hip_rows(rows_of(controlnet_cond), n, controlnet_cond.shape[2],
                                                                 controlnet_cond.shape[3])
        if (ch_, cw_) != (h, w):
            raise ValueError(f"controlnet_cond embeds to {ch_}x{cw_}, the latents are {h}x{w}")
        w_in = self.packed("conv_in64", lambda: ops.pad_cols(ops.pack_conv_weight(self.conv_in.weight.detach()), 64))
        x = ops.conv3x3_cin_small_gemm(rows_of(sample), w_in, w16(self.conv_in.bias), n, h, w, add_=cond)
        res = [(x, geo)]
        for blk in self.down_blocks:
            x, geo, outs = blk.hip_forward(x, ctx, geo)
            res.extend(outs)
        x = self.mid_block.hip_forward(x, ctx, geo)

        scales = [float(conditioning_scale)] * (len(res) + 1)
        if guess_mode:  # 0.1 ... 1.0 from the shallowest residual to the mid block (published ControlNetModel.forward)
            scales = [float(v) * float(conditioning_scale) for v in torch.logspace(-1, 0, len(res) + 1).tolist()]

        def nchw(rows: torch.Tensor, g: Geo) -> torch.Tensor:
            return rows.view(n, g.h, g.w, rows.shape[1]).permute(0, 3, 1, 2)  # channels-last view, no copy

        down = tuple(nchw(self._zero_conv(f"controlnet_down_blocks.{i}", self.controlnet_down_blocks[i], r, scales[i]), g)
                     for i, (r, g) in enumerate(res))
        mid = nchw(self._zero_conv("controlnet_mid_block", self.controlnet_mid_block, x, scales[-1]), geo)
        if not return_dict:
            return down, mid
        return ControlNetOutput(down_block_res_samples=down, mid_block_res_sample=mid)
