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

from .. import ops
from .layers import HipModule, w16

__all__ = ["PoseGuider"]


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
        n = b * f
        x = ops.bcthw_to_bthwc(conditioning)  # rows (b, f, y, x) x c: "b c f h w -> (b f) c h w" is index arithmetic
        x, h, w = self._conv("conv_in", self.conv_in, x, n, h, w, ops.MV_ACT_SILU)
        for i, conv in enumerate(self.blocks):
            x, h, w = self._conv(f"blocks.{i}", conv, x, n, h, w, ops.MV_ACT_SILU)
        x, h, w = self._conv("conv_out", self.conv_out, x, n, h, w, ops.MV_ACT_NONE)
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
