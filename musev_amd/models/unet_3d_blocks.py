"""Down / mid / up blocks of the UNet3D (musev/models/unet_3d_blocks.py) on HIP kernels.

Layer order inside a CrossAttn*Block3D: ResnetBlock2D -> TemporalConvLayer -> Transformer2DModel ->
TransformerTemporalModel (-> AdaIN: a no-op for 4-D inputs, data_util.py:600-601, skipped) -> ReferEmbFuseAttention
(unet_3d_blocks.py:684-743; up :1192-1232; mid :379-431).  Skip connections are passed to the next resnet as a second
GEMM source instead of being concatenated (torch.cat at :1130,1342)."""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import torch
from torch import nn

from .attention_processor import ReferEmbFuseAttention
from .layers import Downsample2D, ResnetBlock2D, Upsample2D
from .resnet import TemporalConvLayer
from .runtime import Ctx, Geo
from .temporal_transformer import TransformerTemporalModel
from .transformer_2d import Transformer2DModel


def _refer(query_dim: int, heads: int) -> ReferEmbFuseAttention:
    return ReferEmbFuseAttention(query_dim=query_dim, heads=heads, dim_head=query_dim // heads)


class CrossAttnDownBlock3D(nn.Module):
    """unet_3d_blocks.py:436-772"""
    has_cross_attention = True

    def __init__(self, in_channels, out_channels, temb_channels, femb_channels, num_layers=1, resnet_eps=1e-6,
                 resnet_groups=32, attn_num_head_channels=1, cross_attention_dim=1280, add_downsample=True,
                 downsample_padding=1, temporal_conv_block=TemporalConvLayer, temporal_transformer=TransformerTemporalModel,
                 need_t2i_ip_adapter=False, ip_adapter_cross_attn=False, need_t2i_facein=False,
                 need_t2i_ip_adapter_face=False, resnet_2d_skip_time_act=False, need_refer_emb=False, **_unused):
        super().__init__()
        resnets, attentions, temp_attentions, temp_convs, refer = [], [], [], [], []
        self.need_refer_emb = need_refer_emb
        for i in range(num_layers):
            cin = in_channels if i == 0 else out_channels
            resnets.append(ResnetBlock2D(cin, out_channels, temb_channels, eps=resnet_eps, groups=resnet_groups,
                                         skip_time_act=resnet_2d_skip_time_act))
            temp_convs.append(temporal_conv_block(out_channels, out_channels, dropout=0.1, femb_channels=femb_channels)
                              if temporal_conv_block is not None else None)
            attentions.append(Transformer2DModel(attn_num_head_channels, out_channels // attn_num_head_channels,
                                                 in_channels=out_channels, num_layers=1, cross_attention_dim=cross_attention_dim,
                                                 norm_num_groups=resnet_groups, cross_attn_temporal_cond=need_t2i_ip_adapter,
                                                 ip_adapter_cross_attn=ip_adapter_cross_attn, need_t2i_facein=need_t2i_facein,
                                                 need_t2i_ip_adapter_face=need_t2i_ip_adapter_face))
            temp_attentions.append(temporal_transformer(attn_num_head_channels, out_channels // attn_num_head_channels,
                                                        in_channels=out_channels, num_layers=1, femb_channels=femb_channels,
                                                        cross_attention_dim=cross_attention_dim, norm_num_groups=resnet_groups)
                                   if temporal_transformer is not None else None)
            if need_refer_emb:
                refer.append(_refer(out_channels, attn_num_head_channels))
        self.resnets = nn.ModuleList(resnets)
        self.temp_convs = nn.ModuleList(temp_convs)
        self.attentions = nn.ModuleList(attentions)
        self.temp_attentions = nn.ModuleList(temp_attentions)
        if add_downsample:
            self.downsamplers = nn.ModuleList([Downsample2D(out_channels, padding=downsample_padding)])
            if need_refer_emb:
                refer.append(_refer(out_channels, attn_num_head_channels))
        else:
            self.downsamplers = None
        if need_refer_emb:
            self.refer_emb_attns = nn.ModuleList(refer)

    def hip_forward(self, x, ctx: Ctx, geo: Geo, refer_embs: Optional[Sequence[torch.Tensor]]):
        outs = []
        i = -1
        for i, (res, tconv, attn, tattn) in enumerate(zip(self.resnets, self.temp_convs, self.attentions, self.temp_attentions)):
            # (ctx.shared: computed once for both CFG halves until the first cross-attention closes the shared prefix -- runtime.PrefixMemo)
            x = ctx.shared((id(self), i, "res"), lambda x=x, res=res: res.hip_forward(x, None, ctx, geo))
            if tconv is not None:
                x = ctx.shared((id(self), i, "tconv"), lambda x=x, tconv=tconv: tconv.hip_forward(x, ctx, geo))
            x = attn.hip_forward(x, ctx, geo)
            if tattn is not None:
                x = tattn.hip_forward(x, ctx, geo)
            if self.need_refer_emb and refer_embs is not None:
                x = self.refer_emb_attns[i].hip_forward(x, refer_embs[i], geo)
            outs.append((x, geo))
        if self.downsamplers is not None:
            x = self.downsamplers[0].hip_forward(x, geo)
            geo = geo.down()
            if self.need_refer_emb and refer_embs is not None:
                x = self.refer_emb_attns[i + 1].hip_forward(x, refer_embs[i + 1], geo)
            outs.append((x, geo))
        return x, geo, outs


class DownBlock3D(nn.Module):
    """unet_3d_blocks.py:775-983"""
    has_cross_attention = False

    def __init__(self, in_channels, out_channels, temb_channels, femb_channels, num_layers=1, resnet_eps=1e-6,
                 resnet_groups=32, add_downsample=True, downsample_padding=1, temporal_conv_block=TemporalConvLayer,
                 resnet_2d_skip_time_act=False, need_refer_emb=False, attn_num_head_channels=1, **_unused):
        super().__init__()
        resnets, temp_convs, refer = [], [], []
        self.need_refer_emb = need_refer_emb
        for i in range(num_layers):
            cin = in_channels if i == 0 else out_channels
            resnets.append(ResnetBlock2D(cin, out_channels, temb_channels, eps=resnet_eps, groups=resnet_groups,
                                         skip_time_act=resnet_2d_skip_time_act))
            temp_convs.append(temporal_conv_block(out_channels, out_channels, dropout=0.1, femb_channels=femb_channels)
                              if temporal_conv_block is not None else None)
            if need_refer_emb:
                refer.append(_refer(out_channels, attn_num_head_channels))
        self.resnets = nn.ModuleList(resnets)
        self.temp_convs = nn.ModuleList(temp_convs)
        if add_downsample:
            self.downsamplers = nn.ModuleList([Downsample2D(out_channels, padding=downsample_padding)])
            if need_refer_emb:
                refer.append(_refer(out_channels, attn_num_head_channels))
        else:
            self.downsamplers = None
        if need_refer_emb:
            self.refer_emb_attns = nn.ModuleList(refer)

    def hip_forward(self, x, ctx: Ctx, geo: Geo, refer_embs: Optional[Sequence[torch.Tensor]]):
        outs = []
        i = -1
        for i, (res, tconv) in enumerate(zip(self.resnets, self.temp_convs)):
            x = res.hip_forward(x, None, ctx, geo)
            if tconv is not None:
                x = tconv.hip_forward(x, ctx, geo)
            if self.need_refer_emb and refer_embs is not None:
                x = self.refer_emb_attns[i].hip_forward(x, refer_embs[i], geo)
            outs.append((x, geo))
        if self.downsamplers is not None:
            x = self.downsamplers[0].hip_forward(x, geo)
            geo = geo.down()
            if self.need_refer_emb and refer_embs is not None:
                x = self.refer_emb_attns[i + 1].hip_forward(x, refer_embs[i + 1], geo)
            outs.append((x, geo))
        return x, geo, outs


class UNetMidBlock3DCrossAttn(nn.Module):
    """unet_3d_blocks.py:231-433: res, tconv, [attn, tattn, res, tconv]"""
    has_cross_attention = True

    def __init__(self, in_channels, temb_channels, femb_channels, num_layers=1, resnet_eps=1e-6, resnet_groups=32,
                 attn_num_head_channels=1, cross_attention_dim=1280, temporal_conv_block=TemporalConvLayer,
                 temporal_transformer=TransformerTemporalModel, need_t2i_ip_adapter=False, ip_adapter_cross_attn=False,
                 need_t2i_facein=False, need_t2i_ip_adapter_face=False, resnet_2d_skip_time_act=False, **_unused):
        super().__init__()

        def res():
            return ResnetBlock2D(in_channels, in_channels, temb_channels, eps=resnet_eps, groups=resnet_groups,
                                 skip_time_act=resnet_2d_skip_time_act)

        def tc():
            return (temporal_conv_block(in_channels, in_channels, dropout=0.1, femb_channels=femb_channels)
                    if temporal_conv_block is not None else None)

        resnets, temp_convs, attentions, temp_attentions = [res()], [tc()], [], []
        for _ in range(num_layers):
            attentions.append(Transformer2DModel(attn_num_head_channels, in_channels // attn_num_head_channels,
                                                 in_channels=in_channels, num_layers=1, cross_attention_dim=cross_attention_dim,
                                                 norm_num_groups=resnet_groups, cross_attn_temporal_cond=need_t2i_ip_adapter,
                                                 ip_adapter_cross_attn=ip_adapter_cross_attn, need_t2i_facein=need_t2i_facein,
                                                 need_t2i_ip_adapter_face=need_t2i_ip_adapter_face))
            temp_attentions.append(temporal_transformer(attn_num_head_channels, in_channels // attn_num_head_channels,
                                                        in_channels=in_channels, num_layers=1, femb_channels=femb_channels,
                                                        cross_attention_dim=cross_attention_dim, norm_num_groups=resnet_groups)
                                   if temporal_transformer is not None else None)
            resnets.append(res())
            temp_convs.append(tc())
        self.resnets = nn.ModuleList(resnets)
        self.temp_convs = nn.ModuleList(temp_convs)
        self.attentions = nn.ModuleList(attentions)
        self.temp_attentions = nn.ModuleList(temp_attentions)

    def hip_forward(self, x, ctx: Ctx, geo: Geo):
        x = self.resnets[0].hip_forward(x, None, ctx, geo)
        if self.temp_convs[0] is not None:
            x = self.temp_convs[0].hip_forward(x, ctx, geo)
        for attn, tattn, res, tconv in zip(self.attentions, self.temp_attentions, self.resnets[1:], self.temp_convs[1:]):
            x = attn.hip_forward(x, ctx, geo)
            if tattn is not None:
                x = tattn.hip_forward(x, ctx, geo)
            x = res.hip_forward(x, None, ctx, geo)
            if tconv is not None:
                x = tconv.hip_forward(x, ctx, geo)
        return x


class CrossAttnUpBlock3D(nn.Module):
    """unet_3d_blocks.py:986-1251"""
    has_cross_attention = True

    def __init__(self, in_channels, out_channels, prev_output_channel, temb_channels, femb_channels, num_layers=1,
                 resnet_eps=1e-6, resnet_groups=32, attn_num_head_channels=1, cross_attention_dim=1280, add_upsample=True,
                 temporal_conv_block=TemporalConvLayer, temporal_transformer=TransformerTemporalModel,
                 need_t2i_ip_adapter=False, ip_adapter_cross_attn=False, need_t2i_facein=False,
                 need_t2i_ip_adapter_face=False, resnet_2d_skip_time_act=False, **_unused):
        super().__init__()
        resnets, temp_convs, attentions, temp_attentions = [], [], [], []
        for i in range(num_layers):
            skip = in_channels if i == num_layers - 1 else out_channels
            rin = prev_output_channel if i == 0 else out_channels
            resnets.append(ResnetBlock2D(rin + skip, out_channels, temb_channels, eps=resnet_eps, groups=resnet_groups,
                                         skip_time_act=resnet_2d_skip_time_act))
            temp_convs.append(temporal_conv_block(out_channels, out_channels, dropout=0.1, femb_channels=femb_channels)
                              if temporal_conv_block is not None else None)
            attentions.append(Transformer2DModel(attn_num_head_channels, out_channels // attn_num_head_channels,
                                                 in_channels=out_channels, num_layers=1, cross_attention_dim=cross_attention_dim,
                                                 norm_num_groups=resnet_groups, cross_attn_temporal_cond=need_t2i_ip_adapter,
                                                 ip_adapter_cross_attn=ip_adapter_cross_attn, need_t2i_facein=need_t2i_facein,
                                                 need_t2i_ip_adapter_face=need_t2i_ip_adapter_face))
            temp_attentions.append(temporal_transformer(attn_num_head_channels, out_channels // attn_num_head_channels,
                                                        in_channels=out_channels, num_layers=1, femb_channels=femb_channels,
                                                        cross_attention_dim=cross_attention_dim, norm_num_groups=resnet_groups)
                                   if temporal_transformer is not None else None)
        self.resnets = nn.ModuleList(resnets)
        self.temp_convs = nn.ModuleList(temp_convs)
        self.attentions = nn.ModuleList(attentions)
        self.temp_attentions = nn.ModuleList(temp_attentions)
        self.upsamplers = nn.ModuleList([Upsample2D(out_channels)]) if add_upsample else None

    def hip_forward(self, x, skips: List[torch.Tensor], ctx: Ctx, geo: Geo, upsample_size=None):
        for res, tconv, attn, tattn in zip(self.resnets, self.temp_convs, self.attentions, self.temp_attentions):
            x = res.hip_forward(x, skips.pop(), ctx, geo)
            if tconv is not None:
                x = tconv.hip_forward(x, ctx, geo)
            x = attn.hip_forward(x, ctx, geo)
            if tattn is not None:
                x = tattn.hip_forward(x, ctx, geo)
        if self.upsamplers is not None:
            x, geo = self.upsamplers[0].hip_forward(x, geo, upsample_size)
        return x, geo


class UpBlock3D(nn.Module):
    """unet_3d_blocks.py:1254-1413"""
    has_cross_attention = False

    def __init__(self, in_channels, prev_output_channel, out_channels, temb_channels, femb_channels, num_layers=1,
                 resnet_eps=1e-6, resnet_groups=32, add_upsample=True, temporal_conv_block=TemporalConvLayer,
                 resnet_2d_skip_time_act=False, **_unused):
        super().__init__()
        resnets, temp_convs = [], []
        for i in range(num_layers):
            skip = in_channels if i == num_layers - 1 else out_channels
            rin = prev_output_channel if i == 0 else out_channels
            resnets.append(ResnetBlock2D(rin + skip, out_channels, temb_channels, eps=resnet_eps, groups=resnet_groups,
                                         skip_time_act=resnet_2d_skip_time_act))
            temp_convs.append(temporal_conv_block(out_channels, out_channels, dropout=0.1, femb_channels=femb_channels)
                              if temporal_conv_block is not None else None)
        self.resnets = nn.ModuleList(resnets)
        self.temp_convs = nn.ModuleList(temp_convs)
        self.upsamplers = nn.ModuleList([Upsample2D(out_channels)]) if add_upsample else None

    def hip_forward(self, x, skips: List[torch.Tensor], ctx: Ctx, geo: Geo, upsample_size=None):
        for res, tconv in zip(self.resnets, self.temp_convs):
            x = res.hip_forward(x, skips.pop(), ctx, geo)
            if tconv is not None:
                x = tconv.hip_forward(x, ctx, geo)
        if self.upsamplers is not None:
            x, geo = self.upsamplers[0].hip_forward(x, geo, upsample_size)
        return x, geo


def get_down_block(down_block_type, **kw):
    """unet_3d_blocks.py:50-141"""
    if down_block_type == "DownBlock3D":
        return DownBlock3D(**kw)
    if down_block_type == "CrossAttnDownBlock3D":
        if kw.get("cross_attention_dim") is None:
            raise ValueError("cross_attention_dim must be specified for CrossAttnDownBlock3D")
        return CrossAttnDownBlock3D(**kw)
    raise ValueError(f"{down_block_type} does not exist.")


def get_up_block(up_block_type, **kw):
    """unet_3d_blocks.py:144-228"""
    if up_block_type == "UpBlock3D":
        return UpBlock3D(**kw)
    if up_block_type == "CrossAttnUpBlock3D":
        if kw.get("cross_attention_dim") is None:
            raise ValueError("cross_attention_dim must be specified for CrossAttnUpBlock3D")
        return CrossAttnUpBlock3D(**kw)
    raise ValueError(f"{up_block_type} does not exist.")
