"""load_unet_by_name: the three shipped model flavours (reference musev/models/unet_loader.py:206-273).

``sd_unet_model`` may be a checkpoint directory laid out like the reference expects
(``<path>/unet/config.json`` + ``diffusion_pytorch_model.{safetensors,bin}``, unet_3d_condition.py:1447-1531), an
``nn.Module`` / state dict to copy weights from, or None for a randomly initialised network (benchmarks, tests)."""
from __future__ import annotations

import json
import os
from typing import Optional, Union

import torch
from torch import nn

from .unet_3d_condition import UNet3DConditionModel

FLAVOUR_KWARGS = {
    # unet_loader.py:232-242
    "musev": dict(need_spatial_position_emb=False, need_t2i_ip_adapter=True, need_adain_temporal_cond=True,
                  t2i_ip_adapter_attn_processor="NonParamReferenceIPXFormersAttnProcessor"),
    # unet_loader.py:243-268
    "musev_referencenet": dict(temporal_conv_block="TemporalConvLayer", need_transformer_in=False,
                               temporal_transformer="TransformerTemporalModel", use_anivv1_cfg=True,
                               resnet_2d_skip_time_act=True, need_t2i_ip_adapter=True, need_adain_temporal_cond=True,
                               keep_vision_condtion=True, t2i_ip_adapter_attn_processor="NonParamReferenceIPXFormersAttnProcessor",
                               need_refer_emb=True, need_zero_vis_cond_temb=True, ip_adapter_cross_attn=True,
                               t2i_crossattn_ip_adapter_attn_processor="T2IReferencenetIPAdapterXFormersAttnProcessor"),
}
FLAVOUR_KWARGS["musev_referencenet_pose"] = FLAVOUR_KWARGS["musev_referencenet"]

_2D_TO_3D = {"CrossAttnDownBlock2D": "CrossAttnDownBlock3D", "DownBlock2D": "DownBlock3D", "UpBlock2D": "UpBlock3D",
             "CrossAttnUpBlock2D": "CrossAttnUpBlock3D"}
_CONFIG_KEYS = ("sample_size", "in_channels", "out_channels", "down_block_types", "up_block_types", "block_out_channels",
                "layers_per_block", "downsample_padding", "mid_block_scale_factor", "act_fn", "norm_num_groups", "norm_eps",
                "cross_attention_dim", "attention_head_dim")


def _read_state(path: str):
    st = os.path.join(path, "diffusion_pytorch_model.safetensors")
    if os.path.exists(st):
        from safetensors.torch import load_file
        return load_file(st)
    return torch.load(os.path.join(path, "diffusion_pytorch_model.bin"), map_location="cpu")


def load_unet_by_name(model_name: str, sd_unet_model: Union[str, nn.Module, dict, None] = None,
                      sd_model: Union[str, nn.Module, dict, None] = None, cross_attention_dim: int = 768,
                      dtype: torch.dtype = torch.float16, need_t2i_facein: bool = False,
                      need_t2i_ip_adapter_face: bool = False, strict: bool = True, **config_overrides) -> nn.Module:
    if model_name not in FLAVOUR_KWARGS:
        raise ValueError(f"unsupport model_name={model_name}, only support musev, musev_referencenet, musev_referencenet_pose")
    kwargs = dict(FLAVOUR_KWARGS[model_name])
    kwargs.update(cross_attention_dim=cross_attention_dim, need_t2i_facein=need_t2i_facein,
                  need_t2i_ip_adapter_face=need_t2i_ip_adapter_face)
    state = None
    if isinstance(sd_unet_model, str):
        unet_dir = os.path.join(sd_unet_model, "unet") if os.path.isdir(os.path.join(sd_unet_model, "unet")) else sd_unet_model
        with open(os.path.join(unet_dir, "config.json")) as f:
            cfg = json.load(f)
        # from_pretrained_2d: a 2-D SD config is turned into the 3-D block types (unet_3d_condition.py:140-159)
        for k in ("down_block_types", "up_block_types"):
            if k in cfg:
                cfg[k] = tuple(_2D_TO_3D.get(x, x) for x in cfg[k])
        kwargs.update({k: (tuple(v) if isinstance(v, list) else v) for k, v in cfg.items() if k in _CONFIG_KEYS})
        kwargs["cross_attention_dim"] = cross_attention_dim
        state = _read_state(unet_dir)
    elif isinstance(sd_unet_model, nn.Module):
        state = sd_unet_model.state_dict()
    elif isinstance(sd_unet_model, dict):
        state = sd_unet_model
    kwargs.update(config_overrides)
    unet = UNet3DConditionModel(**kwargs)
    if state is not None:
        missing, unexpected = unet.load_state_dict(state, strict=False)
        if strict:
            assert len(unexpected) == 0, f"unexpected keys: {unexpected[:8]}"
            assert len(missing) == 0, f"missing keys: {missing[:8]}"
    if sd_model is not None:
        # overwrite the T2I weights from the base SD model; unexpected keys must be empty (unet_loader.py:55-78)
        if isinstance(sd_model, str):
            sd_dir = os.path.join(sd_model, "unet") if os.path.isdir(os.path.join(sd_model, "unet")) else sd_model
            t2i = _read_state(sd_dir)
        else:
            t2i = sd_model.state_dict() if isinstance(sd_model, nn.Module) else sd_model
        missing, unexpected = unet.load_state_dict(t2i, strict=False)
        assert len(unexpected) == 0, f"unexpected keys: {unexpected[:8]}"
    return unet.to(dtype=dtype).eval()
