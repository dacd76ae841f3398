"""Stand-ins for the reference's UN-VENDORED third-party dependencies, so that the reference's own source under
/root/reference/musev can be imported and executed in this container to produce golden vectors.

TEST INFRASTRUCTURE ONLY (used by tests/golden/make_reference_goldens.py; never imported by musev_amd).

The reference imports `diffusers` (TMElyralab fork @tme -- its submodule directory is empty here), `xformers`, `mmcm`
and `accelerate`; none is installed and there is no network.  This module registers minimal replacement
modules in ``sys.modules``.  Everything in here is a *restatement of upstream diffusers v0.24 semantics* for exactly
the symbols the hot path touches -- it is NOT the fork's code, so goldens produced through it
pin the reference's vendored logic (block wiring, attention processors, index quirks, temporal layers, window
scheduler, DDIM step) but leave the un-vendored pieces themselves unverifiable against the fork.
"""
from __future__ import annotations

import functools
import inspect
import math
import sys
import types
from collections import OrderedDict
from dataclasses import dataclass, fields
from enum import Enum
from typing import Any, Optional

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn


def _mod(name: str) -> types.ModuleType:
    if name in sys.modules:
        return sys.modules[name]
    m = types.ModuleType(name)
    m.__path__ = []  # behave like a package so that submodule imports resolve through sys.modules
    sys.modules[name] = m
    parent, _, child = name.rpartition(".")
    if parent:
        setattr(_mod(parent), child, m)
    return m


# ---------------------------------------------------------------------------------------------- configuration_utils
class _Cfg(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e


def register_to_config(init):
    @functools.wraps(init)
    def inner(self, *args, **kwargs):
        init(self, *args, **kwargs)
        sig = inspect.signature(init)
        params = [p for n, p in sig.parameters.items() if n != "self"]
        cfg = {p.name: p.default for p in params if p.default is not inspect.Parameter.empty}
        for p, a in zip(params, args):
            cfg[p.name] = a
        cfg.update({k: v for k, v in kwargs.items() if not k.startswith("_")})
        object.__setattr__(self, "_internal_dict", _Cfg(cfg))
    return inner


class ConfigMixin:
    config_name = "config.json"

    @property
    def config(self):
        return self.__dict__.get("_internal_dict", _Cfg())

    def register_to_config(self, **kw):
        d = self.__dict__.get("_internal_dict") or _Cfg()
        d.update(kw)
        object.__setattr__(self, "_internal_dict", d)


class ModelMixin(nn.Module):
    """nn.Module + the config attribute fall-through that diffusers' ModelMixin.__getattr__ provides (the reference
    relies on it: ``self.temporal_transformer`` in UNet3DConditionModel.forward resolves to the config string)."""
    _supports_gradient_checkpointing = False

    def __getattr__(self, name):
        d = self.__dict__.get("_internal_dict")
        if d is not None and name in d and name not in ("_parameters", "_buffers", "_modules"):
            return d[name]
        return super().__getattr__(name)

    @property
    def dtype(self):
        return next(self.parameters()).dtype

    @property
    def device(self):
        return next(self.parameters()).device


class BaseOutput(OrderedDict):
    def __post_init__(self):
        for f in fields(self):
            v = getattr(self, f.name)
            if v is not None:
                self[f.name] = v

    def __getitem__(self, k):
        if isinstance(k, str):
            return dict(self.items())[k]
        return tuple(self.values())[k]

    def to_tuple(self):
        return tuple(self[k] for k in self.keys())


# ---------------------------------------------------------------------------------------------- lora / embeddings
class LoRACompatibleLinear(nn.Linear):
    def forward(self, x, scale: float = 1.0):
        return super().forward(x)


class LoRACompatibleConv(nn.Conv2d):
    def forward(self, x, scale: float = 1.0):
        return super().forward(x)


def get_timestep_embedding(timesteps, embedding_dim, flip_sin_to_cos=False, downscale_freq_shift=1, scale=1, max_period=10000):
    half = embedding_dim // 2
    exponent = -math.log(max_period) * torch.arange(0, half, dtype=torch.float32, device=timesteps.device)
    exponent = exponent / (half - downscale_freq_shift)
    emb = timesteps[:, None].float() * torch.exp(exponent)[None, :]
    emb = scale * emb
    emb = torch.cat([torch.sin(emb), torch.cos(emb)], dim=-1)
    if flip_sin_to_cos:
        emb = torch.cat([emb[:, half:], emb[:, :half]], dim=-1)
    return emb


class Timesteps(nn.Module):
    def __init__(self, num_channels, flip_sin_to_cos, downscale_freq_shift):
        super().__init__()
        self.num_channels, self.flip_sin_to_cos, self.downscale_freq_shift = num_channels, flip_sin_to_cos, downscale_freq_shift

    def forward(self, timesteps):
        return get_timestep_embedding(timesteps, self.num_channels, self.flip_sin_to_cos, self.downscale_freq_shift)


class TimestepEmbedding(nn.Module):
    def __init__(self, in_channels, time_embed_dim, act_fn="silu", out_dim=None, post_act_fn=None, cond_proj_dim=None):
        super().__init__()
        self.linear_1 = nn.Linear(in_channels, time_embed_dim)
        self.act = nn.SiLU()
        self.linear_2 = nn.Linear(time_embed_dim, out_dim or time_embed_dim)

    def forward(self, sample, condition=None):
        return self.linear_2(self.act(self.linear_1(sample)))


# ---------------------------------------------------------------------------------------------- resnet
class ResnetBlock2D(nn.Module):
    def __init__(self, *, in_channels, out_channels=None, conv_shortcut=False, dropout=0.0, temb_channels=512, groups=32,
                 groups_out=None, pre_norm=True, eps=1e-6, non_linearity="swish", skip_time_act=False,
                 time_embedding_norm="default", kernel=None, output_scale_factor=1.0, use_in_shortcut=None, up=False,
                 down=False, conv_shortcut_bias=True, conv_2d_out_channels=None):
        super().__init__()
        out_channels = out_channels or in_channels
        assert time_embedding_norm == "default" and not up and not down
        self.norm1 = nn.GroupNorm(groups, in_channels, eps=eps, affine=True)
        self.conv1 = LoRACompatibleConv(in_channels, out_channels, 3, 1, 1)
        self.time_emb_proj = LoRACompatibleLinear(temb_channels, out_channels) if temb_channels is not None else None
        self.norm2 = nn.GroupNorm(groups_out or groups, out_channels, eps=eps, affine=True)
        self.dropout = nn.Dropout(dropout)
        self.conv2 = LoRACompatibleConv(out_channels, conv_2d_out_channels or out_channels, 3, 1, 1)
        self.nonlinearity = nn.SiLU()
        self.skip_time_act = skip_time_act
        self.output_scale_factor = output_scale_factor
        use_in_shortcut = in_channels != out_channels if use_in_shortcut is None else use_in_shortcut
        self.conv_shortcut = LoRACompatibleConv(in_channels, out_channels, 1, 1, 0, bias=conv_shortcut_bias) if use_in_shortcut else None

    def forward(self, input_tensor, temb, scale: float = 1.0):
        h = self.conv1(self.nonlinearity(self.norm1(input_tensor)))
        if self.time_emb_proj is not None:
            if not self.skip_time_act:
                temb = self.nonlinearity(temb)
            temb = self.time_emb_proj(temb)[:, :, None, None]
        if temb is not None:
            h = h + temb
        h = self.conv2(self.dropout(self.nonlinearity(self.norm2(h))))
        if self.conv_shortcut is not None:
            input_tensor = self.conv_shortcut(input_tensor)
        return (input_tensor + h) / self.output_scale_factor


class Downsample2D(nn.Module):
    def __init__(self, channels, use_conv=False, out_channels=None, padding=1, name="conv"):
        super().__init__()
        assert use_conv
        self.conv = LoRACompatibleConv(channels, out_channels or channels, 3, stride=2, padding=padding)
        self.padding = padding

    def forward(self, hidden_states, scale: float = 1.0):
        if self.padding == 0:
            hidden_states = F.pad(hidden_states, (0, 1, 0, 1), mode="constant", value=0)
        return self.conv(hidden_states)


class Upsample2D(nn.Module):
    def __init__(self, channels, use_conv=False, use_conv_transpose=False, out_channels=None, name="conv"):
        super().__init__()
        assert use_conv and not use_conv_transpose
        self.conv = LoRACompatibleConv(channels, out_channels or channels, 3, padding=1)

    def forward(self, hidden_states, output_size=None, scale: float = 1.0):
        if output_size is None:
            hidden_states = F.interpolate(hidden_states, scale_factor=2.0, mode="nearest")
        else:
            hidden_states = F.interpolate(hidden_states, size=output_size, mode="nearest")
        return self.conv(hidden_states)


# ---------------------------------------------------------------------------------------------- attention
class AttnProcessor2_0:
    def __call__(self, attn, hidden_states, encoder_hidden_states=None, attention_mask=None, temb=None, scale: float = 1.0):
        residual = hidden_states
        batch_size = hidden_states.shape[0]
        query = attn.to_q(hidden_states, scale=scale)
        if encoder_hidden_states is None:
            encoder_hidden_states = hidden_states
        key = attn.to_k(encoder_hidden_states, scale=scale)
        value = attn.to_v(encoder_hidden_states, scale=scale)
        head_dim = key.shape[-1] // attn.heads
        q = query.view(batch_size, -1, attn.heads, head_dim).transpose(1, 2)
        k = key.view(batch_size, -1, attn.heads, head_dim).transpose(1, 2)
        v = value.view(batch_size, -1, attn.heads, head_dim).transpose(1, 2)
        o = F.scaled_dot_product_attention(q, k, v, attn_mask=attention_mask, dropout_p=0.0, is_causal=False)
        o = o.transpose(1, 2).reshape(batch_size, -1, attn.heads * head_dim).to(query.dtype)
        o = attn.to_out[1](attn.to_out[0](o, scale=scale))
        if attn.residual_connection:
            o = o + residual
        return o / attn.rescale_output_factor


class AttnProcessor(AttnProcessor2_0):
    pass


class XFormersAttnProcessor(AttnProcessor2_0):
    def __init__(self, attention_op=None):
        self.attention_op = attention_op


class Attention(nn.Module):
    def __init__(self, query_dim, cross_attention_dim=None, heads=8, dim_head=64, dropout=0.0, bias=False,
                 upcast_attention=False, upcast_softmax=False, cross_attention_norm=None, cross_attention_norm_num_groups=32,
                 added_kv_proj_dim=None, norm_num_groups=None, spatial_norm_dim=None, out_bias=True, scale_qk=True,
                 only_cross_attention=False, eps=1e-5, rescale_output_factor=1.0, residual_connection=False,
                 _from_deprecated_attn_block=False, processor=None):
        super().__init__()
        self.inner_dim = dim_head * heads
        cross_attention_dim = cross_attention_dim if cross_attention_dim is not None else query_dim
        self.cross_attention_dim = cross_attention_dim
        self.rescale_output_factor = rescale_output_factor
        self.residual_connection = residual_connection
        self.scale = dim_head ** -0.5 if scale_qk else 1.0
        self.heads = heads
        self.only_cross_attention = only_cross_attention
        self.group_norm = None
        self.spatial_norm = None
        self.norm_cross = None
        self.to_q = LoRACompatibleLinear(query_dim, self.inner_dim, bias=bias)
        self.to_k = LoRACompatibleLinear(cross_attention_dim, self.inner_dim, bias=bias)
        self.to_v = LoRACompatibleLinear(cross_attention_dim, self.inner_dim, bias=bias)
        self.to_out = nn.ModuleList([LoRACompatibleLinear(self.inner_dim, query_dim, bias=out_bias), nn.Dropout(dropout)])
        self.set_processor(processor if processor is not None else AttnProcessor2_0())

    def set_processor(self, processor, _remove_lora=False):
        self.processor = processor

    def set_use_memory_efficient_attention_xformers(self, use, attention_op=None):
        return None

    def forward(self, hidden_states, encoder_hidden_states=None, attention_mask=None, **cross_attention_kwargs):
        return self.processor(self, hidden_states, encoder_hidden_states=encoder_hidden_states,
                              attention_mask=attention_mask, **cross_attention_kwargs)

    def head_to_batch_dim(self, tensor, out_dim=3):
        b, l, c = tensor.shape
        h = self.heads
        tensor = tensor.reshape(b, l, h, c // h).permute(0, 2, 1, 3)
        if out_dim == 3:
            tensor = tensor.reshape(b * h, l, c // h)
        return tensor

    def batch_to_head_dim(self, tensor):
        bh, l, d = tensor.shape
        h = self.heads
        return tensor.reshape(bh // h, h, l, d).permute(0, 2, 1, 3).reshape(bh // h, l, d * h)

    def prepare_attention_mask(self, attention_mask, target_length, batch_size, out_dim=3):
        if attention_mask is None:
            return None
        raise NotImplementedError("attention masks are not used on the hot path")


class GEGLU(nn.Module):
    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.proj = LoRACompatibleLinear(dim_in, dim_out * 2)

    def forward(self, hidden_states, scale: float = 1.0):
        hidden_states, gate = self.proj(hidden_states).chunk(2, dim=-1)
        return hidden_states * F.gelu(gate)


class FeedForward(nn.Module):
    def __init__(self, dim, dim_out=None, mult=4, dropout=0.0, activation_fn="geglu", final_dropout=False):
        super().__init__()
        assert activation_fn == "geglu"
        inner = int(dim * mult)
        self.net = nn.ModuleList([GEGLU(dim, inner), nn.Dropout(dropout), LoRACompatibleLinear(inner, dim_out or dim)])

    def forward(self, hidden_states, scale: float = 1.0):
        for m in self.net:
            hidden_states = m(hidden_states, scale) if isinstance(m, (GEGLU, LoRACompatibleLinear)) else m(hidden_states)
        return hidden_states


class AdaLayerNorm(nn.Module):
    pass


class AdaLayerNormZero(nn.Module):
    pass


class DiffusersBasicTransformerBlock(nn.Module):
    def __init__(self, dim, num_attention_heads, attention_head_dim, dropout=0.0, cross_attention_dim=None,
                 activation_fn="geglu", num_embeds_ada_norm=None, attention_bias=False, only_cross_attention=False,
                 double_self_attention=False, upcast_attention=False, norm_elementwise_affine=True, norm_type="layer_norm",
                 final_dropout=False, attention_type="default"):
        super().__init__()
        self.only_cross_attention = only_cross_attention
        self.use_ada_layer_norm_zero = False
        self.use_ada_layer_norm = False
        self.norm1 = nn.LayerNorm(dim, elementwise_affine=norm_elementwise_affine)
        self.attn1 = Attention(dim, heads=num_attention_heads, dim_head=attention_head_dim, dropout=dropout, bias=attention_bias,
                               cross_attention_dim=cross_attention_dim if only_cross_attention else None)
        if cross_attention_dim is not None or double_self_attention:
            self.norm2 = nn.LayerNorm(dim, elementwise_affine=norm_elementwise_affine)
            self.attn2 = Attention(dim, cross_attention_dim=cross_attention_dim if not double_self_attention else None,
                                   heads=num_attention_heads, dim_head=attention_head_dim, dropout=dropout, bias=attention_bias)
        else:
            self.norm2 = None
            self.attn2 = None
        self.norm3 = nn.LayerNorm(dim, elementwise_affine=norm_elementwise_affine)
        self.ff = FeedForward(dim, dropout=dropout, activation_fn=activation_fn, final_dropout=final_dropout)
        self._chunk_size = None
        self._chunk_dim = 0


@dataclass
class Transformer2DModelOutput(BaseOutput):
    sample: torch.FloatTensor = None


class DiffusersTransformer2DModel(ModelMixin, ConfigMixin):
    def __init__(self, num_attention_heads=16, attention_head_dim=88, in_channels=None, out_channels=None, num_layers=1,
                 dropout=0.0, norm_num_groups=32, cross_attention_dim=None, attention_bias=False, sample_size=None,
                 num_vector_embeds=None, patch_size=None, activation_fn="geglu", num_embeds_ada_norm=None,
                 use_linear_projection=False, only_cross_attention=False, double_self_attention=False,
                 upcast_attention=False, norm_type="layer_norm", norm_elementwise_affine=True, attention_type="default"):
        super().__init__()
        assert in_channels is not None and patch_size is None and num_vector_embeds is None and not use_linear_projection
        self.use_linear_projection = use_linear_projection
        self.num_attention_heads, self.attention_head_dim = num_attention_heads, attention_head_dim
        inner = num_attention_heads * attention_head_dim
        self.is_input_continuous, self.is_input_vectorized, self.is_input_patches = True, False, False
        self.in_channels = in_channels
        self.norm = nn.GroupNorm(num_groups=norm_num_groups, num_channels=in_channels, eps=1e-6, affine=True)
        self.proj_in = LoRACompatibleConv(in_channels, inner, kernel_size=1, stride=1, padding=0)
        self.transformer_blocks = nn.ModuleList([])
        self.out_channels = in_channels if out_channels is None else out_channels
        self.proj_out = LoRACompatibleConv(inner, in_channels, kernel_size=1, stride=1, padding=0)
        self.adaln_single = None
        self.caption_projection = None
        self.gradient_checkpointing = False


@dataclass
class TransformerTemporalModelOutput(BaseOutput):
    sample: torch.FloatTensor = None


@dataclass
class UNet3DConditionOutputShim(BaseOutput):
    sample: torch.FloatTensor = None


# ---------------------------------------------------------------------------------------------- schedulers
@dataclass
class DDIMSchedulerOutput(BaseOutput):
    prev_sample: torch.FloatTensor = None
    pred_original_sample: Optional[torch.FloatTensor] = None


class KarrasDiffusionSchedulers(Enum):
    DDIMScheduler = 1


class SchedulerMixin:
    pass


class DiffusersDDIMScheduler(SchedulerMixin, ConfigMixin):
    order = 1

    def __init__(self, num_train_timesteps=1000, beta_start=0.0001, beta_end=0.02, beta_schedule="linear", trained_betas=None,
                 clip_sample=True, set_alpha_to_one=True, steps_offset=0, prediction_type="epsilon", thresholding=False,
                 dynamic_thresholding_ratio=0.995, clip_sample_range=1.0, sample_max_value=1.0, timestep_spacing="leading",
                 rescale_betas_zero_snr=False):
        if beta_schedule == "linear":
            self.betas = torch.linspace(beta_start, beta_end, num_train_timesteps, dtype=torch.float32)
        elif beta_schedule == "scaled_linear":
            self.betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
        else:
            raise NotImplementedError(beta_schedule)
        if rescale_betas_zero_snr:  # upstream rescale_zero_terminal_snr
            abs_ = torch.cumprod(1.0 - self.betas, dim=0).sqrt()
            a0, aT = abs_[0].clone(), abs_[-1].clone()
            ab = ((abs_ - aT) * (a0 / (a0 - aT))) ** 2
            self.betas = 1.0 - torch.cat([ab[0:1], ab[1:] / ab[:-1]])
        self.alphas = 1.0 - self.betas
        self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)
        self.final_alpha_cumprod = torch.tensor(1.0) if set_alpha_to_one else self.alphas_cumprod[0]
        self.init_noise_sigma = 1.0
        self.num_inference_steps = None
        self.timesteps = torch.from_numpy(np.arange(0, num_train_timesteps)[::-1].copy().astype(np.int64))

    def scale_model_input(self, sample, timestep=None):
        return sample

    def _get_variance(self, timestep, prev_timestep):
        a_t = self.alphas_cumprod[timestep]
        a_p = self.alphas_cumprod[prev_timestep] if prev_timestep >= 0 else self.final_alpha_cumprod
        return (1 - a_p) / (1 - a_t) * (1 - a_t / a_p)

    def set_timesteps(self, num_inference_steps, device=None):
        self.num_inference_steps = num_inference_steps
        if self.config.timestep_spacing == "trailing":
            ts = np.round(np.arange(self.config.num_train_timesteps, 0, -self.config.num_train_timesteps / num_inference_steps)).astype(np.int64) - 1
        else:
            assert self.config.timestep_spacing == "leading"
            ratio = self.config.num_train_timesteps // num_inference_steps
            ts = (np.arange(0, num_inference_steps) * ratio).round()[::-1].copy().astype(np.int64) + self.config.steps_offset
        self.timesteps = torch.from_numpy(ts).to(device)


@dataclass
class EulerDiscreteSchedulerOutput(BaseOutput):
    prev_sample: torch.FloatTensor = None
    pred_original_sample: Optional[torch.FloatTensor] = None


class DiffusersEulerDiscreteScheduler(SchedulerMixin, ConfigMixin):
    """upstream diffusers v0.24 EulerDiscreteScheduler, the parts the reference's subclass relies on (constructor
    tables, set_timesteps with linear interpolation, scale_model_input, init_noise_sigma, step-index bookkeeping)"""
    order = 1

    def __init__(self, num_train_timesteps=1000, beta_start=0.0001, beta_end=0.02, beta_schedule="linear", trained_betas=None,
                 prediction_type="epsilon", interpolation_type="linear", use_karras_sigmas=False, timestep_spacing="linspace",
                 steps_offset=0):
        if beta_schedule == "linear":
            self.betas = torch.linspace(beta_start, beta_end, num_train_timesteps, dtype=torch.float32)
        elif beta_schedule == "scaled_linear":
            self.betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
        else:
            raise NotImplementedError(beta_schedule)
        self.register_to_config(num_train_timesteps=num_train_timesteps, beta_start=beta_start, beta_end=beta_end,
                                beta_schedule=beta_schedule, prediction_type=prediction_type, interpolation_type=interpolation_type,
                                use_karras_sigmas=use_karras_sigmas, timestep_spacing=timestep_spacing, steps_offset=steps_offset)
        self.alphas = 1.0 - self.betas
        self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)
        sigmas = (((1 - self.alphas_cumprod) / self.alphas_cumprod) ** 0.5).numpy()
        sigmas = np.concatenate([sigmas[::-1], [0.0]]).astype(np.float32)
        self.sigmas = torch.from_numpy(sigmas)
        self.num_inference_steps = None
        self.timesteps = torch.from_numpy(np.linspace(0, num_train_timesteps - 1, num_train_timesteps, dtype=float)[::-1].copy())
        self.is_scale_input_called = False
        self.use_karras_sigmas = use_karras_sigmas
        self._step_index = None

    @property
    def init_noise_sigma(self):
        max_sigma = max(self.sigmas) if isinstance(self.sigmas, list) else self.sigmas.max()
        if self.config.timestep_spacing in ["linspace", "trailing"]:
            return max_sigma
        return (max_sigma ** 2 + 1) ** 0.5

    @property
    def step_index(self):
        return self._step_index

    def _init_step_index(self, timestep):
        if isinstance(timestep, torch.Tensor):
            timestep = timestep.to(self.timesteps.device)
        index_candidates = (self.timesteps == timestep).nonzero()
        step_index = index_candidates[1] if len(index_candidates) > 1 else index_candidates[0]
        self._step_index = step_index.item()

    def scale_model_input(self, sample, timestep):
        if self.step_index is None:
            self._init_step_index(timestep)
        sigma = self.sigmas[self.step_index]
        sample = sample / ((sigma ** 2 + 1) ** 0.5)
        self.is_scale_input_called = True
        return sample

    def set_timesteps(self, num_inference_steps, device=None):
        self.num_inference_steps = num_inference_steps
        n_train = self.config.num_train_timesteps
        if self.config.timestep_spacing == "linspace":
            timesteps = np.linspace(0, n_train - 1, num_inference_steps, dtype=np.float32)[::-1].copy()
        elif self.config.timestep_spacing == "leading":
            step_ratio = n_train // self.num_inference_steps
            timesteps = (np.arange(0, num_inference_steps) * step_ratio).round()[::-1].copy().astype(np.float32)
            timesteps += self.config.steps_offset
        elif self.config.timestep_spacing == "trailing":
            step_ratio = n_train / self.num_inference_steps
            timesteps = (np.arange(n_train, 0, -step_ratio)).round().copy().astype(np.float32)
            timesteps -= 1
        else:
            raise ValueError(self.config.timestep_spacing)
        sigmas = (((1 - self.alphas_cumprod) / self.alphas_cumprod) ** 0.5).numpy()
        sigmas = np.interp(timesteps, np.arange(0, len(sigmas)), sigmas)
        sigmas = np.concatenate([sigmas, [0.0]]).astype(np.float32)
        self.sigmas = torch.from_numpy(sigmas).to(device=device)
        self.timesteps = torch.from_numpy(timesteps).to(device=device)
        self._step_index = None


def randn_tensor(shape, generator=None, device=None, dtype=None, layout=None):
    """diffusers.utils.torch_utils.randn_tensor (upstream v0.24): a list of generators draws one batch item each"""
    if isinstance(generator, list) and len(generator) == 1:
        generator = generator[0]
    if isinstance(generator, list):
        item = (1,) + tuple(shape)[1:]
        return torch.cat([torch.randn(item, generator=g, device=device, dtype=dtype) for g in generator], dim=0)
    return torch.randn(shape, generator=generator, device=device, dtype=dtype)


def memory_efficient_attention(query, key, value, attn_bias=None, op=None, scale=None, p=0.0):
    """xformers.ops.memory_efficient_attention for the 3-D [batch*heads, tokens, dim] layout the reference uses."""
    scale = query.shape[-1] ** -0.5 if scale is None else scale
    # same arithmetic per (batch*head) slice; sliced over the batch axis so that the BASELINE-size cases (208 x 4096 x 8192
    # scores = 28 GB at once) fit the host memory
    per = max(1, (1 << 28) // max(1, query.shape[1] * key.shape[1]))
    outs = []
    for b0 in range(0, query.shape[0], per):
        s = torch.bmm(query[b0:b0 + per], key[b0:b0 + per].transpose(1, 2)) * scale
        if attn_bias is not None:
            s = s + (attn_bias[b0:b0 + per] if attn_bias.dim() == 3 and attn_bias.shape[0] == query.shape[0] else attn_bias)
        outs.append(torch.bmm(torch.softmax(s, dim=-1), value[b0:b0 + per]))
    return outs[0] if len(outs) == 1 else torch.cat(outs, dim=0)


def generate_sample_idxs(total, window_size, step, sample_rate=1, drop_last=False):
    out, s = [], 0
    while s < total:
        idx = list(range(s, min(s + window_size * sample_rate, total), sample_rate))
        if len(idx) < window_size and drop_last:
            break
        out.append(idx)
        s += step
    return out


def install(reference_root: str = "/root/reference") -> None:
    """register the stand-in modules and put the reference on sys.path"""
    d = _mod("diffusers")
    d.__version__ = "0.24.0-shim"
    cu = _mod("diffusers.configuration_utils")
    cu.ConfigMixin, cu.register_to_config = ConfigMixin, register_to_config
    _mod("diffusers.loaders").UNet2DConditionLoadersMixin = type("UNet2DConditionLoadersMixin", (), {})
    u = _mod("diffusers.utils")
    u.BaseOutput = BaseOutput
    for k in ("CONFIG_NAME", "DIFFUSERS_CACHE", "FLAX_WEIGHTS_NAME", "SAFETENSORS_WEIGHTS_NAME", "WEIGHTS_NAME"):
        setattr(u, k, k.lower())
    u.HF_HUB_OFFLINE = True
    u._add_variant = lambda name, variant=None: name
    u._get_model_file = lambda *a, **k: None
    u.deprecate = lambda *a, **k: None
    u.is_accelerate_available = lambda: False
    u.is_torch_version = lambda op, v: True
    lg = _mod("diffusers.utils.logging")
    import logging as _logging
    lg.get_logger = _logging.getLogger
    u.logging = lg
    _mod("diffusers.utils.constants").USE_PEFT_BACKEND = False
    _mod("diffusers.utils.import_utils")._safetensors_available = True
    tu = _mod("diffusers.utils.torch_utils")
    tu.maybe_allow_in_graph = lambda cls: cls
    tu.randn_tensor = randn_tensor
    _mod("diffusers.models")
    em = _mod("diffusers.models.embeddings")
    em.TimestepEmbedding, em.Timesteps = TimestepEmbedding, Timesteps
    em.CombinedTimestepLabelEmbeddings = em.ImagePositionalEmbeddings = em.PatchEmbed = type("Unused", (nn.Module,), {})
    em.get_2d_sincos_pos_embed_from_grid = lambda *a, **k: (_ for _ in ()).throw(NotImplementedError())
    lo = _mod("diffusers.models.lora")
    lo.LoRACompatibleConv, lo.LoRACompatibleLinear = LoRACompatibleConv, LoRACompatibleLinear
    mu = _mod("diffusers.models.modeling_utils")
    mu.ModelMixin = ModelMixin
    mu.load_state_dict = lambda path, variant=None: torch.load(path, map_location="cpu")
    _mod("diffusers.models.modeling_pytorch_flax_utils").load_flax_checkpoint_in_pytorch_model = None
    rs = _mod("diffusers.models.resnet")
    rs.Downsample2D, rs.ResnetBlock2D, rs.Upsample2D = Downsample2D, ResnetBlock2D, Upsample2D
    rs.TemporalConvLayer = type("TemporalConvLayer", (nn.Module,), {})
    ap = _mod("diffusers.models.attention_processor")
    ap.Attention, ap.AttnProcessor, ap.AttnProcessor2_0, ap.XFormersAttnProcessor = Attention, AttnProcessor, AttnProcessor2_0, XFormersAttnProcessor
    ap.AttentionProcessor = object
    at = _mod("diffusers.models.attention")
    at.AdaLayerNorm, at.AdaLayerNormZero, at.FeedForward = AdaLayerNorm, AdaLayerNormZero, FeedForward
    at.BasicTransformerBlock = DiffusersBasicTransformerBlock
    t2 = _mod("diffusers.models.transformer_2d")
    t2.Transformer2DModel, t2.Transformer2DModelOutput = DiffusersTransformer2DModel, Transformer2DModelOutput
    tt = _mod("diffusers.models.transformer_temporal")
    tt.TransformerTemporalModelOutput = TransformerTemporalModelOutput
    tt.TransformerTemporalModel = type("TransformerTemporalModel", (nn.Module,), {})
    u3 = _mod("diffusers.models.unet_3d_condition")
    u3.UNet3DConditionModel = type("UNet3DConditionModel", (nn.Module,), {})
    u3.UNet3DConditionOutput = UNet3DConditionOutputShim
    _mod("diffusers.schedulers")
    su = _mod("diffusers.schedulers.scheduling_utils")
    su.KarrasDiffusionSchedulers, su.SchedulerMixin = KarrasDiffusionSchedulers, SchedulerMixin
    sdm = _mod("diffusers.schedulers.scheduling_ddim")
    sdm.DDIMScheduler, sdm.DDIMSchedulerOutput = DiffusersDDIMScheduler, DDIMSchedulerOutput
    sdm.betas_for_alpha_bar = sdm.rescale_zero_terminal_snr = lambda *a, **k: (_ for _ in ()).throw(NotImplementedError())
    sem = _mod("diffusers.schedulers.scheduling_euler_discrete")
    sem.EulerDiscreteScheduler, sem.EulerDiscreteSchedulerOutput = DiffusersEulerDiscreteScheduler, EulerDiscreteSchedulerOutput
    _mod("diffusers.utils.torch_utils").randn_tensor = randn_tensor
    xf = _mod("xformers")
    xo = _mod("xformers.ops")
    xo.memory_efficient_attention = memory_efficient_attention
    xf.ops = xo
    _mod("mmcm")
    _mod("mmcm.utils")
    _mod("mmcm.utils.itertools_util").generate_sample_idxs = generate_sample_idxs
    _mod("mmcm.utils.gpu_util").get_gpu_status = lambda *a, **k: ""
    acc = _mod("accelerate")
    _mod("accelerate.utils").set_module_tensor_to_device = None
    _mod("accelerate.utils.versions").is_torch_version = lambda op, v: True
    _install_referencenet_extras()
    _install_controlnet_extras()
    if reference_root not in sys.path:
        sys.path.insert(0, reference_root)


def _install_controlnet_extras() -> None:
    """names musev/models/controlnet.py imports at module top; only PoseGuider (plain torch + ModelMixin) is executed, the
    ControlNet wrappers around the un-vendored diffusers ControlNetModel are placeholders"""
    _mod("diffusers.models.controlnet").ControlNetModel = type("ControlNetModel", (nn.Module,), {})
    _mod("diffusers.pipelines")
    _mod("diffusers.pipelines.controlnet")
    _mod("diffusers.pipelines.controlnet.multicontrolnet").MultiControlNetModel = type("MultiControlNetModel", (nn.Module,), {})
    su = _mod("diffusers.schedulers.scheduling_utils")
    if not hasattr(su, "KarrasDiffusionSchedulers"):
        su.KarrasDiffusionSchedulers = type("KarrasDiffusionSchedulers", (), {})
    tu = _mod("diffusers.utils.torch_utils")
    if not hasattr(tu, "is_compiled_module"):
        tu.is_compiled_module = lambda m: False
    import PIL.Image  # noqa: F401  (controlnet.py does `import PIL` and uses PIL.Image in annotations)


def _install_referencenet_extras() -> None:
    """names that musev/models/referencenet.py and musev/models/unet_2d_blocks.py import from diffusers.  Everything the
    SD-1.5 ReferenceNet configuration does not instantiate is an empty placeholder class."""
    def stub(name):
        return type(name, (nn.Module,), {})

    ap = _mod("diffusers.models.attention_processor")
    for n in ("AttnAddedKVProcessor", "AttnAddedKVProcessor2_0", "AttnProcessor", "AttentionProcessor"):
        if not hasattr(ap, n):
            setattr(ap, n, type(n, (), {}))
    for n in ("ADDED_KV_ATTENTION_PROCESSORS", "CROSS_ATTENTION_PROCESSORS"):
        if not hasattr(ap, n):
            setattr(ap, n, ())
    _mod("diffusers.models.dual_transformer_2d").DualTransformer2DModel = stub("DualTransformer2DModel")
    _mod("diffusers.models.normalization").AdaGroupNorm = stub("AdaGroupNorm")
    rs = _mod("diffusers.models.resnet")
    for n in ("FirDownsample2D", "FirUpsample2D", "KDownsample2D", "KUpsample2D"):
        if not hasattr(rs, n):
            setattr(rs, n, stub(n))
    u2 = _mod("diffusers.models.unet_2d_blocks")
    for n in ("AttnDownBlock2D AttnDownEncoderBlock2D AttnSkipDownBlock2D AttnSkipUpBlock2D AttnUpBlock2D AttnUpDecoderBlock2D "
              "DownEncoderBlock2D KCrossAttnDownBlock2D KCrossAttnUpBlock2D KDownBlock2D KUpBlock2D ResnetDownsampleBlock2D "
              "ResnetUpsampleBlock2D SimpleCrossAttnDownBlock2D SimpleCrossAttnUpBlock2D SkipDownBlock2D SkipUpBlock2D "
              "UpDecoderBlock2D").split():
        setattr(u2, n, stub(n))
    tu = _mod("diffusers.utils.torch_utils")
    tu.apply_freeu = lambda *a, **k: (_ for _ in ()).throw(NotImplementedError())
    if not hasattr(tu, "maybe_allow_in_graph"):
        tu.maybe_allow_in_graph = lambda c: c
    uc = _mod("diffusers.models.unet_2d_condition")

    @dataclass
    class UNet2DConditionOutput(BaseOutput):
        sample: torch.FloatTensor = None

    uc.UNet2DConditionModel = type("UNet2DConditionModel", (ModelMixin, ConfigMixin), {})
    uc.UNet2DConditionOutput = UNet2DConditionOutput
    ut = _mod("diffusers.utils")
    _mod("diffusers.utils.deprecation_utils").deprecate = lambda *a, **k: None
    pu = _mod("diffusers.utils.peft_utils")
    pu.scale_lora_layers = pu.unscale_lora_layers = lambda *a, **k: None
    for n, v in (("deprecate", lambda *a, **k: None), ("scale_lora_layers", lambda *a, **k: None),
                 ("unscale_lora_layers", lambda *a, **k: None), ("USE_PEFT_BACKEND", False), ("BaseOutput", BaseOutput)):
        if not hasattr(ut, n):
            setattr(ut, n, v)
    mu = _mod("diffusers.models.modeling_utils")
    if not hasattr(mu, "load_state_dict"):
        mu.load_state_dict = lambda *a, **k: None
    em = _mod("diffusers.models.embeddings")
    for n in ("GaussianFourierProjection ImageHintTimeEmbedding ImageProjection ImageTimeEmbedding PositionNet "
              "TextImageProjection TextImageTimeEmbedding TextTimeEmbedding").split():
        if not hasattr(em, n):
            setattr(em, n, stub(n))
    act = _mod("diffusers.models.activations")
    if not hasattr(act, "get_activation"):
        act.get_activation = lambda name: {"silu": nn.SiLU(), "swish": nn.SiLU(), "mish": nn.Mish(), "gelu": nn.GELU(),
                                           "relu": nn.ReLU()}[name]
