"""UNet3DConditionModel: drop-in for ``musev.models.unet_3d_condition.UNet3DConditionModel``
(reference musev/models/unet_3d_condition.py:179-1740) whose forward runs entirely in hand-written gfx950 kernels.

Same constructor keywords (:213-258), forward signature (:773-803), return convention (tuple / UNet3DConditionOutput),
attributes used by the pipeline (.dtype, .device, .config.in_channels, .ip_adapter_cross_attn,
.set_skip_temporal_layers, .spatial_cross_attns) and the same state_dict keys, so checkpoints written for
the reference load with ``load_state_dict``.  There is no eager fallback: the module only runs on a GPU tensor.

Data layout: the network input [b, c, t, h, w] is converted once to channels-last fp16 rows [(b t h w), c]; every
block consumes and produces that layout; the output is converted back at the end (:1008, :1263)."""
from __future__ import annotations

from dataclasses import dataclass
from types import SimpleNamespace
from typing import Any, Dict, List, Literal, Optional, Tuple, Union

import torch
from torch import nn

from .. import ops
from . import Model_Register
from .attention_processor import (BaseIPAttnProcessor, NonParamReferenceIPXFormersAttnProcessor,  # noqa: F401 (registry)
                                  NonParamT2ISelfReferenceXFormersAttnProcessor, ReferEmbFuseAttention,
                                  T2IReferencenetIPAdapterXFormersAttnProcessor)
from .layers import HipModule, TimestepEmbedding, bump_pack_epoch, w16
from .resnet import TemporalConvLayer  # noqa: F401 (registry)
from .runtime import Ctx, Geo, tensor_key
from .temporal_transformer import TransformerTemporalModel  # noqa: F401 (registry)
from .transformer_2d import Transformer2DModel
from .unet_3d_blocks import UNetMidBlock3DCrossAttn, get_down_block, get_up_block


@dataclass
class UNet3DConditionOutput:
    """reference :166-176"""
    sample: torch.Tensor

    def __getitem__(self, i):
        return (self.sample,)[i]


class UNet3DConditionModel(HipModule):
    _supports_gradient_checkpointing = False

    def __init__(
        self,
        sample_size: Optional[int] = None,
        in_channels: int = 4,
        out_channels: int = 4,
        down_block_types: Tuple[str] = ("CrossAttnDownBlock3D", "CrossAttnDownBlock3D", "CrossAttnDownBlock3D", "DownBlock3D"),
        up_block_types: Tuple[str] = ("UpBlock3D", "CrossAttnUpBlock3D", "CrossAttnUpBlock3D", "CrossAttnUpBlock3D"),
        block_out_channels: Tuple[int] = (320, 640, 1280, 1280),
        layers_per_block: int = 2,
        downsample_padding: int = 1,
        mid_block_scale_factor: float = 1,
        act_fn: str = "silu",
        norm_num_groups: Optional[int] = 32,
        norm_eps: float = 1e-5,
        cross_attention_dim: int = 1024,
        attention_head_dim: Union[int, Tuple[int]] = 8,
        temporal_conv_block: str = "TemporalConvLayer",
        temporal_transformer: str = "TransformerTemporalModel",
        need_spatial_position_emb: bool = False,
        need_transformer_in: bool = True,
        need_t2i_ip_adapter: bool = False,
        need_adain_temporal_cond: bool = False,
        t2i_ip_adapter_attn_processor: str = "NonParamT2ISelfReferenceXFormersAttnProcessor",
        keep_vision_condtion: bool = False,
        use_anivv1_cfg: bool = False,
        resnet_2d_skip_time_act: bool = False,
        need_zero_vis_cond_temb: bool = True,
        norm_spatial_length: bool = False,
        spatial_max_length: int = 2048,
        need_refer_emb: bool = False,
        ip_adapter_cross_attn: bool = False,
        t2i_crossattn_ip_adapter_attn_processor: str = "T2IReferencenetIPAdapterXFormersAttnProcessor",
        need_t2i_facein: bool = False,
        need_t2i_ip_adapter_face: bool = False,
        need_vis_cond_mask: bool = False,
    ):
        super().__init__()
        cfg = {k: v for k, v in locals().items() if k not in ("self", "__class__")}
        self.config = SimpleNamespace(**cfg)  # stands in for diffusers' register_to_config
        # ---- input checks, same messages as the reference (:313-328) ----
        if len(down_block_types) != len(up_block_types):
            raise ValueError(f"Must provide the same number of `down_block_types` as `up_block_types`. `down_block_types`: {down_block_types}. `up_block_types`: {up_block_types}.")
        if len(block_out_channels) != len(down_block_types):
            raise ValueError(f"Must provide the same number of `block_out_channels` as `down_block_types`. `block_out_channels`: {block_out_channels}. `down_block_types`: {down_block_types}.")
        if not isinstance(attention_head_dim, int) and len(attention_head_dim) != len(down_block_types):
            raise ValueError(f"Must provide the same number of `attention_head_dim` as `down_block_types`. `attention_head_dim`: {attention_head_dim}. `down_block_types`: {down_block_types}.")
        if act_fn not in ("silu", "swish"):
            raise NotImplementedError("only act_fn='silu' (SD-1.5)")
        if need_spatial_position_emb:
            raise NotImplementedError("need_spatial_position_emb=True is unused by all shipped configs")
        if norm_num_groups is None:
            raise NotImplementedError("norm_num_groups=None")
        if mid_block_scale_factor != 1:
            raise NotImplementedError("mid_block_scale_factor != 1")

        self.keep_vision_condtion = keep_vision_condtion
        self.use_anivv1_cfg = use_anivv1_cfg
        self.sample_size = sample_size
        self.resnet_2d_skip_time_act = resnet_2d_skip_time_act
        self.need_zero_vis_cond_temb = need_zero_vis_cond_temb
        self.need_refer_emb = need_refer_emb
        self.ip_adapter_cross_attn = ip_adapter_cross_attn
        self.need_t2i_facein = need_t2i_facein
        self.need_t2i_ip_adapter_face = need_t2i_ip_adapter_face
        self.need_spatial_position_emb = need_spatial_position_emb
        self.need_transformer_in = need_transformer_in
        self.need_t2i_ip_adapter = need_t2i_ip_adapter
        self.need_adain_temporal_cond = need_adain_temporal_cond  # AdaIN is a no-op for 4-D inputs (Appendix B.1)
        self.t2i_ip_adapter_attn_processor = t2i_ip_adapter_attn_processor
        self.need_vis_cond_mask = need_vis_cond_mask
        self.layers_per_block = layers_per_block
        self.block_out_channels = block_out_channels

        self.conv_in = nn.Conv2d(in_channels, block_out_channels[0], 3, padding=1)
        time_embed_dim = block_out_channels[0] * 4
        self.time_embedding = TimestepEmbedding(block_out_channels[0], time_embed_dim)
        frame_embed_dim = block_out_channels[0] * 4
        # resolve block classes by name through the registry, like the reference (:382-395)
        tconv_cls = (Model_Register[temporal_conv_block]
                     if isinstance(temporal_conv_block, str) and temporal_conv_block.lower() != "none" else None)
        ttrans_cls = (Model_Register[temporal_transformer]
                      if isinstance(temporal_transformer, str) and temporal_transformer.lower() != "none" else None)
        self.frame_embedding = TimestepEmbedding(block_out_channels[0], frame_embed_dim) if temporal_transformer is not None else None

        if need_transformer_in and ttrans_cls is not None:
            self.transformer_in = ttrans_cls(num_attention_heads=attention_head_dim if isinstance(attention_head_dim, int) else attention_head_dim[0],
                                             attention_head_dim=block_out_channels[0] // (attention_head_dim if isinstance(attention_head_dim, int) else attention_head_dim[0]),
                                             in_channels=block_out_channels[0], num_layers=1, femb_channels=frame_embed_dim,
                                             cross_attention_dim=cross_attention_dim)
        if isinstance(attention_head_dim, int):
            attention_head_dim = (attention_head_dim,) * len(down_block_types)

        need_t2i_ip_adapter_param = (t2i_ip_adapter_attn_processor is not None
                                     and "NonParam" not in t2i_ip_adapter_attn_processor and need_t2i_ip_adapter)
        if need_t2i_ip_adapter_param:
            raise NotImplementedError("parametric T2I IP-Adapter self-attention is not used by any shipped flavour")

        if need_refer_emb:
            self.first_refer_emb_attns = ReferEmbFuseAttention(query_dim=block_out_channels[0], heads=attention_head_dim[0],
                                                               dim_head=block_out_channels[0] // attention_head_dim[0])
            self.mid_block_refer_emb_attns = ReferEmbFuseAttention(query_dim=block_out_channels[-1], heads=attention_head_dim[-1],
                                                                   dim_head=block_out_channels[-1] // attention_head_dim[-1])
        else:
            self.first_refer_emb_attns = None
            self.mid_block_refer_emb_attns = None

        common = dict(temb_channels=time_embed_dim, femb_channels=frame_embed_dim, resnet_eps=norm_eps,
                      resnet_groups=norm_num_groups, cross_attention_dim=cross_attention_dim,
                      temporal_conv_block=tconv_cls, temporal_transformer=ttrans_cls,
                      need_t2i_ip_adapter=need_t2i_ip_adapter_param, ip_adapter_cross_attn=ip_adapter_cross_attn,
                      need_t2i_facein=need_t2i_facein, need_t2i_ip_adapter_face=need_t2i_ip_adapter_face,
                      resnet_2d_skip_time_act=resnet_2d_skip_time_act)
        self.down_blocks = nn.ModuleList([])
        self.up_blocks = nn.ModuleList([])
        output_channel = block_out_channels[0]
        for i, down_block_type in enumerate(down_block_types):
            input_channel = output_channel
            output_channel = block_out_channels[i]
            is_final_block = i == len(block_out_channels) - 1
            self.down_blocks.append(get_down_block(
                down_block_type, num_layers=layers_per_block, in_channels=input_channel, out_channels=output_channel,
                add_downsample=not is_final_block, attn_num_head_channels=attention_head_dim[i],
                downsample_padding=downsample_padding, need_refer_emb=need_refer_emb, **common))
        self.mid_block = UNetMidBlock3DCrossAttn(in_channels=block_out_channels[-1],
                                                 attn_num_head_channels=attention_head_dim[-1], **common)
        self.num_upsamplers = 0
        rev_ch = list(reversed(block_out_channels))
        rev_heads = list(reversed(attention_head_dim))
        output_channel = rev_ch[0]
        for i, up_block_type in enumerate(up_block_types):
            is_final_block = i == len(block_out_channels) - 1
            prev_output_channel = output_channel
            output_channel = rev_ch[i]
            input_channel = rev_ch[min(i + 1, len(block_out_channels) - 1)]
            if not is_final_block:
                self.num_upsamplers += 1
            self.up_blocks.append(get_up_block(
                up_block_type, num_layers=layers_per_block + 1, in_channels=input_channel, out_channels=output_channel,
                prev_output_channel=prev_output_channel, add_upsample=not is_final_block,
                attn_num_head_channels=rev_heads[i], **common))
        self.conv_norm_out = nn.GroupNorm(num_channels=block_out_channels[0], num_groups=norm_num_groups, eps=norm_eps)
        self.conv_act = nn.SiLU()
        self.conv_out = nn.Conv2d(block_out_channels[0], out_channels, 3, padding=1)

        # which key/value segments the spatial attn1 uses is selected by the processor name, as in
        # hack_t2i_sd_layer_attn_with_ip (:116-137)
        reference_only = False
        if need_t2i_ip_adapter and t2i_ip_adapter_attn_processor is not None:
            proc = Model_Register[t2i_ip_adapter_attn_processor]
            reference_only = getattr(proc, "mode", "") == "self_reference"
        if ip_adapter_cross_attn and t2i_crossattn_ip_adapter_attn_processor is not None:
            if getattr(Model_Register[t2i_crossattn_ip_adapter_attn_processor], "mode", "") != "cross_ip":
                raise NotImplementedError(f"unsupported cross-attn processor {t2i_crossattn_ip_adapter_attn_processor}")
        for m in self.modules():
            if isinstance(m, Transformer2DModel):
                m.reference_only = reference_only
        self.insert_spatial_self_attn_idx()
        self.skip_refer_downblock_emb = False
        self._param_version = None
        self._collect = None  # tests: dict receiving named block outputs (channels-last rows + geometry)

    # ---- introspection used by the reference's loaders -------------------------------------------------
    @property
    def dtype(self) -> torch.dtype:
        return self.conv_in.weight.dtype

    @property
    def device(self) -> torch.device:
        return self.conv_in.weight.device

    def _spatial_blocks(self):
        """the reference's ``get_attns(include="attentions", exclude="temp_attentions")`` (:1699-1740), quirk included: its exclude
        test overwrites the include test, so EVERY BasicTransformerBlock whose module name lacks "temp_attentions" is listed --
        the spatial transformers and, for the `musev` flavour, the block of ``transformer_in`` -- sorted by name (:1684-1685).
        Index-based consumers (insert_spatial_self_attn_idx, the IP-Adapter loader) therefore see the reference's numbering."""
        from .attention import BasicTransformerBlock
        out = [(name, m) for name, m in self.named_modules() if isinstance(m, BasicTransformerBlock) and "temp_attentions" not in name]
        return sorted(out, key=lambda nb: nb[0])

    @property
    def spatial_self_attns(self):
        """(attn1 modules, their BasicTransformerBlocks) of the spatial transformers (reference :1677-1700)"""
        blocks = self._spatial_blocks()
        return [(n + ".attn1", b.attn1) for n, b in blocks], blocks

    @property
    def spatial_cross_attns(self):
        """used by ip_adapter_loader.update_unet_ip_adapter_cross_attn_param (ip_adapter_loader.py:308-340)"""
        blocks = self._spatial_blocks()
        return [(n + ".attn2", b.attn2) for n, b in blocks], blocks

    def insert_spatial_self_attn_idx(self):
        attns, blocks = self.spatial_self_attns
        self.self_attn_num = len(attns)
        for i, (_, layer) in enumerate(attns):
            layer.spatial_self_attn_idx = i
        for i, (_, layer) in enumerate(blocks):
            layer.spatial_self_attn_idx = i

    def set_skip_temporal_layers(self, valid: bool) -> None:
        """reference :1639-1661: every module exposing ``skip_temporal_layers`` gets the flag"""
        for m in self.modules():
            if hasattr(m, "skip_temporal_layers"):
                m.skip_temporal_layers = valid

    def enable_xformers_memory_efficient_attention(self, *a, **k):  # pipeline compatibility no-op
        return None

    def _apply(self, fn, *args, **kwargs):
        out = super()._apply(fn, *args, **kwargs)
        bump_pack_epoch()  # packed weights live on the old device / dtype
        return out

    def load_state_dict(self, *args, **kwargs):
        out = super().load_state_dict(*args, **kwargs)
        bump_pack_epoch()
        return out

    def param_epoch(self) -> int:
        """sum of the parameters' in-place version counters: changes when a weight is edited (LoRA merge, load_state_dict)"""
        plist = self.__dict__.get("_plist")
        if plist is None or self.__dict__.get("_plist_training") is not self.training:
            plist = list(self.parameters())
            self.__dict__["_plist"], self.__dict__["_plist_training"] = plist, self.training
        v = 0
        for p in plist:
            v += p._version
        return v

    def _check_param_versions(self):
        v = self.param_epoch()
        if v != self._param_version:
            if self._param_version is not None:
                bump_pack_epoch()  # some weight was edited in place (e.g. LoRA merge): re-pack lazily
            self._param_version = v

    # ---- forward ---------------------------------------------------------------------------------------
    @torch.no_grad()
    def forward(
        self,
        sample: torch.FloatTensor,
        timestep: Union[torch.Tensor, float, int],
        encoder_hidden_states: torch.Tensor,
        class_labels: Optional[torch.Tensor] = None,
        timestep_cond: Optional[torch.Tensor] = None,
        attention_mask: Optional[torch.Tensor] = None,
        cross_attention_kwargs: Optional[Dict[str, Any]] = None,
        down_block_additional_residuals: Optional[Tuple[torch.Tensor]] = None,
        mid_block_additional_residual: Optional[torch.Tensor] = None,
        return_dict: bool = True,
        sample_index: torch.LongTensor = None,
        vision_condition_frames_sample: torch.Tensor = None,
        vision_conditon_frames_sample_index: torch.LongTensor = None,
        sample_frame_rate: int = 10,
        skip_temporal_layers: bool = None,
        frame_index: torch.LongTensor = None,
        down_block_refer_embs: Optional[Tuple[torch.Tensor]] = None,
        mid_block_refer_emb: Optional[torch.Tensor] = None,
        refer_self_attn_emb: Optional[List[torch.Tensor]] = None,
        refer_self_attn_emb_mode: Literal["read", "write"] = "read",
        vision_clip_emb: torch.Tensor = None,
        ip_adapter_scale: float = 1.0,
        face_emb: torch.Tensor = None,
        facein_scale: float = 1.0,
        ip_adapter_face_emb: torch.Tensor = None,
        ip_adapter_face_scale: float = 1.0,
        do_classifier_free_guidance: bool = False,
        pose_guider_emb: torch.Tensor = None,
    ) -> Union[UNet3DConditionOutput, Tuple]:
        if self._device_check and not sample.is_cuda:
            raise RuntimeError("musev_amd.UNet3DConditionModel runs only on an MI355X (HIP) device; there is no CPU path")
        if sample.ndim != 5:
            raise ValueError(f"sample must be b c t h w, got ndim={sample.ndim}")
        b, _, t, h, w = sample.shape
        if sample.dtype == torch.float32 and ops.CARRY and pose_guider_emb is None and 18 * self.conv_in.in_channels <= 128:
            # an fp32 sample enters conv_in unrounded, as two fp16 halves (what the denoise loop does through window_gather)
            x_rows = ops.split_hi_lo(sample.permute(0, 2, 3, 4, 1).reshape(-1, sample.shape[1]))
        else:
            x_rows = ops.bcthw_to_bthwc(sample)
        rows = self.forward_rows(
            x_rows, b, t, h, w, timestep, encoder_hidden_states, class_labels=class_labels,
            timestep_cond=timestep_cond, attention_mask=attention_mask,
            down_block_additional_residuals=down_block_additional_residuals,
            mid_block_additional_residual=mid_block_additional_residual, sample_index=sample_index,
            vision_condition_frames_sample=vision_condition_frames_sample,
            vision_conditon_frames_sample_index=vision_conditon_frames_sample_index, sample_frame_rate=sample_frame_rate,
            skip_temporal_layers=skip_temporal_layers, frame_index=frame_index, down_block_refer_embs=down_block_refer_embs,
            mid_block_refer_emb=mid_block_refer_emb, refer_self_attn_emb=refer_self_attn_emb,
            refer_self_attn_emb_mode=refer_self_attn_emb_mode, vision_clip_emb=vision_clip_emb,
            ip_adapter_scale=ip_adapter_scale, face_emb=face_emb, ip_adapter_face_emb=ip_adapter_face_emb,
            ip_adapter_face_scale=ip_adapter_face_scale, pose_guider_emb=pose_guider_emb)
        out_dtype = sample.dtype if sample.dtype in (torch.float16, torch.float32) else torch.float32
        out = ops.bthwc_to_bcthw(rows, b, t, h, w, dtype=out_dtype)
        if not return_dict:
            return (out,)
        return UNet3DConditionOutput(sample=out)

    @torch.no_grad()
    def forward_rows(self, x: torch.Tensor, b: int, t: int, h: int, w: int, timestep, encoder_hidden_states: torch.Tensor, *,
                     class_labels=None, timestep_cond=None, attention_mask=None, down_block_additional_residuals=None,
                     mid_block_additional_residual=None, sample_index=None, vision_condition_frames_sample=None,
                     vision_conditon_frames_sample_index=None, sample_frame_rate=10, skip_temporal_layers=None,
                     frame_index=None, down_block_refer_embs=None, mid_block_refer_emb=None, refer_self_attn_emb=None,
                     refer_self_attn_emb_mode: str = "read", vision_clip_emb=None, ip_adapter_scale: float = 1.0, face_emb=None, ip_adapter_face_emb=None,
                     ip_adapter_face_scale: float = 1.0, pose_guider_emb=None, prefix_memo=None) -> torch.Tensor:
        """The network on channels-last rows: x fp16 [(b t h w), in_channels] -> fp32 [(b t h w), out_channels] (the
        fp32 accumulator of conv_out, unrounded: CFG and the scheduler amplify the prediction's last-bit error).
        Used directly by musev_amd.pipelines.parallel_denoise (which builds window inputs in this layout)."""
        for name, val in (("class_labels", class_labels), ("timestep_cond", timestep_cond), ("attention_mask", attention_mask),
                          ("vision_condition_frames_sample", vision_condition_frames_sample), ("frame_index", frame_index),
                          ("face_emb", face_emb)):
            if val is not None:
                raise NotImplementedError(f"{name} is outside the hot-path scope of this build (SURVEY.md 8)")
        if skip_temporal_layers is not None:
            self.set_skip_temporal_layers(skip_temporal_layers)
        self._check_param_versions()

        # a latent size that is not a multiple of 2^(number of upsamplers): the up path is told the size of the skip it has to meet
        # (forward_upsample_size, :841-849; upsample_size = down_block_res_samples[-1].shape[2:], :1209-1210)
        forward_upsample_size = any(s % (2 ** self.num_upsamplers) != 0 for s in (h, w))
        geo = Geo(b, t, h, w)
        dev = x.device
        ch0 = self.block_out_channels[0]

        # ---- 1. time embedding (:887-906) ----
        vis_idx = None
        if vision_conditon_frames_sample_index is not None:
            if torch.is_tensor(vision_conditon_frames_sample_index):  # the reference passes a LongTensor (device sync)
                vis_idx = [int(i) for i in vision_conditon_frames_sample_index.reshape(-1).tolist()]
            else:  # host ints: no device round trip (what the parallel-denoise loop passes; hipGraph-capture safe)
                vis_idx = [int(i) for i in vision_conditon_frames_sample_index]

        def embeddings():
            """timestep / frame embeddings and every block's projection of them: functions of (timestep, b, t, frame rate, condition
            positions) only -- the same for both CFG halves, so part of the shared prefix (runtime.PrefixMemo: the second half's forward
            takes the first half's tensors instead of walking the ~12 small launches again at the head of its critical path)"""
            if not torch.is_tensor(timestep):
                tt = torch.tensor([float(timestep)], dtype=torch.float32, device=dev)
            else:
                tt = timestep.to(device=dev, dtype=torch.float32).reshape(-1)
            tt = tt.expand(b).repeat_interleave(t).contiguous()  # one row per frame: emb.repeat_interleave(num_frames)
            emb = self.time_embedding.hip_forward(ops.timestep_embedding(tt, ch0), final_silu=self.use_anivv1_cfg)
            if self.keep_vision_condtion and t > 1 and sample_index is not None and vis_idx is not None:
                ops.zero_rows(emb, self._const_rows(tuple(bi * t + i for bi in range(b) for i in vis_idx), dev))
            temb = emb if self.resnet_2d_skip_time_act else ops.silu(emb)  # ResnetBlock2D applies SiLU unless skip_time_act
            # ---- frame embedding (:909-937): window-local positions ----
            femb_a = None
            if self.frame_embedding is not None:
                fi = torch.arange(t, dtype=torch.float32, device=dev)
                if self.use_anivv1_cfg:
                    fi = (torch.arange(t, device=dev) * sample_frame_rate).to(dtype=torch.long).to(torch.float32)
                fi = fi.repeat(b).contiguous()  # rows (b, t)
                femb = self.frame_embedding.hip_forward(ops.timestep_embedding(fi, ch0), final_silu=self.use_anivv1_cfg)
                femb_a = ops.silu(femb)  # TransformerTemporalModel.nonlinearity (temporal_transformer.py:247-249)
            proj = self._batched_emb_proj(temb, femb_a)
            # (a flat tuple: the caller marks every tensor of a memo entry as used by the second half's stream)
            return (temb, femb_a, proj) + tuple(proj.values())

        emb_all = embeddings() if prefix_memo is None else prefix_memo.get("embeddings", embeddings)
        temb_act, femb_act, emb_proj = emb_all[0], emb_all[1], emb_all[2]

        # ---- conditioning rows ----
        if encoder_hidden_states.ndim != 3:
            raise NotImplementedError("only 3-D encoder_hidden_states [b, n, q] (per-frame 4-D text is outside this build)")
        if encoder_hidden_states.shape[0] != b:
            raise ValueError(f"encoder_hidden_states batch {encoder_hidden_states.shape[0]} != sample batch {b}")
        text = encoder_hidden_states.to(dtype=torch.float16).reshape(-1, encoder_hidden_states.shape[-1]).contiguous()
        clip, clip_len = None, 0
        if self.ip_adapter_cross_attn and vision_clip_emb is not None:
            if vision_clip_emb.ndim != 3 or vision_clip_emb.shape[0] != b:
                raise NotImplementedError("vision_clip_emb must be [b, n, q]")
            clip = vision_clip_emb.to(dtype=torch.float16).reshape(-1, vision_clip_emb.shape[-1]).contiguous()
            clip_len = vision_clip_emb.shape[1]
        refer_self = refer_self_write = None
        if refer_self_attn_emb is not None:
            # attention.py:261-289 ("read"): block i's reference-only self-attention also attends to the tokens of
            # refer_self_attn_emb[i] ([b, c, t, h, w]: a ReferenceNet's self-attention inputs).  "write" (attention.py:240-259,
            # transformer_2d.py:340-359) is the producer side of the same list: block i leaves the input of its self-attention in
            # the CALLER'S list as [(b t), c, h, w] (no shipped flavour runs the UNet3D in this mode).
            mode = str(refer_self_attn_emb_mode).lower()
            if mode not in ("read", "write"):
                raise ValueError(f"refer_self_attn_emb_mode {refer_self_attn_emb_mode!r}")
            if not hasattr(self._spatial_blocks()[0][1], "spatial_self_attn_idx"):
                raise ValueError("must call unet.insert_spatial_self_attn_idx to generate spatial attn index")
            if mode == "read":
                refer_self = list(refer_self_attn_emb)
            else:
                if not isinstance(refer_self_attn_emb, list):
                    raise ValueError("refer_self_attn_emb_mode='write' fills the list it is given: pass a list (one slot per spatial block)")
                refer_self_write = refer_self_attn_emb
        face, face_len = None, 0
        if self.need_t2i_ip_adapter_face and ip_adapter_face_emb is not None:   # (:1000-1006; ignored by models built without it)
            if ip_adapter_face_emb.ndim != 3 or ip_adapter_face_emb.shape[0] != b:
                raise NotImplementedError("ip_adapter_face_emb must be [b, n, q]")
            face = ip_adapter_face_emb.to(dtype=torch.float16).reshape(-1, ip_adapter_face_emb.shape[-1]).contiguous()
            face_len = ip_adapter_face_emb.shape[1]
        ctx = Ctx(emb_proj=emb_proj, temb_act=temb_act, femb_act=femb_act, text=text, text_len=encoder_hidden_states.shape[1], vis_idx=vis_idx,
                  clip=clip, clip_len=clip_len, ip_scale=float(ip_adapter_scale), skip_temporal=False,
                  text_src=encoder_hidden_states, clip_src=vision_clip_emb, face=face, face_len=face_len,
                  face_scale=float(ip_adapter_face_scale), face_src=ip_adapter_face_emb, refer_self=refer_self, refer_self_write=refer_self_write,
                  memo=prefix_memo)

        # ---- 2. pre-process (:1008-1063) ----
        pose = None
        if pose_guider_emb is not None:
            ctx.split()  # a per-half conditioning tensor: nothing is shared between the CFG halves (runtime.PrefixMemo)
            pose = pose_guider_emb.to(torch.float16).permute(0, 2, 3, 1).reshape(geo.rows, ch0).contiguous()
        w_in = self.packed("conv_in", lambda: ops.pack_conv_weight(self.conv_in.weight.detach()))
        x_in = x
        if x.shape[1] == 2 * self.conv_in.in_channels and 18 * self.conv_in.in_channels <= 128:
            # the fp32 sample as two fp16 halves, rows [hi | lo] (ops.window_gather(hi_lo=True) / ops.split_hi_lo): the convolution
            # is linear, so conv_in over the 2 C channels with its weight duplicated convolves the UNROUNDED input
            w_in2 = self.packed("conv_in_hilo", lambda: ops.pad_cols(ops.pack_conv_weight(
                torch.cat([self.conv_in.weight.detach(), self.conv_in.weight.detach()], dim=1)), 128))
            x = ctx.shared("conv_in", lambda: ops.conv3x3_cin_small_gemm(x_in, w_in2, w16(self.conv_in.bias), geo.n, h, w, add_=pose, kpad=128))
        elif 9 * self.conv_in.in_channels <= 64:  # latent input (4 channels): im2col + one MFMA K step
            w_in = self.packed("conv_in64", lambda: ops.pad_cols(w_in, 64))
            x = ctx.shared("conv_in", lambda: ops.conv3x3_cin_small_gemm(x_in, w_in, w16(self.conv_in.bias), geo.n, h, w, add_=pose))
        else:
            x = ctx.shared("conv_in", lambda: ops.conv3x3_cin_small(x_in, w_in, w16(self.conv_in.bias), geo.n, h, w, add_=pose))
        self._tap("conv_in", x, geo)
        if self.need_transformer_in:
            x_t = x
            x = ctx.shared("transformer_in", lambda: self.transformer_in.hip_forward(x_t, ctx, geo))
            self._tap("transformer_in", x, geo)
        use_refer = self.need_refer_emb and down_block_refer_embs is not None and not self.skip_refer_downblock_emb
        if use_refer:
            ctx.split()  # the ReferenceNet features are handed over per CFG half
            x = self.first_refer_emb_attns.hip_forward(x, down_block_refer_embs[0], geo)

        # ---- 3. down (:1076-1156) ----
        skips: List[torch.Tensor] = [x]
        skip_geos: List[Geo] = [geo]
        for i, blk in enumerate(self.down_blocks):
            refer = None
            if use_refer:
                is_final_block = i == len(self.block_out_channels) - 1
                num_block = self.layers_per_block + int(not is_final_block * 1)  # reference quirk (:1090-1095)
                start = 1 + num_block * i
                refer = down_block_refer_embs[start:start + num_block]
            x, geo, outs = blk.hip_forward(x, ctx, geo, refer)
            ctx.split()  # (at the latest: only the first down block's layers are wrapped in ctx.shared)
            skips.extend(o for o, _ in outs)
            skip_geos.extend(g_ for _, g_ in outs)
            for j, (o, g_) in enumerate(outs):
                self._tap(f"down_blocks.{i}.out{j}", o, g_)
        if down_block_additional_residuals is not None:
            skips = [ops.add(s, self._nchw_rows(r)) for s, r in zip(skips, down_block_additional_residuals)]

        # ---- 4. mid (:1159-1195) ----
        x = self.mid_block.hip_forward(x, ctx, geo)
        if self.mid_block_refer_emb_attns is not None and mid_block_refer_emb is not None and not self.skip_refer_downblock_emb:
            x = self.mid_block_refer_emb_attns.hip_forward(x, mid_block_refer_emb, geo)
        if mid_block_additional_residual is not None:
            x = ops.add(x, self._nchw_rows(mid_block_additional_residual))
        self._tap("mid", x, geo)

        # ---- 5. up (:1199-1245) ----
        for i, blk in enumerate(self.up_blocks):
            del skip_geos[-len(blk.resnets):]   # (the block pops its own skips; what is left on top is the size the next block works at)
            upsample_size = None
            if forward_upsample_size and i != len(self.up_blocks) - 1:
                upsample_size = (skip_geos[-1].h, skip_geos[-1].w)
            x, geo = blk.hip_forward(x, skips, ctx, geo, upsample_size)
            self._tap(f"up_blocks.{i}", x, geo)

        # ---- 6. post-process (:1258-1263) ----
        # (carry: the norm reads the residual stream's two fp16 halves and hands conv_out two halves as well -- the last layers of the
        # network see the identity path unrounded, ops.CARRY)
        x = ops.groupnorm(x, w16(self.conv_norm_out.weight), w16(self.conv_norm_out.bias), geo.n, geo.hw,
                          eps=self.conv_norm_out.eps, silu=True, groups=self.conv_norm_out.num_groups, carry=True)
        w_out = self.packed("conv_out", lambda: ops.pack_conv_weight(self.conv_out.weight.detach()))
        x = ops.conv3x3_cout_small(x, w_out, w16(self.conv_out.bias), geo.n, geo.h, geo.w, out_dtype=torch.float32)
        if skip_temporal_layers is not None:
            self.set_skip_temporal_layers(not skip_temporal_layers)
        return x

    def _batched_emb_proj(self, temb_act: torch.Tensor, femb_act: Optional[torch.Tensor]) -> Dict[int, torch.Tensor]:
        """time_emb_proj of every ResnetBlock2D (diffusers ResnetBlock2D: temb = time_emb_proj(act(temb))) and
        frame_emb_proj of every TransformerTemporalModel (temporal_transformer.py:247-251) in two GEMMs."""
        from .layers import ResnetBlock2D, lin_b, lin_w
        from .temporal_transformer import TransformerTemporalModel

        def pack(kind, attr):
            mods = [m for m in self.modules() if isinstance(m, kind)]
            if not mods:
                return None
            w = torch.cat([lin_w(getattr(m, attr)) for m in mods], 0).contiguous()
            b = torch.cat([lin_b(getattr(m, attr)) for m in mods], 0).contiguous()
            offs, o = [], 0
            for m in mods:
                n = getattr(m, attr).weight.shape[0]
                offs.append((id(m), o, n))
                o += n
            return w, b, offs

        out: Dict[int, torch.Tensor] = {}
        for name, kind, attr, src in (("tproj_all", ResnetBlock2D, "time_emb_proj", temb_act),
                                      ("fproj_all", TransformerTemporalModel, "frame_emb_proj", femb_act)):
            if src is None:
                continue
            pk = self.packed(name, lambda kind=kind, attr=attr: pack(kind, attr))
            if pk is None:
                continue
            w, b, offs = pk
            allp = ops.gemm(src, w, bias=b)
            for mid, o, n in offs:
                out[mid] = allp[:, o:o + n]
        return out

    def _const_rows(self, rows: tuple, dev: torch.device) -> torch.Tensor:
        """small constant index tensors, uploaded once (a host->device copy per forward would break graph capture)"""
        cache = self.__dict__.setdefault("_const_idx", {})
        key = (rows, str(dev))
        if key not in cache:
            cache[key] = torch.tensor(list(rows), dtype=torch.int32, device=dev)
        return cache[key]

    def _tap(self, name: str, x: torch.Tensor, geo: Geo) -> None:
        if self._collect is not None:
            self._collect[name] = x.float().reshape(geo.n, geo.h, geo.w, -1).permute(0, 3, 1, 2).cpu()

    @staticmethod
    def _nchw_rows(r: torch.Tensor) -> torch.Tensor:
        """ControlNet residual [(b t), c, h, w] -> channels-last fp16 rows (config 5 only; layout glue)"""
        return r.to(torch.float16).permute(0, 2, 3, 1).reshape(-1, r.shape[1]).contiguous()
