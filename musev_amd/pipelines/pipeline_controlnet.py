"""``MusevControlNetPipeline`` -- the reference's pipeline class (musev/pipelines/pipeline_controlnet.py:105-2215) as a thin host
object over the HIP hot path, so that code written against ``MusevControlNetPipeline.__call__`` (the predictor,
musev/pipelines/pipeline_controlnet_predictor.py:643-745, and through it scripts/inference/*.py) reaches the kernels WITHOUT an edit
inside the reference's pipeline: ``__call__`` keeps the reference's keyword list (:1295-1420), does the once-per-call preparation
in the reference's order (:1507-1846: timesteps, condition latents / indices, initial noise, guidance schedule, side-model
embeddings) and hands the denoise loop (:1847-2147) to ``ParallelDenoiser``; the chunked VAE decode (:2157-2171) is
``pipelines.video.decode_latents``.

Scope.  The loop, the UNet, ReferenceNet2D, ControlNet / PoseGuider, the VAE DECODER and the IP-Adapter image projection run here
on HIP kernels.  The ENCODERS of the reference pipeline -- CLIP text encoder + tokenizer, CLIP vision tower, VAE encoder -- are
outside the hot path and are injected as callables, or bypassed with the tensor the reference
itself accepts in their place:

    reference keyword                    what reaches the loop                         accepted here
    prompt / negative_prompt             prompt_embeds [2 b, 77, 768]                  ``prompt_embeds`` (+ ``negative_prompt_embeds``), or strings
                                                                                       with ``text_encoder=callable(prompts) -> embeds``
    condition_images                     condition_latents                             ``condition_latents``, or images with ``vae_encode=callable``
    ip_adapter_image                     vision_clip_emb [2 b, n_tok, 768]             ``ip_adapter_image_emb``, or images with ``image_encoder=callable``
    refer_image                          ReferenceNet feature maps                     ``refer_image_vae_emb`` (latents), or images with ``vae_encode``
    control_image                        ControlNet residuals / PoseGuider embedding   tensor [b, 3, t, H, W] in [0, 1] (the reference's prepare_image
                                                                                       output); PIL / numpy inputs are the caller's to convert

``image`` (+ ``strength``, ``add_latents_noise``): the img2img / video2video start of prepare_latents (:283-430) and get_timesteps
(:1627-1633) -- ``image`` goes through ``vae_encode``; the schedule then starts at step N - int(N * strength) when ``latents`` are given too.

Keywords of branches this package does not build raise ``NotImplementedError`` when they are USED (not when they are merely
passed with their default): FaceIn / IP-Adapter-FaceID (``refer_face_image``, ``ip_adapter_face_image``), the serial-denoise recording hooks
(``record_mid_video_*``, ``last_mid_video_*``), histogram matching and ``interpolation_factor``."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Any, Callable, Dict, List, Optional, Union

import torch

from ..utils.noise_util import prepare_noise_latents
from .conditioning import cfg_refer_image_latents, get_referencenet_emb_sharded
from .parallel_denoise import ParallelDenoiser
from .video import decode_latents

__all__ = ["MusevControlNetPipeline", "VideoPipelineOutput"]


@dataclass
class VideoPipelineOutput:
    """the reference's output record (pipeline_controlnet.py:69-77); ``videos`` [b, 3, t, H, W] in [0, 1]"""
    videos: Optional[torch.Tensor]
    latents: torch.Tensor
    videos_mid: list
    down_block_res_samples: Any = None
    mid_block_res_samples: Any = None
    up_block_res_samples: Any = None
    mid_video_latents: Optional[list] = None
    mid_video_noises: Optional[list] = None


def _unsupported(name: str, value, default=None) -> None:
    used = value is not None and value is not False and value != default
    if used:
        raise NotImplementedError(f"MusevControlNetPipeline(musev_amd): `{name}` belongs to a branch outside the HIP hot path "
                                  "(see the module docstring)")


class MusevControlNetPipeline:
    """Constructor keywords follow the reference (:112-170): ``vae``, ``unet``, ``scheduler``, ``controlnet``, ``referencenet``,
    ``pose_guider``, ``ip_adapter_image_proj``; the encoders are optional callables (module docstring).  ``group``: a
    torch.distributed process group to shard the (window, CFG half) units over (None = this process alone)."""

    def __init__(self, vae=None, unet=None, scheduler=None, controlnet=None, referencenet=None, pose_guider=None,
                 ip_adapter_image_proj=None, text_encoder: Optional[Callable] = None, vae_encode: Optional[Callable] = None,
                 image_encoder: Optional[Callable] = None, group=None, **_ignored):
        if unet is None:
            raise ValueError("MusevControlNetPipeline needs a unet")
        self.vae, self.unet, self.scheduler = vae, unet, scheduler
        self.controlnet, self.referencenet, self.pose_guider = controlnet, referencenet, pose_guider
        self.ip_adapter_image_proj = ip_adapter_image_proj
        self.text_encoder, self.vae_encode, self.image_encoder = text_encoder, vae_encode, image_encoder
        self.group = group
        self._denoisers: Dict[tuple, ParallelDenoiser] = {}
        self.print_idx = 0

    # ---- pieces of the reference's preparation ------------------------------------------------------------------------------
    def _denoiser(self, context_schedule, context_frames, context_stride, context_overlap, context_batch_size) -> ParallelDenoiser:
        key = (context_schedule, context_frames, context_stride, context_overlap, context_batch_size, id(self.scheduler))
        den = self._denoisers.get(key)
        if den is None:  # one loop object per window configuration: its captured graphs are reused by later calls
            den = self._denoisers[key] = ParallelDenoiser(self.unet, scheduler=self.scheduler, context_frames=context_frames,
                                                          context_overlap=context_overlap, context_stride=context_stride,
                                                          context_schedule=context_schedule, context_batch_size=context_batch_size)
        return den

    def _prompt_embeds(self, prompt, negative_prompt, prompt_embeds, negative_prompt_embeds, do_cfg: bool, device) -> torch.Tensor:
        """encode_weighted_prompt's contract (:1562-1576): [negative | positive] on dim 0 under classifier-free guidance"""
        if prompt_embeds is None:
            if self.text_encoder is None:
                raise ValueError("pass `prompt_embeds` (and `negative_prompt_embeds`), or construct the pipeline with "
                                 "text_encoder=callable(list of prompts) -> [n, 77, 768] embeddings")
            prompts = [prompt] if isinstance(prompt, str) else list(prompt)
            prompt_embeds = self.text_encoder(prompts)
            if do_cfg and negative_prompt_embeds is None:
                neg = negative_prompt if negative_prompt is not None else ""
                negs = [neg] * len(prompts) if isinstance(neg, str) else list(neg)
                negative_prompt_embeds = self.text_encoder(negs)
        prompt_embeds = prompt_embeds.to(device)
        if not do_cfg:
            return prompt_embeds
        if negative_prompt_embeds is None:
            if prompt_embeds.shape[0] % 2 != 0:
                raise ValueError("classifier-free guidance needs negative_prompt_embeds, or prompt_embeds already holding "
                                 "[negative | positive] on dim 0")
            return prompt_embeds  # already [negative | positive]
        return torch.cat([negative_prompt_embeds.to(device), prompt_embeds], dim=0)

    # ---- the call ------------------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def __call__(self, video_length: Optional[int], prompt: Union[str, List[str], None] = None, image=None, control_image=None,
                 condition_images=None, condition_latents: Optional[torch.Tensor] = None, latents: Optional[torch.Tensor] = None,
                 add_latents_noise: bool = False, height: Optional[int] = None, width: Optional[int] = None, strength: float = 0.8,
                 num_inference_steps: int = 50, guidance_scale: float = 7.5, guidance_scale_end: Optional[float] = None,
                 guidance_scale_method: str = "linear", negative_prompt=None, num_videos_per_prompt: Optional[int] = 1,
                 eta: float = 0.0, generator=None, controlnet_condition_images=None, controlnet_condition_latents=None,
                 controlnet_latents=None, prompt_embeds: Optional[torch.Tensor] = None,
                 negative_prompt_embeds: Optional[torch.Tensor] = None, output_type: Optional[str] = "tensor",
                 return_dict: bool = True, callback: Optional[Callable] = None, callback_steps: int = 1,
                 cross_attention_kwargs: Optional[Dict[str, Any]] = None, controlnet_conditioning_scale: float = 1.0,
                 guess_mode: bool = False, control_guidance_start: float = 0.0, control_guidance_end: float = 1.0,
                 need_middle_latents: bool = False, w_ind_noise: float = 0.5, initial_common_latent: Optional[torch.Tensor] = None,
                 latent_index=None, vision_condition_latent_index=None, noise_type: str = "random",
                 need_img_based_video_noise: bool = False, skip_temporal_layer: bool = False, img_weight: float = 1e-3,
                 need_hist_match: bool = False, motion_speed: float = 8.0, refer_image=None, ip_adapter_image=None,
                 refer_face_image=None, ip_adapter_scale: float = 1.0, facein_scale: float = 1.0, ip_adapter_face_scale: float = 1.0,
                 ip_adapter_face_image=None, prompt_only_use_image_prompt: bool = False, record_mid_video_noises: bool = False,
                 last_mid_video_noises=None, record_mid_video_latents: bool = False, last_mid_video_latents=None,
                 video_overlap: int = 1, context_schedule="uniform", context_frames=12, context_stride=1, context_overlap=4,
                 context_batch_size=1, interpolation_factor=1, decoder_t_segment: int = 200,
                 # tensors the reference computes with its encoders, accepted directly (module docstring)
                 ip_adapter_image_emb: Optional[torch.Tensor] = None, refer_image_vae_emb: Optional[torch.Tensor] = None):
        for name, value, default in (("refer_face_image", refer_face_image, None),
                                     ("ip_adapter_face_image", ip_adapter_face_image, None),
                                     ("record_mid_video_noises", record_mid_video_noises, False),
                                     ("last_mid_video_noises", last_mid_video_noises, None),
                                     ("record_mid_video_latents", record_mid_video_latents, False),
                                     ("last_mid_video_latents", last_mid_video_latents, None), ("need_hist_match", need_hist_match, False),
                                     ("controlnet_condition_images", controlnet_condition_images, None),
                                     ("controlnet_condition_latents", controlnet_condition_latents, None),
                                     ("controlnet_latents", controlnet_latents, None),
                                     ("cross_attention_kwargs", cross_attention_kwargs, None), ("interpolation_factor", interpolation_factor, 1),
                                     ("need_middle_latents", need_middle_latents, False)):
            _unsupported(name, value, default)
        if num_videos_per_prompt not in (None, 1):
            raise NotImplementedError("num_videos_per_prompt != 1 (the reference: 'may be wrong')")
        if eta != 0.0:
            raise NotImplementedError("eta != 0 (the reference scripts never pass it)")
        dev = next(self.unet.parameters()).device
        do_cfg = guidance_scale > 1.0                                                   # :1547
        embeds = self._prompt_embeds(prompt, negative_prompt, prompt_embeds, negative_prompt_embeds, do_cfg, dev)
        batch_size = embeds.shape[0] // (2 if do_cfg else 1)
        if batch_size != 1:
            raise NotImplementedError("the parallel-denoise loop runs one video per call (batch_size 1), like the reference's scripts")

        # ---- condition latents and their indices (prepare_condition_latents_and_index, :1633-1648) ----
        if condition_latents is None and condition_images is not None:
            if self.vae_encode is None:
                raise ValueError("`condition_images` needs vae_encode=callable(images) -> latents; or pass `condition_latents`")
            condition_latents = self.vae_encode(condition_images)
        n_cond = 0 if condition_latents is None else int(condition_latents.shape[2])
        # vision_condition_latent_index (-1 = the last of the n_cond + video_length frames, :995-1003) goes to the loop as it is;
        # latent_index is what the reference derives from it (:1017-1029) -- anything else it would only use for a shape (:1903-1907)
        vis_given = None if vision_condition_latent_index is None else [int(i) for i in (
            vision_condition_latent_index.tolist() if torch.is_tensor(vision_condition_latent_index) else vision_condition_latent_index)]
        if latent_index is not None and n_cond:
            total_frames = n_cond + int(video_length)
            vis_res = list(range(n_cond)) if vis_given is None else [i if i != -1 else total_frames - 1 for i in vis_given]
            if sorted(map(int, latent_index)) != sorted(set(range(total_frames)) - set(vis_res)):
                raise NotImplementedError("latent_index other than the positions the vision-condition frames leave free")

        # ---- scheduler, initial latents (:1613-1676) ----
        den = self._denoiser(context_schedule, context_frames, context_stride, context_overlap, context_batch_size)
        sched = den.scheduler
        sched.set_timesteps(num_inference_steps)
        c_lat = self.unet.config.in_channels if hasattr(self.unet, "config") else 4
        # ---- img2img / video2video start (:1627-1633 timesteps, :283-430 prepare_latents) ----
        # `image` (frames to start from) is VAE-encoded to init_latents; with `latents` also given, `strength` shortens the schedule
        # (get_timesteps of the img2img pipelines: the last int(N * strength) entries) and the caller's latents are used as they are
        # (x init_noise_sigma) or noised (`add_latents_noise`); with `image` alone the init latents are noised to the first timestep.
        # get_timesteps / add_noise live in the un-vendored diffusers base classes: restated from upstream, parity unpinned.
        start_step = 0
        user_latents = latents
        noise = None
        if user_latents is None or add_latents_noise:
            if height is None or width is None:
                if user_latents is None:
                    raise ValueError("height / width (pixels) are needed to draw the initial latents")
                shape = tuple(user_latents.shape)
            else:
                shape = (1, c_lat, int(video_length), height // 8, width // 8)
            noise = prepare_noise_latents(shape, dtype=torch.float32, device=dev, generator=generator, noise_type=noise_type,
                                          w_ind_noise=w_ind_noise, initial_common_latent=initial_common_latent,
                                          condition_latents=condition_latents if (image is None and user_latents is None) else None,
                                          need_img_based_video_noise=need_img_based_video_noise and image is None and user_latents is None,
                                          img_weight=img_weight, init_noise_sigma=1.0)
        if strength and image is not None and user_latents is not None:
            init_t = min(int(num_inference_steps * float(strength)), int(num_inference_steps))
            start_step = max(int(num_inference_steps) - init_t, 0)
        sigma0 = float(getattr(sched, "init_noise_sigma", 1.0))
        if user_latents is None:
            if image is None:
                latents = noise * sigma0
            else:
                if self.vae_encode is None:
                    raise ValueError("`image` needs vae_encode=callable(frames) -> latents [1, c, t, h, w]")
                init_latents = self.vae_encode(image).to(dev, torch.float32)
                if init_latents.shape[-2:] != noise.shape[-2:]:
                    init_latents = torch.nn.functional.interpolate(
                        init_latents.permute(0, 2, 1, 3, 4).reshape(-1, c_lat, *init_latents.shape[-2:]), size=tuple(noise.shape[-2:]),
                        mode="bilinear").reshape(1, -1, c_lat, *noise.shape[-2:]).permute(0, 2, 1, 3, 4)
                latents = sched.add_noise(init_latents, noise, start_step)
        else:
            latents = user_latents.to(dev, torch.float32)
            latents = sched.add_noise(latents, noise, start_step) if add_latents_noise else latents * sigma0
        latents = latents.to(dev)
        if condition_latents is not None:
            condition_latents = condition_latents.to(dev)
        lat_h, lat_w = latents.shape[-2:]

        # ---- side-model embeddings (:1733-1779) ----
        unet_kwargs: Dict[str, Any] = {}
        if ip_adapter_image_emb is None and ip_adapter_image is not None:
            if self.image_encoder is None:
                raise ValueError("`ip_adapter_image` needs image_encoder=callable(images) -> CLIP image embeddings; or pass `ip_adapter_image_emb`")
            clip = self.image_encoder(ip_adapter_image).to(dev)                                          # [n_img, n_tok, q]
            # get_ip_adapter_image_emb (:704-760): project, fold the images of one item into its token axis, and give the
            # unconditional half the projection of a ZERO embedding (plain zeros without a projection model)
            proj = self.ip_adapter_image_proj
            cond_emb = proj(clip) if proj is not None else clip
            cond_emb = cond_emb.reshape(1, -1, cond_emb.shape[-1])                                       # "(b t) n q -> b (t n) q", b = 1
            if do_cfg:
                unc = proj(torch.zeros_like(clip)) if proj is not None else torch.zeros_like(cond_emb)
                cond_emb = torch.cat([unc.reshape(1, -1, unc.shape[-1]), cond_emb], dim=0)
            ip_adapter_image_emb = cond_emb
        if ip_adapter_image_emb is not None:
            ip_adapter_image_emb = ip_adapter_image_emb.to(dev)
            if prompt_only_use_image_prompt and not getattr(self.unet, "ip_adapter_cross_attn", False):
                embeds = ip_adapter_image_emb                                                           # :1748-1753
            else:
                unet_kwargs["vision_clip_emb"] = ip_adapter_image_emb
                unet_kwargs["ip_adapter_scale"] = float(ip_adapter_scale)
        if refer_image_vae_emb is None and refer_image is not None:
            if self.vae_encode is None:
                raise ValueError("`refer_image` needs vae_encode=callable(images) -> latents; or pass `refer_image_vae_emb`")
            refer_image_vae_emb = self.vae_encode(refer_image)
        if refer_image_vae_emb is not None and self.referencenet is not None:
            ref = refer_image_vae_emb.to(dev)
            if ref.ndim == 5:                                                                            # b c t h w -> (b t) c h w
                n_ref = ref.shape[2]
                ref = ref.permute(0, 2, 1, 3, 4).reshape(-1, ref.shape[1], ref.shape[3], ref.shape[4])
            else:
                n_ref = ref.shape[0]
            ref = cfg_refer_image_latents(ref, n_ref, do_cfg)                                            # :838-859
            down, mid, _ = get_referencenet_emb_sharded(self.referencenet, ref, n_ref, ip_adapter_image_emb, embeds, group=self.group,
                                                        device=dev)
            unet_kwargs["down_block_refer_embs"], unet_kwargs["mid_block_refer_emb"] = down, mid

        controlnet = None
        ctrl = None
        if control_image is not None:
            if not torch.is_tensor(control_image) or control_image.ndim != 5:
                raise ValueError("control_image must be a tensor [1, 3, n_cond + video_length, H, W] in [0, 1] (prepare_image's output)")
            ctrl = control_image.to(dev)
            if self.pose_guider is not None:
                # :1776-1783 -- with a PoseGuider installed the ControlNet is not run (:1217); its embedding of ALL control frames goes
                # to every UNet call unchanged (:2066), i.e. the reference supports it for runs whose one window covers every frame
                emb = self.pose_guider(ctrl.to(next(self.pose_guider.parameters()).dtype))               # [1, C, n_cond + T, h, w]
                emb = emb.permute(0, 2, 1, 3, 4).reshape(-1, emb.shape[1], emb.shape[3], emb.shape[4])   # "b c t h w -> (b t) c h w"
                if emb.shape[0] != n_cond + int(video_length) or int(video_length) > int(context_frames):
                    raise ValueError("pose_guider: control_image must cover the condition + generated frames, all inside ONE window "
                                     "(the reference passes the whole embedding to every window's UNet call)")
                unet_kwargs["pose_guider_emb"] = emb.repeat(2, 1, 1, 1) if do_cfg else emb                # CFG-duplicated control image (:476-477)
                ctrl = None
            else:
                controlnet = self.controlnet
                if controlnet is None:
                    raise ValueError("control_image given but the pipeline has neither a controlnet nor a pose_guider")

        if skip_temporal_layer:
            self.unet.set_skip_temporal_layers(True)                                                    # :1713-1714
        try:
            cb = None
            if callback is not None:
                def cb(i, t, lat, _cb=callback):                                                         # :2141-2147
                    if i % callback_steps == 0:
                        _cb(i, t, lat)
            out = den(latents, embeds, num_inference_steps=num_inference_steps, guidance_scale=guidance_scale,
                      condition_latents=condition_latents, motion_speed=motion_speed, unet_kwargs=unet_kwargs, group=self.group,
                      callback=cb, guidance_scale_end=guidance_scale_end, guidance_scale_method=guidance_scale_method,
                      generator=generator, noise_type=noise_type, w_ind_noise=w_ind_noise, controlnet=controlnet, control_image=ctrl,
                      controlnet_conditioning_scale=float(controlnet_conditioning_scale), control_guidance_start=float(control_guidance_start),
                      control_guidance_end=float(control_guidance_end), guess_mode=bool(guess_mode), start_step=start_step,
                      vision_condition_latent_index=vis_given)
        finally:
            if skip_temporal_layer:
                self.unet.set_skip_temporal_layers(False)                                               # :2175-2176

        video = None
        if output_type != "latent" and self.vae is not None:
            video = decode_latents(self.vae, out, decoder_t_segment)                                    # :2157-2171
        self.print_idx += 1
        if not return_dict:
            return (video, out, [], None, None)
        return VideoPipelineOutput(videos=video, latents=out, videos_mid=[], mid_video_latents=None, mid_video_noises=None)
