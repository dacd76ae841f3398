"""The steps either side of the denoise loop: temporally chunked VAE decode of the final latents and the
predictor's multi-shot loop ("infinite length": every shot is conditioned on the last frames of the previous one).

Reference: ``MusevControlNetPipeline.decode_latents`` / the chunked decode at the end of ``__call__``
(musev/pipelines/pipeline_controlnet.py:233-238, 2157-2171) and ``MusevControlNetPredictor.run_pipe_text2video``'s shot loop
(musev/pipelines/pipeline_controlnet_predictor.py:643-745).  Host orchestration only; the compute is
musev_amd.models.vae.AutoencoderKL (HIP kernels) and musev_amd.pipelines.parallel_denoise.ParallelDenoiser."""
from __future__ import annotations

from typing import Callable, List, Optional

import torch

__all__ = ["decode_latents", "multi_shot_denoise"]

_MAX_CALL_BYTES = 3 << 29  # 1.5 GiB: every operand of one kernel call must span < 2 GiB (32-bit buffer offsets)


def max_frames_per_call(block_out_channels, h: int, w: int) -> int:
    """frames one ``vae.decode`` call may carry so that every kernel operand stays below the 2 GiB limit.  The widest activation
    at resolution level k (8h / 2^k x 8w / 2^k) is NOT C_k wide: an up block's Upsample2D emits the next finer resolution with the
    COARSER level's channel count (up_blocks[2] of the SD VAE: 8h x 8w x block_out_channels[1] = 256 channels, which feeds
    up_blocks[3].resnets[0].norm1 / conv1 / conv_shortcut) -- level k holds block_out_channels[min(k + 1, last)] channels."""
    last = len(block_out_channels) - 1
    per_frame = max(((8 * h) >> k) * ((8 * w) >> k) * block_out_channels[min(k + 1, last)] * 2 for k in range(last + 1))
    return max(1, _MAX_CALL_BYTES // per_frame)


@torch.no_grad()
def decode_latents(vae, latents: torch.Tensor, decoder_t_segment: int = 200) -> torch.Tensor:
    """latents [b, c, t, h, w] -> video [b, 3, t, 8h, 8w] fp32 in [0, 1] (the reference returns the same values as a numpy array):
    ``latents / scaling_factor`` -> ``vae.decode`` -> ``(x / 2 + 0.5).clamp(0, 1)``, in slices of ``decoder_t_segment`` frames
    along t exactly as the reference (:2157-2171).  A slice is further cut so that the largest activation of one kernel call
    (``max_frames_per_call``) stays below the kernels' 2 GiB operand limit; results do not depend on the slicing."""
    if latents.ndim != 5:
        raise ValueError("latents must be [b, c, t, h, w]")
    b, c, t, h, w = latents.shape
    max_frames = max_frames_per_call(vae.config.block_out_channels, h, w)
    out: List[torch.Tensor] = []
    for s0 in range(0, t, decoder_t_segment):
        seg = latents[:, :, s0:s0 + decoder_t_segment]
        f = seg.shape[2]
        z = (seg.permute(0, 2, 1, 3, 4).reshape(b * f, c, h, w).float() / vae.config.scaling_factor)   # "b c f h w -> (b f) c h w"
        imgs = [vae.decode(z[i:i + max_frames])[0] for i in range(0, b * f, max_frames)]
        img = (torch.cat(imgs, dim=0) / 2 + 0.5).clamp_(0, 1)
        out.append(img.reshape(b, f, img.shape[1], img.shape[2], img.shape[3]).permute(0, 2, 1, 3, 4))  # "(b f) c h w -> b c f h w"
    return torch.cat(out, dim=2)


@torch.no_grad()
def multi_shot_denoise(denoiser, make_noise: Callable[[int], torch.Tensor], prompt_embeds: torch.Tensor, *,
                       condition_latents: Optional[torch.Tensor], n_vision_condition: int = 1, max_batch_num: int = 1,
                       fix_condition_images: bool = False, vae=None, decoder_t_segment: int = 200,
                       on_shot: Optional[Callable[[int, torch.Tensor], None]] = None, **loop_kwargs):
    """``run_pipe_text2video``'s shot loop (:643-745): shot 0 is conditioned on ``condition_latents`` ([1, c, n_vision_condition, h,
    w]); every later shot on the LAST ``n_vision_condition`` latent frames of the previous shot (unless ``fix_condition_images``),
    and its leading ``n_vision_condition`` output frames -- the re-inserted condition frames -- are dropped before concatenation
    (``result_overlap``).  ``make_noise(shot)`` returns the initial latents [1, c, video_length, h, w] of a shot (the reference draws
    them inside the pipeline from the caller's generator; musev_amd.utils.noise_util restates the three noise types).
    ``loop_kwargs`` go to the denoiser (num_inference_steps, guidance_scale, unet_kwargs, group, ...).
    Returns (latents [1, c, T_total, h, w], video [1, 3, T_total, 8h, 8w] or None when no ``vae`` is given)."""
    if max_batch_num < 1:
        raise ValueError("max_batch_num must be >= 1")
    if condition_latents is not None and condition_latents.shape[2] != n_vision_condition:
        raise ValueError("condition_latents must hold n_vision_condition frames")
    lat_parts, vid_parts = [], []
    cond = condition_latents
    for shot in range(max_batch_num):
        out = denoiser(make_noise(shot), prompt_embeds, condition_latents=cond, **loop_kwargs)  # [1, c, n_cond + T, h, w]
        overlap = 0 if shot == 0 else (n_vision_condition if cond is not None else 0)
        keep = out[:, :, overlap:]
        lat_parts.append(keep)
        if vae is not None:
            vid_parts.append(decode_latents(vae, keep, decoder_t_segment))
        if on_shot is not None:
            on_shot(shot, keep)
        if cond is not None and n_vision_condition > 0 and not fix_condition_images:
            cond = out[:, :, -n_vision_condition:].clone()   # out_latents_batch[:, :, -n_vision_condition:] (:656-659)
    latents = torch.cat(lat_parts, dim=2)
    video = torch.cat(vid_parts, dim=2) if vid_parts else None
    return latents, video
