"""Initial-noise construction of the denoise loop (reference musev/utils/noise_util.py:9-83 and the noise branch of
MusevControlNetPipeline.prepare_latents, pipeline_controlnet.py:300-345,407-410).  Runs once per call, before the hot loop;
plain torch on whatever device the generator lives on.  The draws consume the generator in the reference's order
(common noise first, then the per-frame noise), so a seeded run starts from the same latents as the reference."""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple, Union

import torch

__all__ = ["random_noise", "video_fusion_noise", "img_based_video_noise", "prepare_noise_latents"]

Gen = Optional[Union[torch.Generator, List[torch.Generator]]]


def _randn(shape: Sequence[int], generator, device, dtype) -> torch.Tensor:
    """diffusers.utils.torch_utils.randn_tensor: draw on the generator's device (CPU generators keep seeded runs
    reproducible across devices), then move; a list of generators draws one batch item each."""
    device = torch.device(device) if device is not None else torch.device("cpu")
    if isinstance(generator, (list, tuple)):
        if len(generator) == 1:
            generator = generator[0]
        else:
            parts = [_randn((1, *shape[1:]), g, device, dtype) for g in generator]
            return torch.cat(parts, dim=0)
    draw_dev = generator.device if generator is not None else device
    if draw_dev.type != device.type and draw_dev.type != "cpu":
        raise ValueError(f"Cannot generate a {device} tensor from a generator of type {draw_dev.type}.")
    return torch.randn(tuple(shape), generator=generator, device=draw_dev, dtype=dtype).to(device)


def _resolve(tensor, shape, dtype, device):
    if tensor is not None:
        shape, device, dtype = tensor.shape, tensor.device, tensor.dtype
    return tuple(shape), dtype, (torch.device(device) if isinstance(device, str) else device)


def random_noise(tensor: torch.Tensor = None, shape: Tuple[int, ...] = None, dtype: torch.dtype = None, device=None,
                 generator: Gen = None, noise_offset: Optional[float] = None) -> torch.Tensor:
    """N(0, 1) of ``shape`` (noise_util.py:9-28); ``noise_offset`` adds a per-(batch, channel) offset drawn from the
    global RNG, as upstream."""
    shape, dtype, device = _resolve(tensor, shape, dtype, device)
    noise = _randn(shape, generator, device, dtype)
    if noise_offset is not None:
        noise += noise_offset * torch.randn((shape[0], shape[1], 1, 1, 1), device=device)
    return noise


def video_fusion_noise(tensor: torch.Tensor = None, shape: Tuple[int, ...] = None, dtype: torch.dtype = None, device=None,
                       w_ind_noise: float = 0.5, generator: Gen = None, initial_common_noise: torch.Tensor = None) -> torch.Tensor:
    """VideoFusion decomposed noise for [b, c, t, h, w] latents (noise_util.py:31-83):
    sqrt(1 - w) * common[b, c, 1, h, w] + sqrt(w) * individual[b, c, t, h, w]."""
    shape, dtype, device = _resolve(tensor, shape, dtype, device)
    b, c, t, h, w = shape
    if isinstance(generator, list):
        if len(generator) != b:
            raise ValueError(f"You have passed a list of generators of length {len(generator)}, but requested an effective batch"
                             f" size of {b}. Make sure the batch size matches the length of the generators.")
        items = [video_fusion_noise(shape=(1, c, t, h, w), dtype=dtype, device=device, w_ind_noise=w_ind_noise, generator=g,
                                    initial_common_noise=initial_common_noise) for g in generator]
        return torch.cat(items, dim=0).to(device)
    if initial_common_noise is not None:
        common = initial_common_noise.to(device, dtype=dtype)
    else:
        common = _randn((b, c, 1, h, w), generator, device, dtype)
    individual = _randn(shape, generator, device, dtype)
    s = torch.tensor(w_ind_noise, device=device, dtype=dtype)
    return torch.sqrt(1 - s) * common + torch.sqrt(s) * individual


def img_based_video_noise(noise: torch.Tensor, condition_latents: torch.Tensor, img_weight: float = 1e-3) -> torch.Tensor:
    """``need_img_based_video_noise`` (pipeline_controlnet.py:325-344): blend the time-mean of the vision-condition latents
    into every frame's noise with weights sqrt(img_weight), sqrt(1 - img_weight)."""
    mean = condition_latents.mean(dim=2, keepdim=True).expand(-1, -1, noise.shape[2], -1, -1)
    return img_weight ** 0.5 * mean + (1 - img_weight) ** 0.5 * noise


def prepare_noise_latents(shape: Tuple[int, ...], *, dtype: torch.dtype, device, generator: Gen = None, noise_type: str = "random",
                          w_ind_noise: float = 0.5, initial_common_latent: torch.Tensor = None,
                          condition_latents: torch.Tensor = None, need_img_based_video_noise: bool = False,
                          img_weight: float = 1e-3, init_noise_sigma: float = 1.0) -> torch.Tensor:
    """The text2video branch of prepare_latents (``latents is None and image is None``): noise of the requested type,
    optionally image-based, scaled by the scheduler's ``init_noise_sigma`` (:311-323, 325-344, 407-409)."""
    if noise_type == "random":
        noise = random_noise(shape=shape, dtype=dtype, device=device, generator=generator)
    elif noise_type == "video_fusion":
        noise = video_fusion_noise(shape=shape, dtype=dtype, device=device, generator=generator, w_ind_noise=w_ind_noise,
                                   initial_common_noise=initial_common_latent)
    else:
        raise ValueError(f"noise_type must be 'random' or 'video_fusion', got {noise_type}")
    if need_img_based_video_noise and condition_latents is not None:
        noise = img_based_video_noise(noise, condition_latents.to(noise.device, noise.dtype), img_weight)
    return noise * init_noise_sigma
