"""Once-per-call conditioning glue around the denoise loop (reference musev/pipelines/pipeline_controlnet.py): how the
pipeline turns the reference-image latents into the ``down_block_refer_embs`` / ``mid_block_refer_emb`` inputs of the UNet.
Runs before the hot loop; the heavy part is ReferenceNet2D (musev_amd.models.referencenet, HIP kernels)."""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch

__all__ = ["cfg_refer_image_latents", "get_referencenet_emb"]


def cfg_refer_image_latents(refer_image_vae_emb: torch.Tensor, n_refer_image: int, do_classifier_free_guidance: bool) -> torch.Tensor:
    """``get_referencenet_image_vae_emb`` after the VAE (:838-859): with classifier-free guidance the unconditional half
    uses the SAME reference latents ("mode 3" of the reference) -- [(b t), c, h, w] -> [(2 b t), c, h, w], batch-major."""
    if not do_classifier_free_guidance:
        return refer_image_vae_emb
    bt, c, h, w = refer_image_vae_emb.shape
    if bt % n_refer_image != 0:
        raise ValueError("reference latents are not a whole number of (b, t) items")
    x = refer_image_vae_emb.reshape(bt // n_refer_image, n_refer_image, c, h, w)
    return torch.cat([x, x], dim=0).reshape(2 * bt, c, h, w)


def get_referencenet_emb(referencenet, refer_image_vae_emb: Optional[torch.Tensor], n_refer_image: int,
                         ip_adapter_image_emb: Optional[torch.Tensor], prompt_embeds: Optional[torch.Tensor]
                         ) -> Tuple[Optional[List[torch.Tensor]], Optional[torch.Tensor], Optional[list]]:
    """``MusevControlNetPipeline.get_referencenet_emb`` (:867-964): one ReferenceNet forward at timestep 0 on the (already
    CFG-duplicated) reference latents; it prefers the IP-Adapter image tokens over the text embeddings as its
    cross-attention input (:889-893), so the two CFG halves see different tokens ([proj(zeros), proj(clip(image))]) and get
    different features.  The token batch must equal the reference batch (b t), as in the
    reference (ReferenceNet2D raises ValueError otherwise).  Returns (down_block_refer_embs, mid_block_refer_emb, refer_self_attn_emb) as ``b c t h w``."""
    if referencenet is None or refer_image_vae_emb is None:
        return None, None, None
    tokens = ip_adapter_image_emb if ip_adapter_image_emb is not None else prompt_embeds
    if tokens is None:
        raise ValueError("ReferenceNet needs ip_adapter_image_emb or prompt_embeds as encoder_hidden_states")
    timestep = torch.zeros((), dtype=torch.long, device=refer_image_vae_emb.device)       # ref_timestep (:885)
    return referencenet(sample=refer_image_vae_emb, timestep=timestep, encoder_hidden_states=tokens,
                        num_frames=n_refer_image, return_ndim=5)
