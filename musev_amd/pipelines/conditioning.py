"""Once-per-call conditioning glue around the denoise loop (reference musev/pipelines/pipeline_controlnet.py): how the
pipeline turns the reference-image latents into the ``down_block_refer_embs`` / ``mid_block_refer_emb`` inputs of the UNet.
Runs before the hot loop; the heavy part is ReferenceNet2D (musev_amd.models.referencenet, HIP kernels)."""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch

__all__ = ["cfg_refer_image_latents", "get_referencenet_emb", "broadcast_side_model_outputs", "get_referencenet_emb_sharded"]


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


def broadcast_side_model_outputs(tensors: Optional[List[Optional[torch.Tensor]]], group, src: int = 0, device=None
                                 ) -> List[Optional[torch.Tensor]]:
    """Once-per-call outputs of a side model (ReferenceNet features, ControlNet-independent conditioning, CLIP / IP-Adapter
    tokens) computed on rank ``src`` only and handed to every rank of ``group`` ("ReferenceNet/CLIP/VAE run on rank 0
    and are broadcast once", ~40 MB of reference features at 512x512).  The shapes travel as one small host object, the payload as
    ONE flat buffer per dtype -> one RCCL broadcast instead of 13 (xGMI is per-link bound: few, large messages).
    ``tensors`` is the list on rank ``src`` (entries may be None) and ignored elsewhere."""
    import torch.distributed as dist
    rank = dist.get_rank(group)
    src_global = dist.get_global_rank(group, src) if hasattr(dist, "get_global_rank") else src
    meta = [None]
    if rank == src:
        meta[0] = None if tensors is None else [None if t is None else (tuple(t.shape), str(t.dtype).split(".")[-1]) for t in tensors]
    dist.broadcast_object_list(meta, src=src_global, group=group)
    if meta[0] is None:
        return []
    if rank == src:
        device = next(t.device for t in tensors if t is not None)
    out: List[Optional[torch.Tensor]] = [None] * len(meta[0])
    for dt in sorted({m[1] for m in meta[0] if m is not None}):
        idxs = [i for i, m in enumerate(meta[0]) if m is not None and m[1] == dt]
        numels = [int(torch.Size(meta[0][i][0]).numel()) for i in idxs]
        tdt = getattr(torch, dt)
        if rank == src:
            flat = torch.cat([tensors[i].reshape(-1) for i in idxs])
        else:
            flat = torch.empty(sum(numels), dtype=tdt, device=device)
        dist.broadcast(flat, src=src_global, group=group)
        off = 0
        for i, n in zip(idxs, numels):
            out[i] = flat[off:off + n].view(meta[0][i][0])
            off += n
    return out


def get_referencenet_emb_sharded(referencenet, refer_image_vae_emb, n_refer_image: int, ip_adapter_image_emb, prompt_embeds,
                                 group=None, src: int = 0, device=None):
    """``get_referencenet_emb`` for a sharded run: the ReferenceNet forward happens on rank ``src`` only (the other ranks may pass
    ``referencenet=None`` and need not hold its weights), its 12 + 1 feature maps are broadcast once.  group=None: plain call."""
    if group is None:
        return get_referencenet_emb(referencenet, refer_image_vae_emb, n_refer_image, ip_adapter_image_emb, prompt_embeds)
    import torch.distributed as dist
    payload = None
    if dist.get_rank(group) == src:
        down, mid, _ = get_referencenet_emb(referencenet, refer_image_vae_emb, n_refer_image, ip_adapter_image_emb, prompt_embeds)
        payload = None if down is None else list(down) + [mid]
    got = broadcast_side_model_outputs(payload, group, src=src, device=device)
    if not got:
        return None, None, None
    return got[:-1], got[-1], None
