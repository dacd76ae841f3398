"""Per-forward runtime context shared by the HIP-backed modules: tensor geometry, conditioning rows and caches.

Activations travel as fp16 2-D "rows x channels" tensors in (b, t, y, x) row order -- i.e. the canonical
[B, T, H, W, C] layout flattened.  All three views the reference materialises with permute copies
("(b t) c h w", "b c t h w", "(b h w) t c"; resnet.py:106-133, temporal_transformer.py:234-279) are index
arithmetic on this one buffer."""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Optional

import torch


@dataclass
class Geo:
    b: int
    t: int
    h: int
    w: int

    @property
    def n(self) -> int:  # frames through the spatial layers
        return self.b * self.t

    @property
    def hw(self) -> int:
        return self.h * self.w

    @property
    def rows(self) -> int:
        return self.b * self.t * self.h * self.w

    def down(self) -> "Geo":
        return Geo(self.b, self.t, (self.h + 2 - 3) // 2 + 1, (self.w + 2 - 3) // 2 + 1)

    def up(self) -> "Geo":
        return Geo(self.b, self.t, self.h * 2, self.w * 2)


class PrefixMemo:
    """The part of a window forward that does not depend on the CFG half, computed once for both halves.

    The loop hands both classifier-free-guidance halves the SAME latents (`latent_model_input = torch.cat([latents] * 2)`,
    pipeline_controlnet.py:1908-1910) and the same timestep; the halves only start to differ at the first text cross-attention.
    Everything before it -- the timestep / frame embeddings and every block's projection of them, conv_in, transformer_in, the first ResnetBlock2D / TemporalConvLayer, the first Transformer2DModel's
    norm / proj_in / reference-only self-attention / to_q -- is the same arithmetic on the same values twice.  The first half's
    forward RECORDS those tensors (``get`` computes and stores), the second half's forward REPLAYS them (``get`` returns the stored
    tensor) and starts computing at the split.  Bit-identical to computing them twice; the model calls ``split()`` before the first
    use of anything that may differ between the halves (text, per-half conditioning tensors), which closes the memo: from then on
    ``get`` just computes.  ``on_split`` lets the caller order its streams (the second half's stream waits there)."""

    def __init__(self):
        self.store = {}
        self.replaying = False
        self.closed = False
        self.on_split = None
        self.hits = 0

    def get(self, key, fn):
        if self.closed:
            return fn()
        if not self.replaying:
            v = self.store[key] = fn()
            return v
        self.hits += 1
        return self.store[key]

    def split(self) -> None:
        if self.closed:
            return
        self.closed = True
        if not self.replaying and self.on_split is not None:
            self.on_split()

    def replay(self) -> "PrefixMemo":
        """arm the memo for the second half's forward"""
        self.replaying, self.closed = True, False
        return self


@dataclass
class Ctx:
    """Everything a block needs besides its activations."""
    temb_act: torch.Tensor            # [N, 1280] fp16: act(temb) as consumed by ResnetBlock2D.time_emb_proj
    femb_act: Optional[torch.Tensor]  # [B*T, 1280] fp16: SiLU(femb) as consumed by frame_emb_proj
    text: torch.Tensor                # [B*L_text, cross_dim] fp16 rows of encoder_hidden_states
    text_len: int
    vis_idx: Optional[List[int]]      # vision-condition frame positions inside the window (host ints)
    clip: Optional[torch.Tensor]      # [B*L_ip, cross_dim] fp16 rows of vision_clip_emb
    clip_len: int
    ip_scale: float
    skip_temporal: bool
    text_src: Optional[torch.Tensor] = None  # the caller's tensors the rows were made from (cache identity)
    clip_src: Optional[torch.Tensor] = None
    face: Optional[torch.Tensor] = None      # [B*L_face, cross_dim] fp16 rows of ip_adapter_face_emb (IP-Adapter-FaceID)
    face_len: int = 0
    face_scale: float = 0.0
    face_src: Optional[torch.Tensor] = None
    refer_self: Optional[List[torch.Tensor]] = None  # refer_self_attn_emb ("read"): per spatial block [b, c, t, h, w]
    refer_self_write: Optional[list] = None          # refer_self_attn_emb ("write"): the caller's list, filled per spatial block
    # every ResnetBlock2D.time_emb_proj / TransformerTemporalModel.frame_emb_proj of the network applied in ONE GEMM at
    # the top of the forward ([frames, sum of C_out]); a block takes its column slice.  (39 one-tile launches with
    # M = 26 rows, each a serial 20-step K loop, become 2 launches that fill the chip.)
    emb_proj: Optional[Dict[int, torch.Tensor]] = None
    memo: Optional[PrefixMemo] = None   # shared CFG prefix (see PrefixMemo); None = every call computes

    def shared(self, key, fn):
        """``fn()``, computed once for both CFG halves while the forward is still in its half-independent prefix"""
        return fn() if self.memo is None else self.memo.get(key, fn)

    def split(self) -> None:
        """the next operation may depend on the CFG half: the shared prefix ends here"""
        if self.memo is not None:
            self.memo.split()

    def proj_for(self, module) -> Optional[torch.Tensor]:
        return None if self.emb_proj is None else self.emb_proj.get(id(module))


def tensor_key(t: Optional[torch.Tensor]) -> tuple:
    """Identity of a tensor's current contents for caching derived data across denoise steps: storage pointer,
    shape/strides and the in-place version counter (shared by all views of one storage).  A key is only meaningful
    while the tensor it was taken from is alive -- SourceCache keeps a strong reference for exactly that reason
    (otherwise the caching allocator may hand the same address to a different tensor)."""
    if t is None:
        return ()
    return (t.data_ptr(), tuple(t.shape), tuple(t.stride()), t._version, str(t.dtype))


class SourceCache:
    """values derived from source tensors (K/V projections of the prompt, the image prompt, the ReferenceNet features),
    recomputed when a source changes identity or is modified in place.  Holds a few entries so that the two CFG halves
    (different slices of one embedding tensor) do not evict each other when they are processed separately."""

    __slots__ = ("entries",)
    MAX_ENTRIES = 4

    def __init__(self):
        self.entries = {}  # key -> (src, value); the strong reference to src pins the storage, so a pointer is never recycled

    def get(self, src: torch.Tensor, build):
        key = tensor_key(src)
        hit = self.entries.get(key)
        if hit is None:
            if len(self.entries) >= self.MAX_ENTRIES:
                self.entries.pop(next(iter(self.entries)))  # oldest
            hit = (src, build(src))
            self.entries[key] = hit
        return hit[1]
