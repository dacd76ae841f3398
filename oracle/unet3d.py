"""ORACLE -- test infrastructure only.  Never imported by the product path (musev_amd/*); only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may use it, and only as the checker / CPU baseline.

Plain-PyTorch fp32 CPU restatement of the reference's per-step network,
``musev/models/unet_3d_condition.py:773-1280`` (UNet3DConditionModel.forward), written functionally over a flat
state dict whose keys are the reference's state_dict keys.  Layout and op order follow the reference
literally (NCHW, einops rearranges, torch.cat of skips) -- nothing here is shared with musev_amd.

Pinning status: the musev-specific logic restated here is pinned against the reference's own source executed in this
container (tests/golden/make_reference_goldens.py imports /root/reference/musev with stand-ins for the un-vendored
third-party packages and records input/output vectors that tests/test_oracle_golden.py replays).  The un-vendored
diffusers pieces (ResnetBlock2D, Attention, FeedForward/GEGLU, Timesteps, TimestepEmbedding, Down/Upsample2D) follow
upstream diffusers v0.24 semantics as listed in SURVEY.md 8c and cannot be checked against the TMElyralab fork
(its submodule directory is empty): for those pieces parity is UNPINNED.

Functions cite the reference lines they follow.
"""
from __future__ import annotations

import math
from collections import OrderedDict
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F
from einops import rearrange, repeat

Tensor = torch.Tensor

# Numerical-experiment hook (tools/gpu_error_attribution.py): HOOK(kind, tensor) -> tensor is called at every point where an fp16
# implementation of this network rounds a value ("gemm": a projection / convolution output incl. its fused bias / row bias / gate;
# "gn" / "ln": a normalisation output; "stream_outer": the sum of a residual add on the network's identity path (resnet, temporal
# conv, Transformer2DModel / TransformerTemporalModel outer residual, ReferEmbFuseAttention); "stream_read": what the layers of a block read of that stream -- rounding it
# but not "stream_outer" models a two-fp16 (hi + lo) carry on the identity path whose consumers read the hi half; "stream_inner": the residual adds inside
# a BasicTransformerBlock; "attn_q" / "attn_p" / "attn_o": pre-scaled queries, unnormalised probabilities, attention output;
# "conv_in"; "emb").  None (the default) = the oracle proper: no call, the arithmetic below is untouched.
HOOK = None


def _h(kind: str, x: Tensor) -> Tensor:
    return x if HOOK is None else HOOK(kind, x)


# --------------------------------------------------------------------------------------------------------
# configuration: the three shipped flavours, musev/models/unet_loader.py:232-268
# --------------------------------------------------------------------------------------------------------
_BASE = dict(
    in_channels=4, out_channels=4, block_out_channels=(320, 640, 1280, 1280), layers_per_block=2,
    attention_head_dim=8, cross_attention_dim=768, norm_num_groups=32, norm_eps=1e-5,
    down_block_types=("CrossAttnDownBlock3D", "CrossAttnDownBlock3D", "CrossAttnDownBlock3D", "DownBlock3D"),
    up_block_types=("UpBlock3D", "CrossAttnUpBlock3D", "CrossAttnUpBlock3D", "CrossAttnUpBlock3D"),
)


def flavour_config(name: str, **overrides) -> dict:
    cfg = dict(_BASE)
    if name == "musev":
        cfg.update(need_transformer_in=True, use_anivv1_cfg=False, resnet_2d_skip_time_act=False,
                   keep_vision_condtion=False, need_refer_emb=False, ip_adapter_cross_attn=False)
    elif name in ("musev_referencenet", "musev_referencenet_pose"):
        cfg.update(need_transformer_in=False, use_anivv1_cfg=True, resnet_2d_skip_time_act=True,
                   keep_vision_condtion=True, need_refer_emb=True, ip_adapter_cross_attn=True)
    else:
        raise ValueError(f"unsupport model_name={name}, only support musev, musev_referencenet, musev_referencenet_pose")
    cfg["flavour"] = name
    cfg.update(overrides)
    return cfg


# --------------------------------------------------------------------------------------------------------
# parameter inventory (key -> shape), derived from the reference constructors
# --------------------------------------------------------------------------------------------------------
def _lin(d, p, o, i, bias=True):
    d[p + ".weight"] = (o, i)
    if bias:
        d[p + ".bias"] = (o,)


def _norm(d, p, c):
    d[p + ".weight"] = (c,)
    d[p + ".bias"] = (c,)


def _attn(d, p, c, kv_dim, ip_dim=None, face_dim=None):
    # diffusers Attention: bias-free q/k/v, biased out; IPAttention adds to_k_ip/to_v_ip (attention_processor.py:117-119) and, for
    # IP-Adapter-FaceID, ip_adapter_face_to_k_ip / _to_v_ip (:127-135)
    _lin(d, p + ".to_q", c, c, bias=False)
    _lin(d, p + ".to_k", c, kv_dim, bias=False)
    _lin(d, p + ".to_v", c, kv_dim, bias=False)
    _lin(d, p + ".to_out.0", c, c)
    if ip_dim is not None:
        _lin(d, p + ".to_k_ip", c, ip_dim, bias=False)
        _lin(d, p + ".to_v_ip", c, ip_dim, bias=False)
    if face_dim is not None:
        _lin(d, p + ".ip_adapter_face_to_k_ip", c, face_dim, bias=False)
        _lin(d, p + ".ip_adapter_face_to_v_ip", c, face_dim, bias=False)


def _basic_block(d, p, c, cross_dim, ip, face=False):
    # musev/models/attention.py:52-153 (+ diffusers BasicTransformerBlock norms / FeedForward(geglu))
    _norm(d, p + ".norm1", c)
    _attn(d, p + ".attn1", c, c)
    _norm(d, p + ".norm2", c)
    _attn(d, p + ".attn2", c, cross_dim if cross_dim is not None else c, ip_dim=cross_dim if ip else None,
          face_dim=cross_dim if face else None)
    _norm(d, p + ".norm3", c)
    _lin(d, p + ".ff.net.0.proj", 8 * c, c)
    _lin(d, p + ".ff.net.2", c, 4 * c)


def _resnet(d, p, cin, cout, temb=1280):
    _norm(d, p + ".norm1", cin)
    d[p + ".conv1.weight"] = (cout, cin, 3, 3)
    d[p + ".conv1.bias"] = (cout,)
    _lin(d, p + ".time_emb_proj", cout, temb)
    _norm(d, p + ".norm2", cout)
    d[p + ".conv2.weight"] = (cout, cout, 3, 3)
    d[p + ".conv2.bias"] = (cout,)
    if cin != cout:
        d[p + ".conv_shortcut.weight"] = (cout, cin, 1, 1)
        d[p + ".conv_shortcut.bias"] = (cout,)


def _temp_conv(d, p, c):
    # musev/models/resnet.py:56-89: conv1 = [GN, SiLU, Conv3d] (idx 0, 2); conv2..4 = [GN, SiLU, Dropout, Conv3d] (0, 3)
    for i, ci in ((1, 2), (2, 3), (3, 3), (4, 3)):
        _norm(d, f"{p}.conv{i}.0", c)
        d[f"{p}.conv{i}.{ci}.weight"] = (c, c, 3, 1, 1)
        d[f"{p}.conv{i}.{ci}.bias"] = (c,)
    d[p + ".temporal_weight"] = (1,)


def _transformer2d(d, p, c, cross_dim, ip, face=False):
    _norm(d, p + ".norm", c)
    d[p + ".proj_in.weight"] = (c, c, 1, 1)
    d[p + ".proj_in.bias"] = (c,)
    _basic_block(d, p + ".transformer_blocks.0", c, cross_dim, ip, face)
    d[p + ".proj_out.weight"] = (c, c, 1, 1)
    d[p + ".proj_out.bias"] = (c,)


def _temporal_transformer(d, p, c, femb=1280):
    # musev/models/temporal_transformer.py:117-177
    _norm(d, p + ".norm", c)
    _lin(d, p + ".proj_in", c, c)
    _lin(d, p + ".frame_emb_proj", c, femb)
    _basic_block(d, p + ".transformer_blocks.0", c, None, False)
    _lin(d, p + ".proj_out", c, c)
    d[p + ".temporal_weight"] = (1,)


def _refer_attn(d, p, c):
    _attn(d, p, c, c)


def param_shapes(cfg: dict) -> "OrderedDict[str, Tuple[int, ...]]":
    d: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    ch = cfg["block_out_channels"]
    L = cfg["layers_per_block"]
    xd = cfg["cross_attention_dim"]
    ip = cfg["ip_adapter_cross_attn"]
    face = cfg.get("need_t2i_ip_adapter_face", False)
    ref = cfg["need_refer_emb"]
    temb = ch[0] * 4
    d["conv_in.weight"] = (ch[0], cfg["in_channels"], 3, 3)
    d["conv_in.bias"] = (ch[0],)
    for e in ("time_embedding", "frame_embedding"):
        _lin(d, e + ".linear_1", temb, ch[0])
        _lin(d, e + ".linear_2", temb, temb)
    if cfg["need_transformer_in"]:
        _temporal_transformer(d, "transformer_in", ch[0], temb)
    if ref:
        _refer_attn(d, "first_refer_emb_attns", ch[0])
        _refer_attn(d, "mid_block_refer_emb_attns", ch[-1])
    cin = ch[0]
    for i, bt in enumerate(cfg["down_block_types"]):
        cout = ch[i]
        final = i == len(ch) - 1
        p = f"down_blocks.{i}"
        for j in range(L):
            _resnet(d, f"{p}.resnets.{j}", cin if j == 0 else cout, cout, temb)
            _temp_conv(d, f"{p}.temp_convs.{j}", cout)
            if bt == "CrossAttnDownBlock3D":
                _transformer2d(d, f"{p}.attentions.{j}", cout, xd, ip, face)
                _temporal_transformer(d, f"{p}.temp_attentions.{j}", cout, temb)
        if not final:
            d[f"{p}.downsamplers.0.conv.weight"] = (cout, cout, 3, 3)
            d[f"{p}.downsamplers.0.conv.bias"] = (cout,)
        if ref:
            for k in range(L + (0 if final else 1)):
                _refer_attn(d, f"{p}.refer_emb_attns.{k}", cout)
        cin = cout
    c = ch[-1]
    _resnet(d, "mid_block.resnets.0", c, c, temb)
    _temp_conv(d, "mid_block.temp_convs.0", c)
    _transformer2d(d, "mid_block.attentions.0", c, xd, ip, face)
    _temporal_transformer(d, "mid_block.temp_attentions.0", c, temb)
    _resnet(d, "mid_block.resnets.1", c, c, temb)
    _temp_conv(d, "mid_block.temp_convs.1", c)
    rev = list(reversed(ch))
    out_c = rev[0]
    for i, bt in enumerate(cfg["up_block_types"]):
        prev = out_c
        out_c = rev[i]
        in_c = rev[min(i + 1, len(ch) - 1)]
        p = f"up_blocks.{i}"
        for j in range(L + 1):
            skip = in_c if j == L else out_c
            rin = prev if j == 0 else out_c
            _resnet(d, f"{p}.resnets.{j}", rin + skip, out_c, temb)
            _temp_conv(d, f"{p}.temp_convs.{j}", out_c)
            if bt == "CrossAttnUpBlock3D":
                _transformer2d(d, f"{p}.attentions.{j}", out_c, xd, ip, face)
                _temporal_transformer(d, f"{p}.temp_attentions.{j}", out_c, temb)
        if i != len(ch) - 1:
            d[f"{p}.upsamplers.0.conv.weight"] = (out_c, out_c, 3, 3)
            d[f"{p}.upsamplers.0.conv.bias"] = (out_c,)
    _norm(d, "conv_norm_out", ch[0])
    d["conv_out.weight"] = (cfg["out_channels"], ch[0], 3, 3)
    d["conv_out.bias"] = (cfg["out_channels"],)
    return d


_RESIDUAL_OUT = ("conv2.weight", "to_out.0.weight", "ff.net.2.weight", "proj_out.weight", "conv4.3.weight")


def init_state_dict(cfg: dict, seed: int = 3, gain: float = 1.0, residual_gain: float = 0.3) -> "OrderedDict[str, Tensor]":
    """Seeded random weights (fp32).  The reference zero-initialises conv4 of every TemporalConvLayer, proj_out of
    every TransformerTemporalModel, to_out of every ReferEmbFuseAttention and sets temporal_weight = 1e-5
    (resnet.py:83-92, temporal_transformer.py:171-187, attention_processor.py:626-627); those are re-randomised here
    (SURVEY.md 8c "oracle hygiene") so that parity tests exercise the temporal / refer branches.

    Weights are N(0, 1/fan_in) except the last projection of every residual branch, which is scaled by
    ``residual_gain``: with unit gain the residual stream of this ~100-layer random network grows to |x| ~ 50 and the
    un-normalised ReferEmbFuseAttention logits saturate the softmax, which makes the *function itself* chaotic
    (fp16-level input perturbations change the output by O(1)) -- a regime no trained checkpoint is in and in which
    no implementation can be compared with another."""
    g = torch.Generator().manual_seed(seed)
    sd: "OrderedDict[str, Tensor]" = OrderedDict()
    for k, shp in param_shapes(cfg).items():
        if k.endswith("temporal_weight"):
            v = 0.1 + 0.9 * torch.rand(shp, generator=g)
            if torch.rand((), generator=g) < 0.5:
                v = -v  # the reference uses |temporal_weight|
        elif k.endswith(".weight") and len(shp) == 1:
            v = 1.0 + 0.1 * torch.randn(shp, generator=g)
        elif k.endswith(".bias"):
            v = 0.05 * torch.randn(shp, generator=g)
        else:
            fan_in = 1
            for s in shp[1:]:
                fan_in *= s
            gn = gain * (residual_gain if k.endswith(_RESIDUAL_OUT) else 1.0)
            v = torch.randn(shp, generator=g) * (gn / math.sqrt(fan_in))
        sd[k] = v
    return sd


def calibrate_as_denoiser(sd: "OrderedDict[str, Tensor]", cfg: dict, noise_gain: float = 0.9, random_gain: float = 0.18,
                          carrier: float = 4.0, carrier_mix_seed: Optional[int] = None, carrier_route: str = "skip0") -> "OrderedDict[str, Tensor]":
    """Turns a seeded random state dict (init_state_dict) into one that BEHAVES LIKE A NOISE PREDICTOR, in place, for the loop
    parity tests: eps = noise_gain * (group-normalised input latent) + random_gain * (the random network's prediction).

    Why: a random UNet's eps is uncorrelated with its input, so DDIM's x0 = (x - sqrt(1 - a_t) eps) / sqrt(a_t) (a_951 = 0.006)
    blows the latents up to |x| = 20-55 within a few steps, where the north-star bound |delta latent|max < 1e-2 is below half
    an fp16 ulp of the values the UNet is fed; a trained SD-1.5 checkpoint predicts eps ~ the noise that dominates x_t, which
    keeps the latents O(4).  The same holds here by construction, with weights only (same architecture, every layer still runs
    and contributes through ``random_gain``), using the one linear path the architecture has from input to output:
      conv_in centre tap        x_c -> feature 2c = +carrier * x_c, feature 2c+1 = -carrier * x_c      (unet_3d_condition.py:1008)
      skip 0 -> last up resnet  conv_shortcut of up_blocks[-1].resnets[-1] copies skip features 0..7  (unet_3d_blocks.py:1130)
      residual branches         (temporal conv / transformers) only ADD to the stream
      conv_norm_out + SiLU      unit gain, zero shift on features 0..7; SiLU(z) - SiLU(-z) = z exactly   (:1258-1262)
      conv_out centre tap       eps_c = noise_gain * (feature 2c - feature 2c+1) + random_gain * (random conv_out)
    so the structured part of eps is the latent normalised by its group's statistics (unit variance, like real noise).

    ``random_gain`` sets how far the prediction strays from the noise it is fed: with 0.18 the random network contributes a
    deviation of standard deviation ~0.2 (mean squared error ~0.04 against the input noise -- the order of a trained
    epsilon-predictor's loss at t >= 500, and of the text-dependent difference classifier-free guidance then amplifies); with
    1.0 the function is the plain random network again.  ``noise_gain`` 0.9 x the group statistics' 1.118 makes eps ~ unit
    variance.

    Fixture-sensitivity variants (round 5, tools/cpu_fixture_sweep.py -- is the loop-parity margin a property of THIS carrier?):
    ``carrier_mix_seed``: the latent channels enter the carrier features through a random orthogonal matrix R (feature 2k / 2k+1 =
    +- carrier * (R x)_k) and conv_out un-mixes with R^T -- the same function, but every carrier feature is a mixture of all latent
    channels; ``carrier_route="skip1"``: the carrier is NOT taken from conv_in's output (skip 0) but from the stream one whole level-0
    stage later (skip 1: behind the first ResnetBlock2D / temporal convolution / transformers, whose residual branches have added to
    it), copied by the conv_shortcut of the last up block's second-to-last resnet and handed through its last resnet's conv_shortcut
    from the hidden stream."""
    ch = cfg["block_out_channels"]
    nin = cfg["in_channels"]
    last = len(cfg["up_block_types"]) - 1
    L = cfg["layers_per_block"]
    sc = f"up_blocks.{last}.resnets.{L}.conv_shortcut"
    n = 2 * nin
    if n > ch[0] // cfg["norm_num_groups"]:
        raise ValueError("calibrate_as_denoiser: the carrier features must fit one normalisation group of the first level")
    if carrier_route not in ("skip0", "skip1"):
        raise ValueError("carrier_route")
    R = torch.eye(nin)
    if carrier_mix_seed is not None:
        R = torch.linalg.qr(torch.randn(nin, nin, generator=torch.Generator().manual_seed(carrier_mix_seed)))[0]
    w = sd["conv_in.weight"]
    w[:n] = 0
    sd["conv_in.bias"][:n] = 0
    for k in range(nin):
        for c in range(nin):
            if R[k, c] != 0:   # (the identity writes exactly the entries the unmixed construction always wrote)
                w[2 * k, c, 1, 1] = carrier * R[k, c]
                w[2 * k + 1, c, 1, 1] = -carrier * R[k, c]
    ws = sd[sc + ".weight"]  # [C, C_hidden + C_skip, 1, 1]: torch.cat([hidden, skip]) -> the skip's features come second
    ws[:n] = 0
    sd[sc + ".bias"][:n] = 0
    hidden = ws.shape[1] - ch[0]
    if carrier_route == "skip0":
        for k in range(n):
            ws[k, hidden + k, 0, 0] = 1.0
    else:
        # skip 1 (the stream behind the first level-0 stage) enters through the second-to-last resnet of the last up block; the last
        # resnet hands features 0 .. n-1 of its HIDDEN input on (its own skip, skip 0, no longer feeds them)
        for k in range(n):
            ws[k, k, 0, 0] = 1.0
        sc1 = f"up_blocks.{last}.resnets.{L - 1}.conv_shortcut"
        w1 = sd[sc1 + ".weight"]
        w1[:n] = 0
        sd[sc1 + ".bias"][:n] = 0
        hidden1 = w1.shape[1] - ch[0]
        for k in range(n):
            w1[k, hidden1 + k, 0, 0] = 1.0
    sd["conv_norm_out.weight"][:n] = 1.0
    sd["conv_norm_out.bias"][:n] = 0.0
    wo = sd["conv_out.weight"]
    wo *= random_gain
    sd["conv_out.bias"] *= random_gain
    wo[:, :n] = 0
    for c in range(nin):
        for k in range(nin):
            if R[k, c] != 0:
                wo[c, 2 * k, 1, 1] = noise_gain * R[k, c]
                wo[c, 2 * k + 1, 1, 1] = -noise_gain * R[k, c]
    return sd


# --------------------------------------------------------------------------------------------------------
# un-vendored diffusers pieces (semantics per SURVEY.md 8c; unverifiable against the fork)
# --------------------------------------------------------------------------------------------------------
def timesteps_sincos(t: Tensor, dim: int) -> Tensor:
    """diffusers Timesteps(dim, flip_sin_to_cos=True, downscale_freq_shift=0)."""
    half = dim // 2
    exponent = -math.log(10000.0) * torch.arange(half, dtype=torch.float32, device=t.device) / half
    emb = t.float()[:, None] * torch.exp(exponent)[None, :]
    return torch.cat([torch.cos(emb), torch.sin(emb)], dim=-1)


def timestep_embedding_mlp(sd, p: str, x: Tensor) -> Tensor:
    x = F.linear(x, sd[p + ".linear_1.weight"], sd[p + ".linear_1.bias"])
    x = _h("emb", F.silu(x))
    return _h("emb", F.linear(x, sd[p + ".linear_2.weight"], sd[p + ".linear_2.bias"]))


def resnet_block_2d(sd, p: str, x: Tensor, temb: Tensor, cfg) -> Tensor:
    """diffusers ResnetBlock2D (time_embedding_norm="default", pre_norm, groups 32, eps norm_eps, scale 1)."""
    g, eps = cfg["norm_num_groups"], cfg["norm_eps"]
    xr = _h("stream_read", x)   # what the block's layers read of the stream (the identity path below keeps `x` itself)
    h = F.group_norm(xr, g, sd[p + ".norm1.weight"], sd[p + ".norm1.bias"], eps)
    h = _h("gn", F.silu(h))
    h = F.conv2d(h, sd[p + ".conv1.weight"], sd[p + ".conv1.bias"], padding=1)
    t = temb if cfg["resnet_2d_skip_time_act"] else _h("emb", F.silu(temb))
    t = _h("emb", F.linear(t, sd[p + ".time_emb_proj.weight"], sd[p + ".time_emb_proj.bias"]))[:, :, None, None]
    h = _h("gemm", h + t)
    h = F.group_norm(h, g, sd[p + ".norm2.weight"], sd[p + ".norm2.bias"], eps)
    h = _h("gn", F.silu(h))
    h = _h("gemm", F.conv2d(h, sd[p + ".conv2.weight"], sd[p + ".conv2.bias"], padding=1))
    if (p + ".conv_shortcut.weight") in sd:
        x = _h("gemm", F.conv2d(xr, sd[p + ".conv_shortcut.weight"], sd[p + ".conv_shortcut.bias"]))
    return _h("stream_outer", x + h)


def _heads(x: Tensor, heads: int) -> Tensor:  # [B, L, H*d] -> [B, H, L, d]
    b, l, c = x.shape
    return x.view(b, l, heads, c // heads).transpose(1, 2)


def sdp_attention(q: Tensor, k: Tensor, v: Tensor, heads: int) -> Tensor:
    """softmax(q k^T * d^-0.5) v per head; what xformers.memory_efficient_attention / SDPA compute."""
    qh, kh, vh = _heads(q, heads), _heads(k, heads), _heads(v, heads)
    scale = qh.shape[-1] ** -0.5
    # the same arithmetic per batch item; sliced over the batch so that the BASELINE-size cases (26 frames x 8 heads x
    # 4096 x 8192 scores = 28 GB at once) fit the host memory
    per = max(1, (1 << 28) // max(1, qh.shape[1] * qh.shape[2] * kh.shape[2]))
    outs = []
    for b0 in range(0, qh.shape[0], per):
        if HOOK is None:
            s = torch.matmul(qh[b0:b0 + per], kh[b0:b0 + per].transpose(-1, -2)) * scale
            outs.append(torch.matmul(torch.softmax(s, dim=-1), vh[b0:b0 + per]))
        else:  # the same attention with the rounding points of a flash-style fp16 kernel exposed
            s = torch.matmul(_h("attn_q", qh[b0:b0 + per] * scale), kh[b0:b0 + per].transpose(-1, -2))
            pr = _h("attn_p", torch.exp(s - s.amax(dim=-1, keepdim=True)))
            outs.append(torch.matmul(pr, vh[b0:b0 + per]) / pr.sum(dim=-1, keepdim=True))
    o = outs[0] if len(outs) == 1 else torch.cat(outs, dim=0)
    b, h, l, d = o.shape
    return _h("attn_o", o.transpose(1, 2).reshape(b, l, h * d))


def feed_forward_geglu(sd, p: str, x: Tensor) -> Tensor:
    h = F.linear(x, sd[p + ".net.0.proj.weight"], sd[p + ".net.0.proj.bias"])
    a, gate = h.chunk(2, dim=-1)
    return _h("gemm", F.linear(_h("gemm", a * F.gelu(gate)), sd[p + ".net.2.weight"], sd[p + ".net.2.bias"]))


# --------------------------------------------------------------------------------------------------------
# musev-specific pieces
# --------------------------------------------------------------------------------------------------------
def align_repeat(src: Tensor, target_length: int, dim: int = 0) -> Tensor:
    """musev/data/data_util.py:605-652 for the divisible case used on the hot path (repeat_interleave)."""
    n = src.shape[dim]
    if target_length > n:
        assert target_length % n == 0
        return src.repeat_interleave(target_length // n, dim=dim)
    if target_length < n:
        return src.index_select(dim, torch.arange(target_length, device=src.device))
    return src


def temporal_conv_layer(sd, p: str, x: Tensor, num_frames: int) -> Tensor:
    """musev/models/resnet.py:95-135."""
    h = rearrange(x, "(b t) c h w -> b c t h w", t=num_frames)
    identity = h
    h = _h("stream_read", h)
    for i, ci in ((1, 2), (2, 3), (3, 3), (4, 3)):
        h = F.group_norm(h, 32, sd[f"{p}.conv{i}.0.weight"], sd[f"{p}.conv{i}.0.bias"], 1e-5)
        h = _h("gn", F.silu(h))
        h = F.conv3d(h, sd[f"{p}.conv{i}.{ci}.weight"], sd[f"{p}.conv{i}.{ci}.bias"], padding=(1, 0, 0))
        if i != 4:
            h = _h("gemm", h)
    h = _h("stream_outer", identity + _h("gemm", torch.abs(sd[p + ".temporal_weight"]) * h))
    return rearrange(h, "b c t h w -> (b t) c h w")


def attn_self_reference_only(sd, p: str, x: Tensor, heads: int, num_frames: int, vis_idx: Optional[Tensor],
                             refer_emb: Optional[Tensor] = None) -> Tensor:
    """NonParamT2ISelfReferenceXFormersAttnProcessor, musev/models/attention_processor.py:378-546
    (refer_emb -- refer_self_attn_emb[block] in "read" mode, attention.py:261-289 -- is None for all shipped flavours:
    referencenet_loader.py:111-119)."""
    ehs = x
    if (vis_idx is not None and num_frames > 1) or refer_emb is not None:     # :431-433
        e = rearrange(x, "(b t) hw c -> b t hw c", t=num_frames)
        if vis_idx is not None and num_frames > 1:
            ip = e.index_select(1, vis_idx)
            ip = rearrange(ip, "b t hw c -> b 1 (t hw) c")
            ip = align_repeat(ip, num_frames, dim=1)
            e = torch.cat([e, ip], dim=2)
        if refer_emb is not None:                                                # :476-491
            r = rearrange(refer_emb, "b c t h w -> b 1 (t h w) c")
            r = align_repeat(r, num_frames, dim=1)
            e = torch.cat([e, r], dim=2)
        ehs = rearrange(e, "b t hw c -> (b t) hw c")
    q = _h("gemm", F.linear(x, sd[p + ".to_q.weight"]))
    k = _h("gemm", F.linear(ehs, sd[p + ".to_k.weight"]))
    v = _h("gemm", F.linear(ehs, sd[p + ".to_v.weight"]))
    o = sdp_attention(q, k, v, heads)
    return _h("gemm", F.linear(o, sd[p + ".to_out.0.weight"], sd[p + ".to_out.0.bias"]))


def attn_cross(sd, p: str, x: Tensor, ehs: Tensor, heads: int, vision_clip_emb: Optional[Tensor], ip_scale: float,
               use_ip: bool, face_emb: Optional[Tensor] = None, face_scale: float = 0.0) -> Tensor:
    """text cross-attention; with use_ip the T2IReferencenetIPAdapterXFormersAttnProcessor branch
    (attention_processor.py:176-359), else diffusers' default processor (same math, no IP term)."""
    q = _h("gemm", F.linear(x, sd[p + ".to_q.weight"]))
    e = align_repeat(ehs, x.shape[0], dim=0)
    o = sdp_attention(q, _h("gemm", F.linear(e, sd[p + ".to_k.weight"])), _h("gemm", F.linear(e, sd[p + ".to_v.weight"])), heads)
    if use_ip and ip_scale > 0 and vision_clip_emb is not None:
        batch = ehs.shape[0]  # attention_processor.py:212-216: batch_size is taken from encoder_hidden_states
        ik = align_repeat(_h("gemm", F.linear(vision_clip_emb, sd[p + ".to_k_ip.weight"])), batch, dim=0)
        iv = align_repeat(_h("gemm", F.linear(vision_clip_emb, sd[p + ".to_v_ip.weight"])), batch, dim=0)
        o = o + ip_scale * sdp_attention(q, ik, iv, heads)
    if face_emb is not None and face_scale > 0:   # IP-Adapter-FaceID, attention_processor.py:308-338
        batch = ehs.shape[0]
        fk = align_repeat(_h("gemm", F.linear(face_emb, sd[p + ".ip_adapter_face_to_k_ip.weight"])), batch, dim=0)
        fv = align_repeat(_h("gemm", F.linear(face_emb, sd[p + ".ip_adapter_face_to_v_ip.weight"])), batch, dim=0)
        o = o + face_scale * sdp_attention(q, fk, fv, heads)
    return _h("gemm", F.linear(o, sd[p + ".to_out.0.weight"], sd[p + ".to_out.0.bias"]))


def attn_plain_self(sd, p: str, x: Tensor, heads: int) -> Tensor:
    q = _h("gemm", F.linear(x, sd[p + ".to_q.weight"]))
    k = _h("gemm", F.linear(x, sd[p + ".to_k.weight"]))
    v = _h("gemm", F.linear(x, sd[p + ".to_v.weight"]))
    return _h("gemm", F.linear(sdp_attention(q, k, v, heads), sd[p + ".to_out.0.weight"], sd[p + ".to_out.0.bias"]))


def _ln(sd, p, x):
    return _h("ln", F.layer_norm(x, (x.shape[-1],), sd[p + ".weight"], sd[p + ".bias"], 1e-5))


def basic_block_spatial(sd, p: str, x: Tensor, ehs: Tensor, heads: int, ctx: dict) -> Tensor:
    """musev/models/attention.py:172-431, spatial instance (attn1 = reference-only self-attn, attn2 = text cross).
    The CFG recompute at :319-334 writes a value that is never read (dead) and is not restated."""
    n = _ln(sd, p + ".norm1", x)
    refer_emb = None
    if ctx.get("refer_self_attn_emb_write") is not None:   # attention.py:240-259 ("write"): norm_hidden_states, [bt, hw, c] here
        ctx["refer_self_attn_emb_write"][ctx["spatial_idx"][p]] = n
    if ctx.get("refer_self_attn_emb") is not None:   # attention.py:261-289: indexed by the block's spatial_self_attn_idx
        refer_emb = ctx["refer_self_attn_emb"][ctx["spatial_idx"][p]]
    x = _h("stream_inner", attn_self_reference_only(sd, p + ".attn1", n, heads, ctx["num_frames"], ctx["vis_idx"], refer_emb) + x)
    n = _ln(sd, p + ".norm2", x)
    x = _h("stream_inner", attn_cross(sd, p + ".attn2", n, ehs, heads, ctx["vision_clip_emb"], ctx["ip_adapter_scale"], ctx["use_ip"],
                                      ctx.get("ip_adapter_face_emb"), ctx.get("ip_adapter_face_scale", 0.0)) + x)
    n = _ln(sd, p + ".norm3", x)
    return _h("stream_inner", feed_forward_geglu(sd, p + ".ff", n) + x)


def basic_block_temporal(sd, p: str, x: Tensor, heads: int) -> Tensor:
    """same class with double_self_attention=True: attn1 and attn2 are self-attention over T (attention.py:80-81,
    345-366; default SDPA processor, temporal_transformer.py:50-52)."""
    x = _h("stream_inner", attn_plain_self(sd, p + ".attn1", _ln(sd, p + ".norm1", x), heads) + x)
    x = _h("stream_inner", attn_plain_self(sd, p + ".attn2", _ln(sd, p + ".norm2", x), heads) + x)
    return _h("stream_inner", feed_forward_geglu(sd, p + ".ff", _ln(sd, p + ".norm3", x)) + x)


def transformer_2d(sd, p: str, x: Tensor, ehs: Tensor, heads: int, ctx: dict) -> Tensor:
    """musev/models/transformer_2d.py:172-445, continuous-input branch; GroupNorm eps 1e-6 (diffusers ctor)."""
    b, c, h, w = x.shape
    res = x
    y = _h("gn", F.group_norm(_h("stream_read", x), 32, sd[p + ".norm.weight"], sd[p + ".norm.bias"], 1e-6))
    y = _h("gemm", F.conv2d(y, sd[p + ".proj_in.weight"], sd[p + ".proj_in.bias"]))
    y = y.permute(0, 2, 3, 1).reshape(b, h * w, c)
    y = basic_block_spatial(sd, p + ".transformer_blocks.0", y, ehs, heads, ctx)
    if ctx.get("refer_self_attn_emb_write") is not None:   # transformer_2d.py:340-359: "bt (h w) c -> bt c h w"
        i = ctx["spatial_idx"][p + ".transformer_blocks.0"]
        ctx["refer_self_attn_emb_write"][i] = ctx["refer_self_attn_emb_write"][i].reshape(b, h, w, c).permute(0, 3, 1, 2).contiguous()
    y = y.reshape(b, h, w, c).permute(0, 3, 1, 2).contiguous()
    y = _h("gemm", F.conv2d(y, sd[p + ".proj_out.weight"], sd[p + ".proj_out.bias"]))
    return _h("stream_outer", y + res)


def transformer_temporal(sd, p: str, x: Tensor, femb: Tensor, heads: int, num_frames: int) -> Tensor:
    """musev/models/temporal_transformer.py:189-308."""
    bt, c, h, w = x.shape
    b = bt // num_frames
    y = rearrange(x, "(b t) c h w -> b c t h w", b=b)
    res = y
    y = _h("gn", F.group_norm(_h("stream_read", y), 32, sd[p + ".norm.weight"], sd[p + ".norm.bias"], 1e-6))
    y = rearrange(y, "b c t h w -> (b h w) t c")
    y = F.linear(y, sd[p + ".proj_in.weight"], sd[p + ".proj_in.bias"])
    f = _h("emb", F.linear(_h("emb", F.silu(femb)), sd[p + ".frame_emb_proj.weight"], sd[p + ".frame_emb_proj.bias"]))  # [b, t, c]
    y = _h("gemm", y + align_repeat(f, y.shape[0], dim=0))
    y = basic_block_temporal(sd, p + ".transformer_blocks.0", y, heads)
    y = F.linear(y, sd[p + ".proj_out.weight"], sd[p + ".proj_out.bias"])
    y = rearrange(y, "(b h w) t c -> b c t h w", b=b, h=h, w=w)
    out = _h("stream_outer", res + _h("gemm", torch.abs(sd[p + ".temporal_weight"]) * y))
    return rearrange(out, "b c t h w -> (b t) c h w")


def refer_emb_fuse_attention(sd, p: str, x: Tensor, ref: Tensor, heads: int, num_frames: int) -> Tensor:
    """ReferEmbFuseAttention.forward, musev/models/attention_processor.py:629-750: K/V = [ref tokens, self tokens]."""
    residual = x
    y = rearrange(x, "(b t) c h w -> b c t h w", t=num_frames)
    b, c, t1, h, w = y.shape
    e = rearrange(ref, "b c t2 h w -> b (t2 h w) c")
    e = repeat(e, "b n c -> (b t) n c", t=t1)
    y = rearrange(y, "b c t h w -> (b t) (h w) c")
    e = torch.cat([e, y], dim=1)
    q = _h("gemm", F.linear(y, sd[p + ".to_q.weight"]))
    o = sdp_attention(q, _h("gemm", F.linear(e, sd[p + ".to_k.weight"])), _h("gemm", F.linear(e, sd[p + ".to_v.weight"])), heads)
    o = _h("gemm", F.linear(o, sd[p + ".to_out.0.weight"], sd[p + ".to_out.0.bias"]))
    o = rearrange(o, "bt (h w) c -> bt c h w", h=h, w=w)
    return _h("stream_outer", o + residual)


# --------------------------------------------------------------------------------------------------------
# the network
# --------------------------------------------------------------------------------------------------------
def unet3d_forward(
    sd: Dict[str, Tensor], cfg: dict, sample: Tensor, timestep, encoder_hidden_states: Tensor, *,
    sample_index: Optional[Tensor] = None, vision_conditon_frames_sample_index: Optional[Tensor] = None,
    sample_frame_rate: int = 10, down_block_refer_embs: Optional[Sequence[Tensor]] = None,
    mid_block_refer_emb: Optional[Tensor] = None, vision_clip_emb: Optional[Tensor] = None,
    ip_adapter_scale: float = 1.0, down_block_additional_residuals: Optional[Sequence[Tensor]] = None,
    mid_block_additional_residual: Optional[Tensor] = None, pose_guider_emb: Optional[Tensor] = None,
    skip_temporal_layers: bool = False, collect: Optional[dict] = None, ip_adapter_face_emb: Optional[Tensor] = None,
    ip_adapter_face_scale: float = 1.0, refer_self_attn_emb: Optional[Sequence[Tensor]] = None,
    refer_self_attn_emb_mode: str = "read",
) -> Tensor:
    """UNet3DConditionModel.forward (unet_3d_condition.py:773-1280).  sample [b, c, t, h, w] -> same shape.
    `collect` (optional dict) receives named intermediate activations for block-level parity tests."""
    if refer_self_attn_emb is not None and refer_self_attn_emb_mode.lower() not in ("read", "write"):
        raise ValueError(f"refer_self_attn_emb_mode {refer_self_attn_emb_mode!r}")
    write_embs = refer_self_attn_emb is not None and refer_self_attn_emb_mode.lower() == "write"
    ch = cfg["block_out_channels"]
    heads = cfg["attention_head_dim"]
    L = cfg["layers_per_block"]
    b, _, num_frames, height, width = sample.shape
    vis_idx = vision_conditon_frames_sample_index

    # 1. time embedding (:887-906)
    t = timestep if torch.is_tensor(timestep) else torch.tensor([timestep])
    t = t.reshape(-1).expand(b).to(sample.device)
    wdt = sd["conv_in.weight"].dtype  # the weights' dtype: fp32 (the oracle) or fp16 (the "fp16 torch" drift floor of the tests)
    emb = timestep_embedding_mlp(sd, "time_embedding", timesteps_sincos(t, ch[0]).to(wdt))
    if cfg["use_anivv1_cfg"]:
        emb = F.silu(emb)
    emb = emb.repeat_interleave(num_frames, dim=0)
    if cfg["keep_vision_condtion"] and num_frames > 1 and sample_index is not None and vis_idx is not None:
        emb = rearrange(emb, "(b t) d -> b t d", t=num_frames).clone()
        emb[:, vis_idx, :] = 0
        emb = rearrange(emb, "b t d -> (b t) d")
    # frame embedding (:909-937); frame_index is window-local
    frame_index = torch.arange(num_frames, dtype=torch.long, device=sample.device)
    if cfg["use_anivv1_cfg"]:
        frame_index = (frame_index * sample_frame_rate).to(dtype=torch.long)
    femb = repeat(timesteps_sincos(frame_index, ch[0]).to(wdt), "t d -> b t d", b=b)
    femb = timestep_embedding_mlp(sd, "frame_embedding", femb)
    if cfg["use_anivv1_cfg"]:
        femb = F.silu(femb)
    ehs = align_repeat(encoder_hidden_states, emb.shape[0], dim=0)  # :938-941

    ctx = dict(num_frames=num_frames, vis_idx=vis_idx, vision_clip_emb=vision_clip_emb,
               ip_adapter_scale=ip_adapter_scale, use_ip=cfg["ip_adapter_cross_attn"],
               # unet_3d_condition.py:1000-1006: only models built with need_t2i_ip_adapter_face hand the face tokens on
               ip_adapter_face_emb=ip_adapter_face_emb if cfg.get("need_t2i_ip_adapter_face", False) else None,
               ip_adapter_face_scale=ip_adapter_face_scale,
               # refer_self_attn_emb ("read"): the spatial blocks are numbered in the sorted order of their module names
               # (insert_spatial_self_attn_idx, unet_3d_condition.py:1663-1686)
               refer_self_attn_emb=None if write_embs else refer_self_attn_emb,
               # "write" (attention.py:240-259, transformer_2d.py:340-359): every spatial block leaves the INPUT of its self-attention
               # (norm1's output) in the caller's list, as [(b t), c, h, w]
               refer_self_attn_emb_write=refer_self_attn_emb if write_embs else None,
               # -- every BasicTransformerBlock outside "temp_attentions", i.e. incl. transformer_in's (get_attns' exclude test
               # overwrites its include test, :1720-1726)
               spatial_idx={k: i for i, k in enumerate(sorted({key[:-len(".norm1.weight")] for key in sd
                                                               if "temp_attentions" not in key and key.endswith(".transformer_blocks.0.norm1.weight")}))})

    def rec(name, x):
        if collect is not None:
            collect[name] = x.detach().clone()

    def tconv(p, x):
        return x if skip_temporal_layers else temporal_conv_layer(sd, p, x, num_frames)

    def tattn(p, x):
        return x if skip_temporal_layers else transformer_temporal(sd, p, x, femb, heads, num_frames)

    skip_refer = False  # skip_refer_downblock_emb is initialised False (:605-607) and never set (:1646-1655 commented out)
    use_refer = cfg["need_refer_emb"] and down_block_refer_embs is not None and not skip_refer

    # 2. pre-process (:1008-1063)
    x = rearrange(sample, "b c t h w -> (b t) c h w")
    x = F.conv2d(x, sd["conv_in.weight"], sd["conv_in.bias"], padding=1)
    if pose_guider_emb is not None:
        x = x + pose_guider_emb
    x = _h("conv_in", x)
    rec("conv_in", x)
    if cfg["need_transformer_in"]:
        x = tattn("transformer_in", x)
        rec("transformer_in", x)
    if use_refer:
        x = refer_emb_fuse_attention(sd, "first_refer_emb_attns", x, down_block_refer_embs[0], heads, num_frames)

    # 3. down (:1076-1156)
    skips: List[Tensor] = [x]
    for i, bt in enumerate(cfg["down_block_types"]):
        p = f"down_blocks.{i}"
        final = i == len(ch) - 1
        refer = None
        if use_refer:
            nb = L + int(not final * 1)  # quirk (:1090): `not is_final_block * 1`
            start = 1 + nb * i
            refer = down_block_refer_embs[start:start + nb]
        for j in range(L):
            x = resnet_block_2d(sd, f"{p}.resnets.{j}", x, emb, cfg)
            x = tconv(f"{p}.temp_convs.{j}", x)
            if bt == "CrossAttnDownBlock3D":
                x = transformer_2d(sd, f"{p}.attentions.{j}", x, ehs, heads, ctx)
                x = tattn(f"{p}.temp_attentions.{j}", x)
            # batch_adain_conditioned_tensor is a no-op for 4-D input (data_util.py:600-601)
            if refer is not None:
                x = refer_emb_fuse_attention(sd, f"{p}.refer_emb_attns.{j}", x, refer[j], heads, num_frames)
            skips.append(x)
            rec(f"{p}.out{j}", x)
        if not final:
            x = _h("gemm", F.conv2d(_h("stream_read", x), sd[f"{p}.downsamplers.0.conv.weight"], sd[f"{p}.downsamplers.0.conv.bias"], stride=2, padding=1))
            if refer is not None:
                x = refer_emb_fuse_attention(sd, f"{p}.refer_emb_attns.{L}", x, refer[L], heads, num_frames)
            skips.append(x)
            rec(f"{p}.out{L}", x)
    if down_block_additional_residuals is not None:
        skips = [s + r for s, r in zip(skips, down_block_additional_residuals)]

    # 4. mid (:1159-1195; unet_3d_blocks.py:379-431)
    x = resnet_block_2d(sd, "mid_block.resnets.0", x, emb, cfg)
    x = tconv("mid_block.temp_convs.0", x)
    x = transformer_2d(sd, "mid_block.attentions.0", x, ehs, heads, ctx)
    x = tattn("mid_block.temp_attentions.0", x)
    x = resnet_block_2d(sd, "mid_block.resnets.1", x, emb, cfg)
    x = tconv("mid_block.temp_convs.1", x)
    if cfg["need_refer_emb"] and mid_block_refer_emb is not None and not skip_refer:
        x = refer_emb_fuse_attention(sd, "mid_block_refer_emb_attns", x, mid_block_refer_emb, heads, num_frames)
    if mid_block_additional_residual is not None:
        x = x + mid_block_additional_residual
    rec("mid", x)

    # 5. up (:1199-1245)
    for i, bt in enumerate(cfg["up_block_types"]):
        p = f"up_blocks.{i}"
        for j in range(L + 1):
            x = torch.cat([x, skips.pop()], dim=1)
            x = resnet_block_2d(sd, f"{p}.resnets.{j}", x, emb, cfg)
            x = tconv(f"{p}.temp_convs.{j}", x)
            if bt == "CrossAttnUpBlock3D":
                x = transformer_2d(sd, f"{p}.attentions.{j}", x, ehs, heads, ctx)
                x = tattn(f"{p}.temp_attentions.{j}", x)
        if i != len(ch) - 1:
            # forward_upsample_size (:841-849): a latent size that is not a multiple of 2^(number of upsamplers) hands the upsampler the size
            # of the skip on top of the remaining stack (upsample_size = down_block_res_samples[-1].shape[2:], :1209-1210; diffusers
            # Upsample2D then interpolates to size=output_size instead of x 2 -- un-vendored, restated from its published forward)
            if any(v % (2 ** (len(ch) - 1)) != 0 for v in sample.shape[-2:]):
                x = F.interpolate(_h("stream_read", x), size=tuple(skips[-1].shape[-2:]), mode="nearest")
            else:
                x = F.interpolate(_h("stream_read", x), scale_factor=2.0, mode="nearest")
            x = _h("gemm", F.conv2d(x, sd[f"{p}.upsamplers.0.conv.weight"], sd[f"{p}.upsamplers.0.conv.bias"], padding=1))
        rec(p, x)

    # 6. post-process (:1258-1263)
    x = F.group_norm(_h("stream_read", x), cfg["norm_num_groups"], sd["conv_norm_out.weight"], sd["conv_norm_out.bias"], cfg["norm_eps"])
    x = _h("gn_out", F.silu(x))
    x = F.conv2d(x, sd["conv_out.weight"], sd["conv_out.bias"], padding=1)
    return rearrange(x, "(b t) c h w -> b c t h w", t=num_frames)
