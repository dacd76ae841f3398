"""Tensor-level wrappers over the C ABI (libmusev_hip.so).

PyTorch is used here only as plumbing: device memory (caching allocator), the current HIP stream and dtype
bookkeeping.  Every function launches hand-written gfx950 kernels on ``torch.cuda.current_stream()``; there is no
eager/CPU fallback -- a missing library or a non-CUDA tensor raises.

Activations are "rows x channels" fp16 matrices in channels-last order ([B, T, H, W, C] flattened to [B*T*H*W, C]);
2-D views with a unit inner stride and an arbitrary row stride are accepted everywhere (column slices of a fused
QKV projection are passed without copies).
"""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Optional, Sequence, Tuple

import torch

from . import _lib
from ._lib import (AttnDesc, GemmDesc, MV_ACT_NONE, MV_ACT_SILU, MV_GEMM_CONV3X3, MV_GEMM_LINEAR, MV_GEMM_TCONV3,
                   check)

__all__ = [
    "gemm", "ln_fold_applies", "fold_layernorm", "conv3x3", "tconv3", "groupnorm", "groupnorm_fold_linear", "xab_fused_applies", "pack_xab_q", "xattn_block", "layernorm", "attention", "temporal_attention", "geglu", "silu", "add", "softmax_rows_",
    "conv3x3_cin_small", "conv3x3_cin_small_gemm", "pad_cols", "conv3x3_cout_small", "conv3x3_direct", "timestep_embedding", "zero_rows", "bcthw_to_bthwc", "bthwc_to_bcthw",
    "window_gather", "window_scatter_add", "window_units_reduce", "cfg_ddim_step", "cfg_affine_step", "pack_conv_weight", "probe_tr16", "MV_ACT_NONE", "MV_ACT_SILU",
]


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


# ---- optional launch recorder (bench.py's roofline pass) --------------------------------------------------------------
# When a list is installed here, every implicit-GEMM launch appends (descriptor copy, tensors kept alive, algorithmic HBM bytes)
# in launch order IN ADDITION to being launched.  bench.py records one step this way and then re-issues the recorded launches
# back to back on one stream between ONE pair of HIP events (`replay_gemms`), so that the measured time is device time of the
# kernels (+ the dispatcher's own back-to-back gaps, which a hipGraph replay has as well) and not host launch latency.
GEMM_RECORD: Optional[list] = None


# Tile configuration / split-K factor forced on every implicit-GEMM launch of this process (tuner and A/B runs; the C library
# itself keeps no such state -- the choice travels in each call's descriptor).  Read once at import:
#   MUSEV_GEMM_CFG    -1 = measured per-shape table, then rules (default); -2 = rules only; >= 0 = that catalogue id
#   MUSEV_GEMM_SPLITK  0 = library's choice (default); >= 1 = that many K slices where the workspace cap allows
#   MUSEV_OPS          "NAME=VALUE,..." sets module switches of this file by name at import (same-box A/B legs of tools/gpu_ab.sh:
#                      COLSTATS, CARRY, CARRY_MAX_C, FFN_FUSED, FFN_ROTATE, TSA_FUSED, LN_FOLD, LN_FOLD_MAX_K, ATTN_GROUPS, XATTN_RESIDENT,
#                      XATTN_RESIDENT_MAX_D, GEMM_WEIGHT_STATIONARY, GN_FOLD, GN_FOLD_MAX_RATIO, XAB_FUSED); applied at the bottom of this file.  The per-feature variables of earlier rounds
#                      (MUSEV_CARRY, MUSEV_SHARE_PREFIX, MUSEV_XATTN_RESIDENT, MUSEV_GEMM_WEIGHT_STATIONARY, ...) are gone: setting one
#                      raises at import, so that an A/B leg written from an old example cannot silently measure the baseline twice.
GEMM_CFG: int = int(os.environ.get("MUSEV_GEMM_CFG", "-1"))
GEMM_SPLITK: int = int(os.environ.get("MUSEV_GEMM_SPLITK", "0"))
# mv_gemm_desc.tile_order = 1: the library MAY take the weight-stationary workgroup order -- an XCD's workgroups cover a few n-tiles x
# all m-tiles, so each XCD streams 1/8 of the weight matrix instead of all of it -- and takes it where its fetch model says the XCDs
# fetch >= 5 % less (the small grids of the 8 x 8-latent level under a K split).  Same results bit for bit.  Round 5, same-box
# (profiles/r05f_ab_config2.log): time-neutral in the two-stream step (52.33 / 52.32 against 52.30 / 52.13 / 52.52); kept on for the
# HBM bytes it saves.
GEMM_WEIGHT_STATIONARY: bool = True


# Producer-side GroupNorm statistics (A/B: MUSEV_OPS="COLSTATS=0" keeps the statistics pass of mv_groupnorm_f16):
# the convolutions / proj_out launches whose output a GroupNorm reads next also emit per-row-tile column sums from their epilogue
# (mv_gemm_desc.colstats); the buffer rides on the output tensor OBJECT (`_mv_colstats`), so a view, a copy or any tensor produced
# some other way simply has none and `groupnorm` takes its own statistics pass.
COLSTATS: bool = True
COLSTATS_HITS: int = 0   # GroupNorm calls served from producer statistics (tests / reports)


# Two-fp16 carry on the identity path of the residual stream (A/B: MUSEV_OPS="CARRY=0" switches it off).  The fp16
# rounding of the stream at every `x + f(x)` is the largest single contribution to the forward's error against the fp32 reference
# (profiles/r04a_attribution.log: 5.2e-3 of 5.6e-3 |delta eps|max alone) -- and it sits at level 0, where the network's largest values
# live.  With the carry the epilogue of a stream-producing launch (conv_in, ResnetBlock2D.conv2 + shortcut, the temporal convolution's
# last conv + identity, Transformer2DModel / TransformerTemporalModel proj_out + residual) stores the fp32 sum as TWO fp16 tensors
# (hi = fp16(s), lo = fp16(s - hi)); every layer reads hi -- an ordinary fp16 tensor -- and only the next residual add picks lo up
# again (the lo tensor rides on the hi tensor OBJECT as `_mv_lo`, like the column statistics: a view / copy / tensor produced some
# other way has none).  Applied where the stream is at most CARRY_MAX_C channels wide (level 0 of SD-1.5: 320; the deeper levels add
# nothing measurable, profiles/r04b_attribution_carry.log).
CARRY: bool = True
CARRY_MAX_C: int = 320
CARRY_HITS: int = 0


def _lo_of(t: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
    """the lo half riding on ``t`` (see CARRY), if it still describes the tensor: same geometry, and no in-place torch op on the hi
    half since it was written (the torch version counter travels with the attribute, like the column statistics')"""
    if t is None:
        return None
    lo = getattr(t, "_mv_lo", None)
    if lo is None or lo.shape != t.shape or lo.stride() != t.stride() or getattr(t, "_mv_lo_version", None) != t._version:
        return None
    return lo


def _set_lo(t: torch.Tensor, lo: torch.Tensor) -> None:
    t._mv_lo = lo
    t._mv_lo_version = t._version


def _carry_setup(d: GemmDesc, o: torch.Tensor, residual: Optional[torch.Tensor], carry: bool, keep: tuple) -> tuple:
    """fills d.c_lo / d.residual_lo for a stream-producing launch; returns the tensors to keep alive"""
    if not (carry and CARRY) or o.shape[1] > CARRY_MAX_C or o.shape[1] % 8 or o.stride(0) % 8 or o.data_ptr() % 16:
        return keep
    if residual is not None and (residual.stride(0) % 8 or residual.data_ptr() % 16):
        return keep
    global CARRY_HITS
    CARRY_HITS += 1
    lo = torch.empty_strided(o.shape, o.stride(), dtype=torch.float16, device=o.device)
    d.c_lo = lo.data_ptr()
    rlo = _lo_of(residual)
    if rlo is not None:
        d.residual_lo = rlo.data_ptr()
        keep = keep + (rlo,)
    _set_lo(o, lo)
    return keep + (lo,)


def _launch_gemm(d: GemmDesc, what: str, dev: torch.device, keep: tuple = (), colstats_for: Optional[torch.Tensor] = None) -> None:
    lib = _lib.load()
    d.cfg, d.splitk = GEMM_CFG, (0 if d.c_lo else GEMM_SPLITK)   # (a carry launch is never split: a forced split would be refused)
    d.tile_order = int(GEMM_WEIGHT_STATIONARY)
    pending_stats = None
    if colstats_for is not None and COLSTATS:
        rpt, nfl = C.c_int32(), C.c_int64()
        check(lib.mv_gemm_stats_layout(C.byref(d), C.byref(rpt), C.byref(nfl)), what)
        if rpt.value > 0:
            cs = torch.empty(nfl.value, dtype=torch.float32, device=dev)
            d.colstats, d.colstats_floats = cs.data_ptr(), nfl.value
            pending_stats = (cs, rpt.value)
            keep = keep + (cs,)
    need = lib.mv_gemm_workspace_bytes(C.byref(d))
    if need < 0:
        check(1, what)
    ws = None
    if need > 0:  # split-K slabs: scratch from torch's caching allocator (inside a graph capture: the capture's private pool)
        ws = torch.empty(need, dtype=torch.uint8, device=dev)
        d.workspace, d.workspace_bytes = ws.data_ptr(), need
    check(lib.mv_gemm_f16(C.byref(d), _stream()), what)
    if pending_stats is not None:
        # (set once the launch was accepted; the tensor's torch version rides along: a later in-place torch op on the tensor -- copy_,
        # add_, ... -- bumps it and groupnorm() then ignores the stale statistics)
        colstats_for._mv_colstats = pending_stats + (colstats_for._version,)
    if GEMM_RECORD is not None:
        # algorithmic HBM bytes of the launch: every operand read once, the output written once
        rows_in = int(d.M)
        if d.mode == MV_GEMM_CONV3X3:
            rows_in = (int(d.M) // (int(d.hout) * int(d.wout))) * int(d.hin) * int(d.win)
        cols = int(d.N) // 2 if d.geglu else int(d.N)
        nbytes = 2 * (rows_in * (int(d.c1) + int(d.c2)) + int(d.N) * int(d.K) +
                      int(d.M) * cols * ((2 if d.residual else 1) + (1 if d.c_lo else 0) + (1 if d.residual_lo else 0)))
        GEMM_RECORD.append((GemmDesc.from_buffer_copy(d), keep + (ws,), nbytes, _stream()))


def _replay_one(lib, d, st) -> None:
    if isinstance(d, _lib.FfnDesc):
        check(lib.mv_ffn_geglu_f16(C.byref(d), st), "mv_ffn_geglu_f16(replay)")
    elif isinstance(d, _lib.TsaDesc):
        check(lib.mv_temporal_attn_block_f16(C.byref(d), st), "mv_temporal_attn_block_f16(replay)")
    elif isinstance(d, _lib.XabDesc):
        check(lib.mv_xattn_block_f16(C.byref(d), st), "mv_xattn_block_f16(replay)")
    else:
        check(lib.mv_gemm_f16(C.byref(d), st), "mv_gemm_f16(replay)")


def record_flops(d) -> float:
    """algorithmic FLOPs of a recorded launch (mv_gemm_f16: 2 M N K; the fused feed-forward: both projections)"""
    if isinstance(d, _lib.FfnDesc):
        return 2.0 * d.M * d.C * 2 * d.H + 2.0 * d.M * d.H * d.C
    if isinstance(d, _lib.TsaDesc):   # the q / k / v projection and to_out (the T x T attention itself is not counted, as before)
        return 2.0 * d.B * d.T * d.HW * d.C * 4 * d.C
    if isinstance(d, _lib.XabDesc):   # the q projection and to_out (the attention over <= 80 keys is not counted, as for the separate launch)
        return 2.0 * d.M * d.C * 2 * d.C
    return 2.0 * d.M * d.N * d.K


def replay_gemms(record: Sequence[tuple], reps: int = 1) -> float:
    """re-issues recorded implicit-GEMM launches (see GEMM_RECORD) ``reps`` times back to back on the current stream between one
    pair of HIP events; returns the elapsed device milliseconds of ALL reps.  The host enqueues a launch in a few microseconds and
    the kernels take tens to hundreds, so the queue never runs dry: the elapsed time is the kernels' own."""
    lib = _lib.load()
    st = _stream()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        for d, _keep, _nb, *_ in record:
            _replay_one(lib, d, st)
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1)


def _on_gpu(t: torch.Tensor) -> bool:
    """every wrapper refuses tensors that are not in device memory through this one predicate (there is no CPU path)"""
    return t.is_cuda


def _p(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _mat(t: torch.Tensor, name: str) -> torch.Tensor:
    if t.dim() != 2 or t.stride(1) != 1:
        raise ValueError(f"{name}: expected a 2-D tensor with unit inner stride, got shape {tuple(t.shape)} strides {t.stride()}")
    if t.dtype != torch.float16 or not _on_gpu(t):
        raise ValueError(f"{name}: expected a CUDA fp16 tensor, got {t.dtype} on {t.device}")
    return t


def _vec(t: Optional[torch.Tensor], name: str, n: Optional[int] = None) -> Optional[torch.Tensor]:
    if t is None:
        return None
    if t.dtype != torch.float16 or not _on_gpu(t) or not t.is_contiguous():
        raise ValueError(f"{name}: expected a contiguous CUDA fp16 tensor")
    if n is not None and t.numel() != n:
        raise ValueError(f"{name}: expected {n} elements, got {t.numel()}")
    return t


def _fill_epilogue(d: GemmDesc, N: int, M: int, bias, rowbias, rows_per_group, residual, alpha, act, out_cols):
    d.bias = _p(_vec(bias, "bias", N))
    if rowbias is not None:
        rb = _mat(rowbias, "rowbias")
        d.rowbias, d.ldrb, d.rows_per_group = rb.data_ptr(), rb.stride(0), int(rows_per_group)
        if rows_per_group <= 0 or rb.shape[0] * rows_per_group < M or rb.shape[1] != N:
            raise ValueError("rowbias: shape / rows_per_group do not cover the output")
    if residual is not None:
        r = _mat(residual, "residual")
        if r.shape[0] != M or r.shape[1] != out_cols:
            raise ValueError("residual: shape mismatch")
        d.residual, d.ldr = r.data_ptr(), r.stride(0)
    if alpha is not None:
        if alpha.dtype != torch.float32 or alpha.numel() != 1 or not _on_gpu(alpha):
            raise ValueError("alpha: expected a CUDA fp32 scalar tensor")
        d.alpha = alpha.data_ptr()
    d.act = int(act)


def _out(out: Optional[torch.Tensor], M: int, cols: int, like: torch.Tensor) -> torch.Tensor:
    if out is None:
        return torch.empty((M, cols), dtype=torch.float16, device=like.device)
    o = _mat(out, "out")
    if o.shape[0] != M or o.shape[1] != cols:
        raise ValueError(f"out: expected {(M, cols)}, got {tuple(o.shape)}")
    if hasattr(o, "_mv_colstats"):  # the tensor is about to be overwritten: statistics of its previous contents do not follow
        del o._mv_colstats
    if hasattr(o, "_mv_lo"):
        del o._mv_lo
    return o


def gemm(a: torch.Tensor, w: torch.Tensor, *, a2: Optional[torch.Tensor] = None, bias=None, rowbias=None,
         rows_per_group: int = 0, residual=None, alpha=None, act: int = MV_ACT_NONE, geglu: bool = False,
         out: Optional[torch.Tensor] = None, ln: Optional[Tuple[torch.Tensor, torch.Tensor, float]] = None,
         colstats: bool = False, carry: bool = False, w_groups: int = 1, rowbias_lo: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out = act(|alpha| * ([a | a2] @ w.T + bias + rowbias[row // rows_per_group])) + residual   (fp16, fp32 accumulate).
    ``colstats``: the output feeds a GroupNorm next -- emit its column statistics from the epilogue (see COLSTATS).
    ``carry``: the output is the residual stream -- keep the fp32 sum as two fp16 tensors (see CARRY).

    ``w`` is [N, K] (torch Linear layout).  With ``geglu`` the rows of ``w`` / ``bias`` must be packed by
    :func:`pack_geglu` and the result has N/2 columns: value * gelu(gate).
    ``ln = (colsum, colbias, eps)`` (see :func:`fold_layernorm`): ``a`` holds RAW rows and ``w`` gamma-scaled weights; the kernel
    forms each row's LayerNorm statistics from the fragments it multiplies and applies them in the epilogue -- the result is
    LayerNorm(a) @ W.T + bias without the normalised tensor ever existing (no bias / rowbias / second source then).
    ``w_groups`` > 1: ``w`` holds that many [N, K] matrices stacked ([w_groups * N, K]); rows [g * M / w_groups, (g + 1) * M / w_groups)
    of ``a`` multiply matrix g (:func:`groupnorm_fold_linear`).  ``rowbias_lo``: second fp16 half of ``rowbias`` (same shape).
    """
    a = _mat(a, "a")
    w = _mat(w, "w")
    if not w.is_contiguous():
        raise ValueError("w must be contiguous [N, K]")
    M, c1 = a.shape
    N, K = w.shape
    d = GemmDesc()
    if w_groups > 1:
        if N % w_groups or M % w_groups:
            raise ValueError("gemm: w_groups must divide the rows of w and of a")
        N //= w_groups
        d.w_group_rows = M // w_groups
    d.a, d.lda, d.c1 = a.data_ptr(), a.stride(0), c1
    if a2 is not None:
        a2 = _mat(a2, "a2")
        if a2.shape[0] != M:
            raise ValueError("a2: row count mismatch")
        d.a2, d.lda2, d.c2 = a2.data_ptr(), a2.stride(0), a2.shape[1]
    if c1 + (a2.shape[1] if a2 is not None else 0) != K:
        raise ValueError(f"gemm: K mismatch: inputs give {c1 + (a2.shape[1] if a2 is not None else 0)}, weight has {K}")
    cols = N // 2 if geglu else N
    o = _out(out, M, cols, a)
    d.w, d.c, d.ldc = w.data_ptr(), o.data_ptr(), o.stride(0)
    d.M, d.N, d.K = M, N, K
    d.mode, d.geglu = MV_GEMM_LINEAR, int(geglu)
    _fill_epilogue(d, N, M, bias, rowbias, rows_per_group, residual, alpha, act, cols)
    if rowbias_lo is not None:
        rl = _mat(rowbias_lo, "rowbias_lo")
        if rowbias is None or rl.shape != rowbias.shape or rl.stride(0) != rowbias.stride(0):
            raise ValueError("rowbias_lo: needs rowbias of the same shape and row stride")
        d.rowbias_lo = rl.data_ptr()
    if ln is not None:
        cs, cb, eps = ln
        for v, nm in ((cs, "ln colsum"), (cb, "ln colbias")):
            if v.dtype != torch.float32 or not _on_gpu(v) or not v.is_contiguous() or v.numel() != N:
                raise ValueError(f"{nm}: expected a contiguous CUDA fp32 tensor of {N} elements")
        if bias is not None or rowbias is not None or a2 is not None:
            raise ValueError("gemm(ln=...): the bias is part of colbias; no rowbias / second source")
        d.ln_colsum, d.ln_colbias, d.ln_eps = cs.data_ptr(), cb.data_ptr(), float(eps)
    keep = (a, a2, w, o, bias, rowbias, rowbias_lo, residual, alpha, ln)
    if carry and not geglu and ln is None:
        keep = _carry_setup(d, o, residual, carry, keep)
    _launch_gemm(d, "mv_gemm_f16", a.device, keep, o if colstats and not geglu else None)
    return o


# The level-0 feed-forward as one launch (mv_ffn_geglu_f16; A/B: MUSEV_OPS="FFN_FUSED=0" keeps LayerNorm + the GEGLU projection + the
# output projection)
FFN_FUSED: bool = True
FFN_FUSED_HITS: int = 0
FFN_ROTATE: bool = True   # mv_ffn_desc.flags bit 0


def ffn_fused_applies(c: int, hidden: int) -> bool:
    return FFN_FUSED and c == 320 and hidden == 1280


def ffn_geglu(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float, w1p: torch.Tensor, b1p: Optional[torch.Tensor],
              w2: torch.Tensor, b2: Optional[torch.Tensor], residual: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """residual + (GEGLU(LayerNorm(x) @ w1p.T + b1p)) @ w2.T + b2 in ONE launch (C = 320, hidden = 1280): ``w1p`` / ``b1p`` are the
    GEGLU-packed first projection (:func:`pack_geglu`), ``w2`` the torch-layout [C, hidden] output projection."""
    x = _mat(x, "x")
    residual = _mat(residual, "residual")
    w1p = _mat(w1p, "w1")
    w2 = _mat(w2, "w2")
    M, c = x.shape
    hidden = w2.shape[1]
    if not (w1p.is_contiguous() and w2.is_contiguous() and tuple(w1p.shape) == (2 * hidden, c) and w2.shape[0] == c and tuple(residual.shape) == (M, c)):
        raise ValueError("ffn_geglu: shape mismatch")
    _vec(gamma, "gamma", c)
    _vec(beta, "beta", c)
    _vec(b1p, "bias1", 2 * hidden)
    _vec(b2, "bias2", c)
    o = _out(out, M, c, x)
    d = _lib.FfnDesc()
    d.x, d.ln_gamma, d.ln_beta, d.w1, d.bias1 = x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), w1p.data_ptr(), _p(b1p)
    d.w2, d.bias2, d.residual, d.out = w2.data_ptr(), _p(b2), residual.data_ptr(), o.data_ptr()
    d.M, d.C, d.H, d.ldx, d.ldr, d.ldo, d.ln_eps = M, c, hidden, x.stride(0), residual.stride(0), o.stride(0), float(eps)
    d.flags = int(FFN_ROTATE)
    check(_lib.load().mv_ffn_geglu_f16(C.byref(d), _stream()), "mv_ffn_geglu_f16")
    global FFN_FUSED_HITS
    FFN_FUSED_HITS += 1
    if GEMM_RECORD is not None:  # the fused feed-forward is matrix work of the same family: recorded next to the mv_gemm_f16 launches
        nbytes = 2 * (3 * M * c + 3 * hidden * c)   # x, the residual and the output once; both weight matrices once
        GEMM_RECORD.append((_lib.FfnDesc.from_buffer_copy(d), (x, gamma, beta, w1p, b1p, w2, b2, residual, o), nbytes, _stream()))
    return o


# One temporal self-attention sub-block as one launch (mv_temporal_attn_block_f16; A/B: MUSEV_OPS="TSA_FUSED=0" keeps the LayerNorm-folded
# QKV projection + mv_temporal_attention_f16 + to_out)
TSA_FUSED: bool = True
TSA_FUSED_HITS: int = 0


def tsa_fused_applies(c: int, heads: int, d: int, t: int, hw: int) -> bool:
    return TSA_FUSED and c == 320 and heads == 8 and d == 40 and 1 <= t <= 16 and hw % 8 == 0


def pack_tsa_qkv(wq: torch.Tensor, wk: torch.Tensor, wv: torch.Tensor, heads: int, d: int) -> torch.Tensor:
    """[heads * d, C] x 3 -> [heads, 128, C]: per head the rows [q_h (d) | k_h (d) | v_h (d) | zero rows up to 128]"""
    c = wq.shape[1]
    out = torch.zeros(heads, 128, c, dtype=torch.float16, device=wq.device)
    for i, w in enumerate((wq, wk, wv)):
        out[:, i * d:(i + 1) * d] = w.reshape(heads, d, c).to(torch.float16)
    return out.contiguous()


def pack_tsa_out(wo: torch.Tensor, heads: int, d: int) -> torch.Tensor:
    """[C, heads * d] -> [C, heads * 64]: column 64 h + j = column d h + j for j < d, zero up to 64"""
    c = wo.shape[0]
    out = torch.zeros(c, heads, 64, dtype=torch.float16, device=wo.device)
    out[:, :, :d] = wo.reshape(c, heads, d).to(torch.float16)
    return out.reshape(c, heads * 64).contiguous()


def temporal_attn_block(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float, wqkv_p: torch.Tensor, wo_p: torch.Tensor,
                        bias_o: Optional[torch.Tensor], b: int, t: int, hw: int, heads: int, d: int, scale: float,
                        out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """x + to_out(softmax_T(q k^T scale) v) with [q | k | v] = LayerNorm(x) Wqkv^T over the t frames of every pixel, in ONE launch
    (C = 320, 8 heads x 40, t <= 16): rows of ``x`` in (b, t, p) order; ``wqkv_p`` / ``wo_p`` from :func:`pack_tsa_qkv` / :func:`pack_tsa_out`."""
    x = _mat(x, "x")
    M, c = x.shape
    if M != b * t * hw or not wqkv_p.is_contiguous() or not wo_p.is_contiguous() or tuple(wqkv_p.shape) != (heads, 128, c) or \
            tuple(wo_p.shape) != (c, heads * 64) or not _on_gpu(wqkv_p) or not _on_gpu(wo_p):
        raise ValueError("temporal_attn_block: shape mismatch")
    _vec(gamma, "gamma", c)
    _vec(beta, "beta", c)
    _vec(bias_o, "bias_o", c)
    o = _out(out, M, c, x)
    ds = _lib.TsaDesc()
    ds.x, ds.ln_gamma, ds.ln_beta, ds.wqkv, ds.wo, ds.bias_o, ds.out = (x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), wqkv_p.data_ptr(),
                                                                       wo_p.data_ptr(), _p(bias_o), o.data_ptr())
    ds.B, ds.T, ds.HW, ds.C, ds.heads, ds.d = b, t, hw, c, heads, d
    ds.ldx, ds.ldo, ds.ln_eps, ds.scale, ds.flags = x.stride(0), o.stride(0), float(eps), float(scale), int(FFN_ROTATE)
    check(_lib.load().mv_temporal_attn_block_f16(C.byref(ds), _stream()), "mv_temporal_attn_block_f16")
    global TSA_FUSED_HITS
    TSA_FUSED_HITS += 1
    if GEMM_RECORD is not None:  # matrix work of the same family (the q / k / v projection + to_out): recorded next to the mv_gemm_f16 launches
        nbytes = 2 * (3 * M * c + 4 * c * c)   # x twice (rows, residual) and the output once; the four weight matrices once
        GEMM_RECORD.append((_lib.TsaDesc.from_buffer_copy(ds), (x, gamma, beta, wqkv_p, wo_p, bias_o, o), nbytes, _stream()))
    return o



# The text cross-attention sub-block of level 0 as one launch (mv_xattn_block_f16; A/B: MUSEV_OPS="XAB_FUSED=0" keeps the LayerNorm-folded
# to_q projection + the resident-K/V cross-attention + to_out)
XAB_FUSED: bool = True
XAB_FUSED_HITS: int = 0


def xab_fused_applies(c: int, heads: int, d: int, n_keys: int, rows_per_kvb: int) -> bool:
    return XAB_FUSED and c == 320 and heads == 8 and d == 40 and 1 <= n_keys <= 80 and rows_per_kvb % 128 == 0


def pack_xab_q(wq: torch.Tensor, heads: int, d: int) -> torch.Tensor:
    """to_q.weight [heads * d, C] -> [heads / 2][128][C]: per head pair the rows [q_a (d) | zero rows up to 64 | q_b (d) | zero rows up to 128]"""
    c = wq.shape[1]
    out = torch.zeros(heads // 2, 2, 64, c, dtype=torch.float16, device=wq.device)
    out[:, :, :d] = wq.reshape(heads // 2, 2, d, c).to(torch.float16)
    return out.reshape(heads // 2, 128, c).contiguous()


def xattn_block(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float, wq_p: torch.Tensor, k: torch.Tensor, v: torch.Tensor,
                n_keys: int, rows_per_kvb: int, wo_p: torch.Tensor, bias_o: Optional[torch.Tensor], heads: int, d: int, scale: float,
                out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """x + to_out(softmax(q k^T scale) v) with q = LayerNorm(x) Wq^T in ONE launch (C = 320, 8 heads x 40, <= 80 keys, one softmax group):
    ``k`` / ``v`` = the prompt's projected keys / values [key batches * n_keys, >= C] (row-major, unit inner stride); rows
    [b * rows_per_kvb, (b + 1) * rows_per_kvb) of ``x`` attend to key batch b; ``wq_p`` / ``wo_p`` from :func:`pack_xab_q` / :func:`pack_tsa_out`."""
    x, k, v = _mat(x, "x"), _mat(k, "k"), _mat(v, "v")
    M, c = x.shape
    nkvb = (M + rows_per_kvb - 1) // rows_per_kvb
    if tuple(wq_p.shape) != (heads // 2, 128, c) or tuple(wo_p.shape) != (c, heads * 64) or not wq_p.is_contiguous() or not wo_p.is_contiguous() or \
            k.shape[0] < nkvb * n_keys or v.shape[0] < nkvb * n_keys or k.shape[1] != c or v.shape[1] != c or not _on_gpu(wq_p) or not _on_gpu(wo_p):
        raise ValueError("xattn_block: shape mismatch")
    _vec(gamma, "gamma", c)
    _vec(beta, "beta", c)
    _vec(bias_o, "bias_o", c)
    o = _out(out, M, c, x)
    ds = _lib.XabDesc()
    ds.x, ds.ln_gamma, ds.ln_beta, ds.wq, ds.k, ds.v = x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), wq_p.data_ptr(), k.data_ptr(), v.data_ptr()
    ds.wo, ds.bias_o, ds.out = wo_p.data_ptr(), _p(bias_o), o.data_ptr()
    ds.M, ds.rows_per_kvb, ds.len, ds.C, ds.heads, ds.d = M, rows_per_kvb, n_keys, c, heads, d
    ds.ldk, ds.ldv, ds.ldx, ds.ldo, ds.ln_eps, ds.scale, ds.flags = k.stride(0), v.stride(0), x.stride(0), o.stride(0), float(eps), float(scale), int(FFN_ROTATE)
    check(_lib.load().mv_xattn_block_f16(C.byref(ds), _stream()), "mv_xattn_block_f16")
    global XAB_FUSED_HITS
    XAB_FUSED_HITS += 1
    if GEMM_RECORD is not None:  # matrix work of the same family (the q projection + to_out): recorded next to the mv_gemm_f16 launches
        nbytes = 2 * (3 * M * c + 2 * c * c)   # x twice (rows, residual) and the output once; the two weight matrices once
        GEMM_RECORD.append((_lib.XabDesc.from_buffer_copy(ds), (x, gamma, beta, wq_p, k, v, wo_p, bias_o, o), nbytes, _stream()))
    return o


def replay_gemms_two_streams(rec_a: Sequence[tuple], rec_b: Sequence[tuple], reps: int = 1, side=None) -> float:
    """the two lists re-issued CONCURRENTLY, one per HIP stream (how the loop runs the two CFG halves); returns the elapsed device
    milliseconds from the common start to the later of the two ends, over all reps.  ``side``: the second stream (the loop's own side
    stream; a fresh stream per call would walk through HIP's few hardware queues and sooner or later share the main stream's)"""
    lib = _lib.load()
    main = torch.cuda.current_stream()
    if side is None:
        from .pipelines.parallel_denoise import ParallelDenoiser
        side = ParallelDenoiser._shared_stream("side", main.device)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(main)
    side.wait_stream(main)
    for _ in range(reps):
        for d, _keep, _nb, *_ in rec_b:
            _replay_one(lib, d, side.cuda_stream)
        for d, _keep, _nb, *_ in rec_a:
            _replay_one(lib, d, main.cuda_stream)
    main.wait_stream(side)
    e1.record(main)
    e1.synchronize()
    return e0.elapsed_time(e1)


# LayerNorm folding (A/B: MUSEV_OPS="LN_FOLD=0" keeps mv_layernorm_f16 + the plain projection everywhere)
LN_FOLD: bool = True
# ... and only for projections with K <= LN_FOLD_MAX_K: beside the pair-timed tile table (round 5) the 1 280-wide levels run faster as
# mv_layernorm_f16 + a plain projection on a 256 x 256 / 256 x 320 tile than folded (the folded tiles stop at 256 x 256 and carry the
# in-loop row statistics): same-box alternating legs 49.65 -> 49.27 ms per config-2 step at 640, 49.35 with no folding at all
# (profiles/r05zf_ab_lnfold{,2}.log; per launch of a pair 32 against 39 us at 3 328 x 3 840 x 1 280).  Round 6, 12 rotated same-box rounds
# per leg (profiles/r06zw_*): 320 against 640 is -0.05 ms at config 2 (47.55 / 47.60), **-0.47 ms at config 3** (54.87 / 55.33), nothing at
# config 5 (898.2 / 897.7) -- the 640-wide level folds no more either.
LN_FOLD_MAX_K: int = 320
_ln_fold_cache: dict = {}


def ln_fold_applies(M: int, N: int, K: int, geglu: bool) -> bool:
    """whether ``gemm(..., ln=)`` is the better form of LayerNorm + projection for this problem: it is wherever the plain
    projection runs as ONE K slice (the folded kernel needs the whole row in one block's K loop); the small-M / long-K problems
    the library splits over K keep mv_layernorm_f16 + the split GEMM, and so do the levels wider than LN_FOLD_MAX_K (above)."""
    if not LN_FOLD or K % 64 != 0 or geglu or K > LN_FOLD_MAX_K:
        # (GEGLU: the gate's epilogue already bounds that launch -- folded it measured 4-11 % SLOWER than LayerNorm + GEMM at every
        # level, profiles/r03e_ln_fold_variants.log; the q / k / v projections gain 5-44 %.  Row statistics emitted by the PRODUCER's
        # epilogue instead of the in-loop ones -- which would also let the GEGLU launch fold -- measured +0.6 ms per step:
        # profiles/r03o_rowstats_ab.log, commit 35438ca; not kept.)
        return False
    key = (M, N, K, bool(geglu), GEMM_CFG, GEMM_SPLITK)   # (the tuner / A-B tools change the forced configuration at run time)
    hit = _ln_fold_cache.get(key)
    if hit is None:
        d = GemmDesc()
        d.a, d.w, d.c = 16, 16, 16  # (no launch: the choice only looks at the geometry)
        d.M, d.N, d.K, d.lda, d.ldc, d.c1 = M, N, K, K, (N // 2 if geglu else N), K
        d.mode, d.geglu, d.cfg, d.splitk = MV_GEMM_LINEAR, int(geglu), GEMM_CFG, GEMM_SPLITK
        cfg, ns = C.c_int32(), C.c_int32()
        check(_lib.load().mv_gemm_choice(C.byref(d), C.byref(cfg), C.byref(ns)), "mv_gemm_choice")
        hit = _ln_fold_cache[key] = ns.value == 1
    return hit


def fold_layernorm(w: torch.Tensor, bias: Optional[torch.Tensor], gamma: torch.Tensor, beta: torch.Tensor):
    """(w * gamma as fp16 [N, K], colsum fp32 [N], colbias fp32 [N]) for ``gemm(..., ln=)``:
    LayerNorm(x) @ W.T + b = rstd * (x @ (W gamma).T - mean * colsum) + (W @ beta + b);  colsum sums the fp16-ROUNDED folded weights
    (what the matrix cores multiply), so that the mean term cancels exactly."""
    wf = (w.float() * gamma.float().reshape(1, -1)).to(torch.float16).contiguous()
    colsum = wf.float().sum(dim=1).contiguous()
    colbias = w.float() @ beta.float().reshape(-1)
    if bias is not None:
        colbias = colbias + bias.float()
    return wf, colsum, colbias.contiguous()


def conv3x3(x: torch.Tensor, w: torch.Tensor, n_img: int, h: int, w_: int, *, x2: Optional[torch.Tensor] = None,
            stride: int = 1, upsample: bool = False, bias=None, rowbias=None, rows_per_group: int = 0, residual=None,
            out: Optional[torch.Tensor] = None, carry: bool = False) -> torch.Tensor:
    """3x3 convolution, padding 1, over channels-last images x = [n_img*h*w_, C1] (+ optional concatenated x2).

    ``w`` is the packed weight [Cout, 9*(C1+C2)] (tap-major, see :func:`pack_conv_weight`).  ``upsample`` applies a
    nearest x2 upsample to the input on the fly (Upsample2D); ``stride=2`` is Downsample2D.
    """
    x = _mat(x, "x")
    w = _mat(w, "w")
    c1 = x.shape[1]
    c2 = 0
    if x.shape[0] != n_img * h * w_:
        raise ValueError("conv3x3: x rows != n_img*h*w")
    d = GemmDesc()
    d.a, d.lda, d.c1 = x.data_ptr(), x.stride(0), c1
    if x2 is not None:
        x2 = _mat(x2, "x2")
        c2 = x2.shape[1]
        d.a2, d.lda2, d.c2 = x2.data_ptr(), x2.stride(0), c2
    N, K = w.shape
    if K != 9 * (c1 + c2):
        raise ValueError(f"conv3x3: weight K={K} != 9*({c1}+{c2})")
    if upsample:
        ho, wo = 2 * h, 2 * w_
    else:
        ho, wo = (h + 2 - 3) // stride + 1, (w_ + 2 - 3) // stride + 1
    M = n_img * ho * wo
    o = _out(out, M, N, x)
    d.w, d.c, d.ldc = w.data_ptr(), o.data_ptr(), o.stride(0)
    d.M, d.N, d.K = M, N, K
    d.mode, d.stride, d.upsample = MV_GEMM_CONV3X3, stride, int(upsample)
    d.hin, d.win, d.hout, d.wout = h, w_, ho, wo
    _fill_epilogue(d, N, M, bias, rowbias, rows_per_group, residual, None, MV_ACT_NONE, N)
    _launch_gemm(d, "mv_gemm_f16(conv3x3)", x.device, _carry_setup(d, o, residual, carry, (x, x2, w, o, bias, rowbias, residual)), o)
    return o


def tconv3(x: torch.Tensor, w: torch.Tensor, b: int, t: int, hw: int, *, bias=None, residual=None, alpha=None,
           out: Optional[torch.Tensor] = None, carry: bool = False) -> torch.Tensor:
    """Conv3d (3,1,1), padding (1,0,0), over x = [b*t*hw, C] in (b, t, p) row order; w packed [Cout, 3*C]."""
    x = _mat(x, "x")
    w = _mat(w, "w")
    M, c1 = x.shape
    if M != b * t * hw:
        raise ValueError("tconv3: x rows != b*t*hw")
    N, K = w.shape
    if K != 3 * c1:
        raise ValueError("tconv3: weight K != 3*C")
    o = _out(out, M, N, x)
    d = GemmDesc()
    d.a, d.lda, d.c1 = x.data_ptr(), x.stride(0), c1
    d.w, d.c, d.ldc = w.data_ptr(), o.data_ptr(), o.stride(0)
    d.M, d.N, d.K = M, N, K
    d.mode, d.t, d.hw = MV_GEMM_TCONV3, t, hw
    _fill_epilogue(d, N, M, bias, None, 0, residual, alpha, MV_ACT_NONE, N)
    _launch_gemm(d, "mv_gemm_f16(tconv3)", x.device, _carry_setup(d, o, residual, carry, (x, w, o, bias, residual, alpha)), o)
    return o


def groupnorm(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, n_items: int, rows: int, *, eps: float,
              silu: bool, x2: Optional[torch.Tensor] = None, groups: int = 32,
              out: Optional[torch.Tensor] = None, carry: bool = False) -> torch.Tensor:
    """GroupNorm(groups) (+SiLU) with statistics over (rows, C/groups) per item; x = [n_items*rows, C1] (+ x2).
    ``carry`` (conv_norm_out): normalise x + its lo half (see CARRY) in fp32 and hand the result on as two fp16 halves as well."""
    x = _mat(x, "x")
    c1 = x.shape[1]
    c2 = 0
    if x.shape[0] != n_items * rows:
        raise ValueError("groupnorm: x rows != n_items*rows")
    if x2 is not None:
        x2 = _mat(x2, "x2")
        c2 = x2.shape[1]
    c = c1 + c2
    _vec(gamma, "gamma", c)
    _vec(beta, "beta", c)
    lib = _lib.load()
    if (gamma.data_ptr() | beta.data_ptr()) & 15:
        raise ValueError("groupnorm: gamma / beta must be 16-byte aligned")
    nsplit = lib.mv_groupnorm_default_nsplit(n_items, rows, c)
    o = _out(out, n_items * rows, c, x)
    x_lo = y_lo = None
    if carry and CARRY and x2 is None and c <= CARRY_MAX_C and o.stride(0) % 8 == 0:
        x_lo = _lo_of(x)
        y_lo = torch.empty_strided(o.shape, o.stride(), dtype=torch.float16, device=o.device)
        _set_lo(o, y_lo)
    cs1 = getattr(x, "_mv_colstats", None) if COLSTATS else None
    cs2 = getattr(x2, "_mv_colstats", None) if (COLSTATS and x2 is not None) else None
    if cs1 is not None and len(cs1) > 2 and cs1[2] != x._version:
        cs1 = None   # the tensor was modified in place after its producer wrote the statistics
    if cs2 is not None and len(cs2) > 2 and cs2[2] != x2._version:
        cs2 = None
    if cs1 is not None and rows % cs1[1] == 0 and (x2 is None or (cs2 is not None and rows % cs2[1] == 0)):
        # the producers of x (and x2) left column statistics behind: fold those instead of reading the tensors once more
        global COLSTATS_HITS
        COLSTATS_HITS += 1
        stat = torch.empty(n_items * 2 * groups, dtype=torch.float32, device=x.device)
        check(lib.mv_groupnorm_cs_f16(x.data_ptr(), _p(x2), c1, c2, x.stride(0), x2.stride(0) if x2 is not None else 0,
                                      n_items, rows, groups, float(eps), gamma.data_ptr(), beta.data_ptr(), int(silu),
                                      o.data_ptr(), o.stride(0), cs1[0].data_ptr(), cs1[1],
                                      cs2[0].data_ptr() if cs2 is not None else None, cs2[1] if cs2 is not None else 0,
                                      nsplit, stat.data_ptr(), _p(x_lo), _p(y_lo), _stream()), "mv_groupnorm_cs_f16")
        return o
    scratch = torch.empty(n_items * nsplit * 2 * groups + n_items * 2 * groups, dtype=torch.float32, device=x.device)
    partial_ptr = scratch.data_ptr()
    stat_ptr = partial_ptr + 4 * n_items * nsplit * 2 * groups
    check(lib.mv_groupnorm_f16(x.data_ptr(), _p(x2), c1, c2, x.stride(0), x2.stride(0) if x2 is not None else 0,
                               n_items, rows, groups, float(eps), gamma.data_ptr(), beta.data_ptr(), int(silu),
                               o.data_ptr(), o.stride(0), partial_ptr, nsplit, stat_ptr, _p(x_lo), _p(y_lo), _stream()), "mv_groupnorm_f16")
    return o



# GroupNorm without activation folded into the projection behind it (Transformer2DModel.norm -> proj_in, TransformerTemporalModel.norm ->
# proj_in): the apply pass -- an HBM round trip of the activation -- is replaced by one scaled copy of the projection's weights per
# normalised item (mv_groupnorm_cs_fold_linear_f16) and a per-item row bias.  Taken where the copies are small next to the tensor:
# items * N * C <= GN_FOLD_MAX_RATIO * rows_total * C.  A/B: MUSEV_OPS="GN_FOLD=0" keeps groupnorm() + the plain projection.
GN_FOLD: bool = True
GN_FOLD_MAX_RATIO: float = 0.35
GN_FOLD_HITS: int = 0


def groupnorm_fold_linear(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, n_items: int, rows: int, *, eps: float, groups: int,
                          w: torch.Tensor, bias: Optional[torch.Tensor], rowbias: Optional[torch.Tensor] = None, rb_per_item: int = 1,
                          out: Optional[torch.Tensor] = None) -> Optional[torch.Tensor]:
    """Linear(GroupNorm(x)) (+ rowbias[row // (rows / rb_per_item)]) with the normalisation folded into per-item weights; ``None`` when
    the fold does not apply (no producer statistics on ``x``, or the weight copies would not be small next to the tensor) -- the caller
    then takes :func:`groupnorm` + :func:`gemm`."""
    if not GN_FOLD:
        return None
    x = _mat(x, "x")
    w = _mat(w, "w")
    M, c = x.shape
    N = w.shape[0]
    cs = getattr(x, "_mv_colstats", None) if COLSTATS else None
    if cs is None or (len(cs) > 2 and cs[2] != x._version) or rows % cs[1] or M != n_items * rows or w.shape[1] != c or not w.is_contiguous():
        return None
    if n_items * N > GN_FOLD_MAX_RATIO * M or rows % 32 or rows % rb_per_item or c % 8 or c > 2048 or c % groups:
        return None
    _vec(gamma, "gamma", c)
    _vec(beta, "beta", c)
    _vec(bias, "bias", N)
    rb_in = None
    if rowbias is not None:
        rb_in = _mat(rowbias, "rowbias")
        if rb_in.shape[0] != n_items * rb_per_item or rb_in.shape[1] != N:
            raise ValueError("groupnorm_fold_linear: rowbias must be [n_items * rb_per_item, N]")
    dev = x.device
    w_f = torch.empty((n_items * N, c), dtype=torch.float16, device=dev)
    rb = torch.empty((2, n_items * rb_per_item, N), dtype=torch.float16, device=dev)
    stat = torch.empty(n_items * 2 * groups, dtype=torch.float32, device=dev)
    check(_lib.load().mv_groupnorm_cs_fold_linear_f16(cs[0].data_ptr(), cs[1], c, n_items, rows, groups, float(eps), gamma.data_ptr(), beta.data_ptr(),
                                                      w.data_ptr(), _p(bias), N, _p(rb_in), rb_in.stride(0) if rb_in is not None else 0, rb_per_item,
                                                      w_f.data_ptr(), rb[0].data_ptr(), rb[1].data_ptr(), stat.data_ptr(), _stream()),
          "mv_groupnorm_cs_fold_linear_f16")
    global GN_FOLD_HITS
    GN_FOLD_HITS += 1
    return gemm(x, w_f, rowbias=rb[0], rowbias_lo=rb[1], rows_per_group=rows // rb_per_item, w_groups=n_items, out=out)


def layernorm(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float = 1e-5,
              out: Optional[torch.Tensor] = None) -> torch.Tensor:
    x = _mat(x, "x")
    M, c = x.shape
    _vec(gamma, "gamma", c)
    _vec(beta, "beta", c)
    o = _out(out, M, c, x)
    check(_lib.load().mv_layernorm_f16(x.data_ptr(), x.stride(0), o.data_ptr(), o.stride(0), M, c, gamma.data_ptr(),
                                       beta.data_ptr(), float(eps), _stream()), "mv_layernorm_f16")
    return o


Seg = Tuple[torch.Tensor, torch.Tensor, int, int, int, int]  # (k, v, len, div, mul, add)
# ATTN_GROUPS = False (A/B): the image-prompt terms of the cross-attention as separate accumulate launches, as in round 2
ATTN_GROUPS: bool = True
# Attentions over at most 128 keys (the text / image-prompt cross-attention at head dims 40 / 80) run on the resident-K/V kernel
# (mv_attn_desc.resident_kv: whole query rows per block, q read once, out written once).  Round 5 (profiles/r05f_xattn_bench.log):
# level 0, 13 frames: 30.4 us against 43.1 tiled (text), 35.5 against 64.6 (text + IP-Adapter group); level 1: 20.7 / 22.6; in the
# two-stream step within noise at config 2, -0.15 ms at config 3 (r05f_ab_config*.log).  0 = the tiled kernel; a value >= 16 = that
# many query rows per block instead of the launcher's choice (tools/gpu_xattn_bench.py sweeps it).
XATTN_RESIDENT: int = 1
XATTN_RESIDENT_HITS: int = 0
# ... up to this head dim.  Round 6, rotated same-box A/B legs of the whole step (profiles/r06zp ... r06zs): at d = 80 (level 1: the kernels
# are equal alone, 20.7 against 22.6 us) the tiled kernel is worth -0.2 ms per config-2 step (-0.10 / -0.21 / -0.23 on three boxes): the
# resident kernel's 512-thread blocks wait longer for a compute unit beside the other half's one-block-per-CU tiles.  Level 0 (d = 40)
# keeps it (35.5 against 64.6 us with the IP-Adapter group).  A/B: MUSEV_OPS="XATTN_RESIDENT_MAX_D=80".
XATTN_RESIDENT_MAX_D: int = 40


def attention(q: torch.Tensor, segs: Sequence[Seg], nb: int, lq: int, heads: int, d: int, scale: float, *,
              out: Optional[torch.Tensor] = None, accumulate: bool = False, out_scale: float = 1.0,
              group_scales: Optional[Sequence[Optional[float]]] = None) -> torch.Tensor:
    """Multi-segment softmax attention.  q = [nb*lq, heads*d]; each segment (k, v, len, div, mul, add) holds 2-D
    key/value matrices whose row (kvb*len + j) is key j of key batch kvb = (n // div) * mul + add for query batch n.
    ``group_scales`` (one entry per segment, head dims 40 / 80): a float starts a new softmax GROUP of that weight at the segment,
    ``None`` continues the group of the segment before: out = sum_g scale_g * softmax_g(q K_g^T) V_g in one launch -- the text
    cross-attention plus ip_adapter_scale * the image-prompt attention (attention_processor.py:258-300).  Default: one group."""
    q = _mat(q, "q")
    if q.shape[0] != nb * lq or q.shape[1] != heads * d:
        raise ValueError("attention: q shape mismatch")
    o = _out(out, nb * lq, heads * d, q)
    ds = AttnDesc()
    ds.q, ds.out, ds.ldq, ds.ldo = q.data_ptr(), o.data_ptr(), q.stride(0), o.stride(0)
    ds.nb, ds.lq, ds.heads, ds.d = nb, lq, heads, d
    ds.scale, ds.nseg = float(scale), len(segs)
    ds.accumulate, ds.out_scale = int(accumulate), float(out_scale)
    if not 1 <= len(segs) <= _lib.MV_ATTN_MAX_SEG:
        raise ValueError("attention: need 1..4 segments")
    for i, (k, v, ln, div, mul, add_) in enumerate(segs):
        k = _mat(k, "k")
        v = _mat(v, "v")
        max_kvb = ((nb - 1) // div) * mul + add_
        if k.shape[0] < (max_kvb + 1) * ln or v.shape[0] < (max_kvb + 1) * ln:
            raise ValueError(f"attention: segment {i} does not cover key batch {max_kvb}")
        s = ds.seg[i]
        s.k, s.v, s.ldk, s.ldv, s.len, s.div, s.mul, s.add = k.data_ptr(), v.data_ptr(), k.stride(0), v.stride(0), ln, div, mul, add_
        if group_scales is not None:
            if len(group_scales) != len(segs) or group_scales[0] is None:
                raise ValueError("attention: group_scales needs one entry per segment, the first one a weight")
            if group_scales[i] is not None:
                s.new_group, s.group_scale = 1, float(group_scales[i])
    lib = _lib.load()
    if XATTN_RESIDENT and d <= XATTN_RESIDENT_MAX_D and not accumulate and lib.mv_attention_resident_ok(C.byref(ds)):
        global XATTN_RESIDENT_HITS
        XATTN_RESIDENT_HITS += 1
        ds.resident_kv = max(1, int(XATTN_RESIDENT))
    check(lib.mv_attention_f16(C.byref(ds), _stream()), "mv_attention_f16")
    return o


def temporal_attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, b: int, t: int, hw: int, heads: int, d: int,
                       scale: float, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    q, k, v = _mat(q, "q"), _mat(k, "k"), _mat(v, "v")
    M = b * t * hw
    if q.shape[0] != M or k.shape[0] != M or v.shape[0] != M:
        raise ValueError("temporal_attention: rows != b*t*hw")
    o = _out(out, M, heads * d, q)
    check(_lib.load().mv_temporal_attention_f16(q.data_ptr(), k.data_ptr(), v.data_ptr(), q.stride(0), k.stride(0),
                                                v.stride(0), o.data_ptr(), o.stride(0), b, t, hw, heads, d,
                                                float(scale), _stream()), "mv_temporal_attention_f16")
    return o


def geglu(x: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    x = _mat(x, "x")
    M, two = x.shape
    half = two // 2
    o = _out(out, M, half, x)
    check(_lib.load().mv_geglu_f16(x.data_ptr(), x.stride(0), o.data_ptr(), o.stride(0), M, half, _stream()), "mv_geglu_f16")
    return o


def softmax_rows_(x: torch.Tensor) -> torch.Tensor:
    """in-place softmax over the columns of an fp16 [rows, cols] matrix (unit inner stride, any row stride % 8)"""
    x = _mat(x, "x")
    if x.data_ptr() & 15 or x.stride(0) % 8 or x.shape[1] % 8:
        raise ValueError("softmax_rows_: x must be 16-byte aligned with cols and row stride in multiples of 8 (16-byte accesses)")
    check(_lib.load().mv_softmax_rows_f16(x.data_ptr(), x.stride(0), x.shape[0], x.shape[1], _stream()), "mv_softmax_rows_f16")
    return x


def silu(x: torch.Tensor) -> torch.Tensor:
    if not x.is_contiguous() or x.dtype != torch.float16:
        raise ValueError("silu: contiguous fp16 expected")
    y = torch.empty_like(x)
    check(_lib.load().mv_silu_f16(x.data_ptr(), y.data_ptr(), x.numel(), _stream()), "mv_silu_f16")
    return y


def add(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    if a.shape != b.shape or not a.is_contiguous() or not b.is_contiguous() or a.dtype != torch.float16 or b.dtype != torch.float16:
        raise ValueError("add: contiguous fp16 tensors of equal shape expected")
    y = torch.empty_like(a)
    a_lo = _lo_of(a) if CARRY else None   # a carried stream tensor keeps both halves through the add (see mv_add_f16)
    y_lo = torch.empty_like(a) if a_lo is not None else None
    check(_lib.load().mv_add_f16(a.data_ptr(), _p(a_lo), b.data_ptr(), y.data_ptr(), _p(y_lo), a.numel(), _stream()), "mv_add_f16")
    if y_lo is not None:
        _set_lo(y, y_lo)
    return y


def conv3x3_cin_small(x: torch.Tensor, w: torch.Tensor, bias, n_img: int, h: int, w_: int,
                      add_: Optional[torch.Tensor] = None) -> torch.Tensor:
    x = _mat(x, "x")
    w = _mat(w, "w")
    cin, cout = x.shape[1], w.shape[0]
    if not x.is_contiguous() or w.shape[1] != 9 * cin:
        raise ValueError("conv3x3_cin_small: bad shapes")
    y = torch.empty((n_img * h * w_, cout), dtype=torch.float16, device=x.device)
    check(_lib.load().mv_conv3x3_cin_small_f16(x.data_ptr(), cin, w.data_ptr(), _p(_vec(bias, "bias", cout)),
                                               _p(add_), y.data_ptr(), cout, n_img, h, w_, _stream()),
          "mv_conv3x3_cin_small_f16")
    return y


def conv3x3_cin_small_gemm(x: torch.Tensor, w: torch.Tensor, bias, n_img: int, h: int, w_: int,
                           add_: Optional[torch.Tensor] = None, kpad: int = 64) -> torch.Tensor:
    """conv_in on the matrix cores: im2col into [rows, kpad] (one 64-deep K step) + the implicit-GEMM kernel.
    ``w`` is the packed conv weight [Cout, 9*Cin]; it is zero-padded to kpad columns here (callers cache the result
    through ``pad_cols``)."""
    x = _mat(x, "x")
    cin = x.shape[1]
    if not x.is_contiguous() or 9 * cin > kpad:
        raise ValueError("conv3x3_cin_small_gemm: bad shapes")
    a = torch.empty((n_img * h * w_, kpad), dtype=torch.float16, device=x.device)
    check(_lib.load().mv_im2col3x3_f16(x.data_ptr(), cin, a.data_ptr(), kpad, n_img, h, w_, _stream()), "mv_im2col3x3_f16")
    wp = w if w.shape[1] == kpad else pad_cols(w, kpad)
    return gemm(a, wp, bias=bias, residual=add_, carry=True)  # conv_in opens the residual stream


def pad_cols(w: torch.Tensor, k: int) -> torch.Tensor:
    out = torch.zeros((w.shape[0], k), dtype=w.dtype, device=w.device)
    out[:, :w.shape[1]] = w
    return out


def conv3x3_cout_small(x: torch.Tensor, w: torch.Tensor, bias, n_img: int, h: int, w_: int,
                       out_dtype: torch.dtype = torch.float16) -> torch.Tensor:
    """conv_out (C -> 4): ``out_dtype=torch.float32`` keeps the fp32 accumulator (the noise prediction feeds CFG and
    the scheduler, which amplify a final fp16 rounding several times)."""
    x = _mat(x, "x")
    w = _mat(w, "w")
    cin, cout = x.shape[1], w.shape[0]
    if not x.is_contiguous() or w.shape[1] != 9 * cin:
        raise ValueError("conv3x3_cout_small: bad shapes")
    if out_dtype not in (torch.float16, torch.float32):
        raise ValueError("conv3x3_cout_small: out_dtype must be fp16 or fp32")
    y = torch.empty((n_img * h * w_, cout), dtype=out_dtype, device=x.device)
    x_lo = _lo_of(x) if CARRY else None   # two-fp16 input (groupnorm(carry=True)): the convolution reads hi + lo
    if x_lo is not None and not x_lo.is_contiguous():
        x_lo = None
    check(_lib.load().mv_conv3x3_cout_small_f16(x.data_ptr(), _p(x_lo), cin, w.data_ptr(), _p(_vec(bias, "bias", cout)),
                                                y.data_ptr(), int(out_dtype == torch.float32), cout, n_img, h, w_, _stream()),
          "mv_conv3x3_cout_small_f16")
    return y


def conv3x3_direct(x: torch.Tensor, w: torch.Tensor, bias, n_img: int, h: int, w_: int, *, stride: int = 1,
                   act: int = MV_ACT_NONE) -> torch.Tensor:
    """3x3 convolution, padding 1, stride 1 | 2, fused bias (+ SiLU), for small / odd channel counts (the PoseGuider conv
    stack): x = [n_img*h*w_, cin] contiguous, w packed [cout, 9*cin], cout % 8 == 0."""
    x = _mat(x, "x")
    w = _mat(w, "w")
    cin, cout = x.shape[1], w.shape[0]
    if not x.is_contiguous() or not w.is_contiguous() or w.shape[1] != 9 * cin or x.shape[0] != n_img * h * w_:
        raise ValueError("conv3x3_direct: bad shapes")
    if stride not in (1, 2):
        raise ValueError("conv3x3_direct: stride must be 1 or 2")
    ho, wo = (h + 2 - 3) // stride + 1, (w_ + 2 - 3) // stride + 1
    y = torch.empty((n_img * ho * wo, cout), dtype=torch.float16, device=x.device)
    check(_lib.load().mv_conv3x3_direct_f16(x.data_ptr(), cin, w.data_ptr(), _p(_vec(bias, "bias", cout)), y.data_ptr(), cout,
                                            n_img, h, w_, stride, int(act), _stream()), "mv_conv3x3_direct_f16")
    return y


def timestep_embedding(t: torch.Tensor, dim: int) -> torch.Tensor:
    t = t.to(dtype=torch.float32).contiguous()
    out = torch.empty((t.numel(), dim), dtype=torch.float16, device=t.device)
    check(_lib.load().mv_timestep_embedding_f16(t.data_ptr(), t.numel(), dim, out.data_ptr(), _stream()),
          "mv_timestep_embedding_f16")
    return out


def upsample_nearest(x: torch.Tensor, n_img: int, h: int, w_: int, ho: int, wo: int) -> torch.Tensor:
    """nearest-neighbour resize of channels-last images x = [n_img*h*w_, C] to an explicit (ho, wo): F.interpolate(size=(ho, wo), mode="nearest")
    (diffusers Upsample2D with output_size).  Only for latent sizes that are not multiples of 2^(number of upsamplers); the exact x 2 case
    is fused into :func:`conv3x3` (``upsample=True``)."""
    x = _mat(x, "x")
    if x.shape[0] != n_img * h * w_:
        raise ValueError("upsample_nearest: x rows != n_img*h*w")
    c = x.shape[1]
    if c % 8:
        raise ValueError("upsample_nearest: channels must be a multiple of 8")
    y = torch.empty((n_img * ho * wo, c), dtype=torch.float16, device=x.device)
    check(_lib.load().mv_upsample_nearest_f16(x.data_ptr(), x.stride(0), y.data_ptr(), y.stride(0), n_img, h, w_, ho, wo, c, _stream()),
          "mv_upsample_nearest_f16")
    return y


def zero_rows(x: torch.Tensor, row_idx: torch.Tensor) -> None:
    x = _mat(x, "x")
    idx = row_idx.to(dtype=torch.int32).contiguous()
    check(_lib.load().mv_zero_rows_f16(x.data_ptr(), x.stride(0), idx.data_ptr(), idx.numel(), x.shape[1], _stream()),
          "mv_zero_rows_f16")


def bcthw_to_bthwc(x: torch.Tensor) -> torch.Tensor:
    """[B, C, T, H, W] (fp16|fp32) -> fp16 [B*T*H*W, C] channels-last rows."""
    b, c, t, h, w = x.shape
    x = x.contiguous()
    if x.dtype not in (torch.float16, torch.float32):
        x = x.float()
    y = torch.empty((b * t * h * w, c), dtype=torch.float16, device=x.device)
    check(_lib.load().mv_bcthw_to_bthwc_f16(x.data_ptr(), int(x.dtype == torch.float32), y.data_ptr(), b, c, t, h * w,
                                            _stream()), "mv_bcthw_to_bthwc_f16")
    return y


def bthwc_to_bcthw(x: torch.Tensor, b: int, t: int, h: int, w: int, dtype=torch.float16) -> torch.Tensor:
    if x.dim() != 2 or not x.is_contiguous() or not _on_gpu(x) or x.dtype not in (torch.float16, torch.float32):
        raise ValueError("bthwc_to_bcthw: contiguous 2-D CUDA fp16|fp32 input expected")
    c = x.shape[1]
    y = torch.empty((b, c, t, h, w), dtype=dtype, device=x.device)
    check(_lib.load().mv_bthwc_to_bcthw_f16(x.data_ptr(), int(x.dtype == torch.float32), y.data_ptr(),
                                            int(dtype == torch.float32), b, c, t, h * w, _stream()), "mv_bthwc_to_bcthw_f16")
    return y


def window_gather(latents: torch.Tensor, cond: Optional[torch.Tensor], idx: torch.Tensor, n_cond: int, copies: int,
                  hi_lo: bool = False, cond_slot: Optional[torch.Tensor] = None) -> torch.Tensor:
    """latents fp32 [C, T_total, HW]; cond fp32 [C, n_cond, HW]; idx int32 [win] -> fp16 [copies*(n_cond+win)*HW, C].
    ``cond_slot`` int32 [n_cond] (device): the window slot of every condition frame (the reference's vision_condition_latent_index;
    None = in front).  Slots < n_cond that no condition frame names are zero, the window's frames always fill n_cond.. (see
    include/musev_hip.h).
    ``hi_lo``: rows of 2 C columns [fp16(v) | fp16(v - fp16(v))] -- the fp32 latents as two fp16 halves (the UNet's conv_in takes
    them with its weight duplicated over the two channel groups: the input is not rounded to fp16, see CARRY)."""
    c, t_total, hw = latents.shape
    win = idx.numel()
    out = torch.empty((copies * (n_cond + win) * hw, 2 * c if hi_lo else c), dtype=torch.float16, device=latents.device)
    if cond_slot is not None and (cond_slot.dtype != torch.int32 or cond_slot.numel() != n_cond or not cond_slot.is_contiguous()):
        raise ValueError("window_gather: cond_slot must be contiguous int32 [n_cond]")
    check(_lib.load().mv_window_gather(latents.data_ptr(), _p(cond), idx.data_ptr(), _p(cond_slot), win, n_cond, c, t_total, hw, copies,
                                       int(hi_lo), out.data_ptr(), _stream()), "mv_window_gather")
    return out


def split_hi_lo(x32: torch.Tensor) -> torch.Tensor:
    """fp32 rows [M, C] -> fp16 [M, 2 C] = [hi | lo] (what window_gather(hi_lo=True) builds inside the loop), for callers that hand
    UNet3DConditionModel.forward an fp32 sample"""
    hi = x32.to(torch.float16)
    return torch.cat([hi, (x32 - hi.float()).to(torch.float16)], dim=1).contiguous()


def window_scatter_add(eps_win: torch.Tensor, idx: torch.Tensor, n_cond: int, halves: int, half_offset: int,
                       eps_acc: torch.Tensor, counter: torch.Tensor, add_counter: bool) -> None:
    """eps_win fp16|fp32 [halves*(n_cond+win)*HW, C]; eps_acc fp32 [H, C, T_total, HW]; counter fp32 [T_total]."""
    _, c, t_total, hw = eps_acc.shape
    win = idx.numel()
    if eps_win.dtype not in (torch.float16, torch.float32) or not eps_win.is_contiguous():
        raise ValueError("window_scatter_add: contiguous fp16|fp32 predictions expected")
    check(_lib.load().mv_window_scatter_add(eps_win.data_ptr(), int(eps_win.dtype == torch.float32), idx.data_ptr(), win,
                                            n_cond, c, t_total, hw, halves,
                                            half_offset, eps_acc.data_ptr(), counter.data_ptr(), int(add_counter),
                                            _stream()), "mv_window_scatter_add")


def window_units_reduce(units: torch.Tensor, table: torch.Tensor, eps_acc: torch.Tensor) -> None:
    """units fp32 [slots, win_max*HW, C] (the all-gathered predictions); table int32 [halves, T_total, maxc, 2] of (slot, j) pairs,
    -1 terminated; eps_acc fp32 [halves, C, T_total, HW] is OVERWRITTEN with the table-ordered sums."""
    halves, c, t_total, hw = eps_acc.shape
    if units.dtype != torch.float32 or not units.is_contiguous() or units.dim() != 3 or units.shape[2] != c:
        raise ValueError("window_units_reduce: units must be contiguous fp32 [slots, rows, C]")
    if table.dtype != torch.int32 or not table.is_contiguous() or table.shape[:2] != (halves, t_total) or table.shape[3] != 2:
        raise ValueError("window_units_reduce: table must be contiguous int32 [halves, T_total, maxc, 2]")
    check(_lib.load().mv_window_units_reduce(units.data_ptr(), units.stride(0), table.data_ptr(), table.shape[2], c, t_total, hw, halves,
                                             eps_acc.data_ptr(), _stream()), "mv_window_units_reduce")


def cfg_ddim_step(latents: torch.Tensor, eps_acc: torch.Tensor, counter: torch.Tensor, guidance: float, alpha_t: float,
                  alpha_prev: float) -> None:
    halves, c, t_total, hw = eps_acc.shape
    check(_lib.load().mv_cfg_ddim_step(latents.data_ptr(), eps_acc.data_ptr(), counter.data_ptr(), c, t_total, hw, halves,
                                       float(guidance), float(alpha_t), float(alpha_prev), _stream()), "mv_cfg_ddim_step")


def cfg_affine_step(latents: torch.Tensor, eps_acc: torch.Tensor, counter: torch.Tensor, guidance: float, cx: float,
                    ce: float) -> None:
    """latents <- cx * latents + ce * CFG(eps_acc / counter), in place (Euler-discrete: cx = 1, ce = sigma_next - sigma)."""
    halves, c, t_total, hw = eps_acc.shape
    check(_lib.load().mv_cfg_affine_step(latents.data_ptr(), eps_acc.data_ptr(), counter.data_ptr(), c, t_total, hw, halves,
                                         float(guidance), float(cx), float(ce), _stream()), "mv_cfg_affine_step")


def pack_conv_weight(w: torch.Tensor) -> torch.Tensor:
    """torch conv weight [O, I, *k] (k = 3x3, 3x1x1 or 1x1) -> packed fp16 [O, taps*I] (tap-major, channel-minor)."""
    o, i = w.shape[0], w.shape[1]
    taps = 1
    for s in w.shape[2:]:
        taps *= s
    w = w.contiguous()
    if w.dtype not in (torch.float16, torch.float32):
        w = w.float()
    out = torch.empty((o, taps * i), dtype=torch.float16, device=w.device)
    check(_lib.load().mv_pack_conv_weight_f16(w.data_ptr(), int(w.dtype == torch.float32), out.data_ptr(), o, i, taps,
                                              _stream()), "mv_pack_conv_weight_f16")
    return out


def pack_geglu(w: torch.Tensor, bias: Optional[torch.Tensor]):
    """Reorder a GEGLU projection (rows [value(4C) | gate(4C)]) into blocks of [16 value | 16 gate] rows so the
    gate can be applied in the GEMM epilogue (pure index shuffle; done once at weight-pack time)."""
    n2 = w.shape[0]
    half = n2 // 2
    if half % 16 != 0:
        raise ValueError("pack_geglu: 4C must be a multiple of 16")
    idx = torch.arange(half, device=w.device).view(-1, 16)
    perm = torch.cat([idx, idx + half], dim=1).reshape(-1)
    wp = w.index_select(0, perm).contiguous()
    bp = bias.index_select(0, perm).contiguous() if bias is not None else None
    return wp, bp


def probe_tr16(image: torch.Tensor) -> torch.Tensor:
    out = torch.empty((64, 4), dtype=torch.int16, device=image.device)
    check(_lib.load().mv_probe_tr16(image.data_ptr(), out.data_ptr(), _stream()), "mv_probe_tr16")
    return out


# every MUSEV_* variable the package, bench.py and tools/ read; anything else with the prefix is a typo or a knob of an earlier round
KNOWN_ENV = ("MUSEV_OPS", "MUSEV_GEMM_CFG", "MUSEV_GEMM_SPLITK", "MUSEV_NO_GRAPH", "MUSEV_HALF_STREAMS", "MUSEV_HIP_LIBRARY",
             "MUSEV_GOLDEN_AT_SIZE", "MUSEV_QUICK_AT_SIZE", "MUSEV_SIM_FULL", "MUSEV_SIM_MODEL")


def _apply_env_overrides() -> None:
    """MUSEV_OPS="NAME=VALUE,...": the one environment hook for A/B runs of this module's switches (ints; booleans as 0 / 1)"""
    unknown = sorted(k for k in os.environ if k.startswith("MUSEV_") and k not in KNOWN_ENV)
    if unknown:
        raise ValueError(f"unknown MUSEV_* environment variable(s) {unknown}: module switches are set through MUSEV_OPS=\"NAME=VALUE,...\" "
                         f"(known variables: {', '.join(KNOWN_ENV)})")
    spec = os.environ.get("MUSEV_OPS", "")
    for item in filter(None, (x.strip() for x in spec.split(","))):
        name, _, val = item.partition("=")
        if name not in ("COLSTATS", "CARRY", "CARRY_MAX_C", "FFN_FUSED", "FFN_ROTATE", "LN_FOLD", "ATTN_GROUPS", "XATTN_RESIDENT",
                        "GEMM_WEIGHT_STATIONARY", "TSA_FUSED", "LN_FOLD_MAX_K", "GN_FOLD", "GN_FOLD_MAX_RATIO", "XAB_FUSED", "XATTN_RESIDENT_MAX_D"):
            raise ValueError(f"MUSEV_OPS: unknown switch {name!r}")
        cur = globals()[name]
        globals()[name] = bool(int(val)) if isinstance(cur, bool) else float(val) if isinstance(cur, float) else int(val)


_apply_env_overrides()
