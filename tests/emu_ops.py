"""CPU emulation of the tensor-level C-ABI wrappers (musev_amd.ops) in plain torch: fp16 operands, fp32 arithmetic, fp16
results -- the storage contract of the kernels in include/musev_hip.h.  TEST INFRASTRUCTURE ONLY: it lets the CPU suite
drive the *host side* of the product (module wiring, weight packing, segment descriptors, geometry, conditioning rows,
the denoise loop) end to end against the oracle when no GPU is present.  The product never imports this module, and
nothing here is timed or shipped; on a GPU box the same module code runs on libmusev_hip.so.

tests/test_emulated_wiring.py first checks every function here against the very torch reference expressions that the
GPU kernels are verified against (tests/kernel_cases.py run with DEV = "cpu"), so "emulation == references" on the CPU
and "kernels == references" on the MI355X describe the same contract."""
from __future__ import annotations

import math
from typing import Optional

import torch
import torch.nn.functional as F

from fake_ops import cfg_affine_step, cfg_ddim_step, window_gather, window_scatter_add, window_units_reduce  # noqa: F401  (loop glue)

MV_ACT_NONE, MV_ACT_SILU = 0, 1

# The argument contract of the real entry points (musev_amd/ops.py::_mat/_vec and the MV_REQUIRE lines of csrc/*.hip) is
# enforced here too, so a module that hands a kernel a non-contiguous view, a misaligned column slice or an unsupported
# width fails on the CPU exactly where it would fail on the GPU.  STRICT_WIDTHS covers the constraints the 1/5-width
# golden cases cannot meet (head dim in {40, 80, 160}; conv sources in multiples of 64 channels).
STRICT_WIDTHS = True


class ContractError(ValueError):
    pass


def _req(cond, msg):
    if not cond:
        raise ContractError(msg)


def _mat(t, name, ld_mult=8):
    _req(t.dim() == 2 and t.stride(1) == 1, f"{name}: 2-D with unit inner stride expected, got {tuple(t.shape)} / {t.stride()}")
    _req(t.dtype == torch.float16, f"{name}: fp16 expected, got {t.dtype}")
    _req(t.stride(0) % ld_mult == 0 or t.shape[0] == 1, f"{name}: leading dimension {t.stride(0)} % {ld_mult}")
    _req(t.storage_offset() % 8 == 0, f"{name}: view starts at element {t.storage_offset()} (16-byte alignment)")
    return t


def _vec(t, name, n=None):
    if t is None:
        return None
    _req(t.dtype == torch.float16 and t.is_contiguous(), f"{name}: contiguous fp16 expected")
    _req(n is None or t.numel() == n, f"{name}: expected {n} elements, got {t.numel()}")
    return t


def _check_epilogue(M, N, cols, bias, rowbias, rows_per_group, residual, alpha):
    _vec(bias, "bias", N)
    if rowbias is not None:
        _mat(rowbias, "rowbias", 4)
        _req(rows_per_group > 0 and rowbias.shape[0] * rows_per_group >= M and rowbias.shape[1] == N, "rowbias: coverage")
    if residual is not None:
        _mat(residual, "residual", 4)
        _req(tuple(residual.shape) == (M, cols), "residual: shape mismatch")
    if alpha is not None:
        _req(alpha.dtype == torch.float32 and alpha.numel() == 1, "alpha: fp32 scalar tensor expected")


def _check_out(out, M, cols, ld_mult=4):
    if out is not None:
        _mat(out, "out", ld_mult)
        _req(tuple(out.shape) == (M, cols), f"out: expected {(M, cols)}, got {tuple(out.shape)}")


def _store(val: torch.Tensor, out: Optional[torch.Tensor], dtype=torch.float16) -> torch.Tensor:
    val = val.to(dtype)
    if out is None:
        return val.contiguous()
    out.copy_(val)
    return out


CARRY_MAX_C = 320  # musev_amd.ops.CARRY_MAX_C: widths of the residual stream that carry a lo half


def _carry_store(s32: torch.Tensor, out):
    """the carry epilogue: hi = fp16(s), lo = fp16(s - hi); lo rides on the hi tensor object (ops._carry_setup)"""
    hi = _store(s32, out)
    hi._mv_lo = (s32 - hi.float()).to(torch.float16)
    hi._mv_lo_version = hi._version
    return hi


def _use_carry(carry, cols, residual):
    return bool(carry) and cols <= CARRY_MAX_C and cols % 8 == 0


def _res_lo(residual):
    lo = getattr(residual, "_mv_lo", None) if residual is not None else None
    return lo if (lo is not None and lo.shape == residual.shape) else None


def _epilogue(acc, *, bias, rowbias, rows_per_group, residual, alpha, act, geglu, residual_lo=None):
    M = acc.shape[0]
    if bias is not None:
        acc = acc + bias.float()
    if rowbias is not None:
        acc = acc + rowbias.float()[torch.arange(M) // int(rows_per_group)]
    if alpha is not None:
        acc = acc * alpha.float().abs().reshape(())
    if act == MV_ACT_SILU:
        acc = F.silu(acc)
    if geglu:  # rows of the packed weight come in blocks of [16 value | 16 gate] (ops.pack_geglu)
        blk = acc.reshape(M, -1, 2, 16)
        acc = (blk[:, :, 0] * F.gelu(blk[:, :, 1])).reshape(M, -1)
    if residual is not None:
        acc = acc + residual.float()
        if residual_lo is not None:
            acc = acc + residual_lo.float()
    return acc


def ln_fold_applies(M, N, K, geglu):
    """the emulation folds wherever the kernel could (K % 64 == 0): the wiring of the folded form is what these tests exercise"""
    return K % 64 == 0


def fold_layernorm(w, bias, gamma, beta):
    wf = (w.float() * gamma.float().reshape(1, -1)).to(torch.float16).contiguous()
    colbias = w.float() @ beta.float().reshape(-1)
    if bias is not None:
        colbias = colbias + bias.float()
    return wf, wf.float().sum(dim=1).contiguous(), colbias.contiguous()


def gemm(a, w, *, a2=None, bias=None, rowbias=None, rows_per_group=0, residual=None, alpha=None, act=MV_ACT_NONE,
         geglu=False, out=None, ln=None, colstats=False, carry=False):
    _mat(a, "a")
    _mat(w, "w")
    _req(w.is_contiguous(), "w must be contiguous [N, K]")
    M, (N, K) = a.shape[0], w.shape
    if a2 is not None:
        _mat(a2, "a2")
        _req(a2.shape[0] == M, "a2: row count mismatch")
        _req(a.shape[1] % 64 == 0, "two-source input needs c1 % 64 == 0")
    x = a if a2 is None else torch.cat([a, a2], dim=1)
    _req(x.shape[1] == K, "gemm: K mismatch")
    _req(N % 4 == 0 and K % 8 == 0 and a.shape[1] % 8 == 0 and (a2 is None or a2.shape[1] % 8 == 0), "gemm: N % 4, K % 8, c % 8")
    if geglu:
        _req(N % 32 == 0 and rowbias is None and residual is None and alpha is None and act == MV_ACT_NONE, "geglu epilogue: bare only")
    cols = N // 2 if geglu else N
    _check_epilogue(M, N, cols, bias, rowbias, rows_per_group, residual, alpha)
    _check_out(out, M, cols)
    acc = x.float() @ w.float().t()
    if ln is not None:
        # the kernel's arithmetic: statistics of the raw fp16 rows in fp32, affine on the fp32 accumulator
        cs, cb, eps = ln
        _req(a2 is None and bias is None and rowbias is None and K % 64 == 0, "gemm(ln=): one source, K % 64 == 0, bias folded into colbias")
        _req(cs.dtype == torch.float32 and cb.dtype == torch.float32 and cs.numel() == N and cb.numel() == N, "gemm(ln=): fp32 [N] column vectors")
        xf = x.float()
        mean = xf.mean(dim=1, keepdim=True)
        var = (xf * xf).mean(dim=1, keepdim=True) - mean * mean
        rstd = torch.rsqrt(var.clamp_min(0.0) + eps)
        acc = rstd * (acc - mean * cs.reshape(1, -1)) + cb.reshape(1, -1)
    if _use_carry(carry, cols, residual) and not geglu and ln is None:
        return _carry_store(_epilogue(acc, bias=bias, rowbias=rowbias, rows_per_group=rows_per_group, residual=residual, alpha=alpha,
                                      act=act, geglu=False, residual_lo=_res_lo(residual)), out)
    return _store(_epilogue(acc, bias=bias, rowbias=rowbias, rows_per_group=rows_per_group, residual=residual,
                            alpha=alpha, act=act, geglu=geglu), out)


def pack_conv_weight(w):
    o, i = w.shape[0], w.shape[1]
    return w.reshape(o, i, -1).permute(0, 2, 1).reshape(o, -1).to(torch.float16).contiguous()  # tap-major, channel-minor


def _unpack(w, cin, kshape):
    o = w.shape[0]
    taps = w.shape[1] // cin
    return w.float().reshape(o, taps, cin).permute(0, 2, 1).reshape(o, cin, *kshape)


def _images(x, n_img, h, w_):
    return x.float().reshape(n_img, h, w_, -1).permute(0, 3, 1, 2)


def _rows(y):
    return y.permute(0, 2, 3, 1).reshape(-1, y.shape[1])


def conv3x3(x, w, n_img, h, w_, *, x2=None, stride=1, upsample=False, bias=None, rowbias=None, rows_per_group=0,
            residual=None, out=None, carry=False):
    _mat(x, "x")
    _mat(w, "w")
    _req(x.shape[0] == n_img * h * w_, "conv3x3: x rows != n_img*h*w")
    if x2 is not None:
        _mat(x2, "x2")
        _req(x2.shape[0] == x.shape[0], "conv3x3: x2 rows")
    xs = x if x2 is None else torch.cat([x, x2], dim=1)
    _req(w.shape[1] == 9 * xs.shape[1], "conv3x3: weight K != 9*(C1+C2)")
    _req(stride in (1, 2) and not (upsample and stride != 1), "conv3x3: stride / upsample")
    if STRICT_WIDTHS:
        _req(xs.shape[1] % 64 == 0 and (x2 is None or x.shape[1] % 64 == 0), "conv modes need cin % 64 == 0")
    ho, wo = (2 * h, 2 * w_) if upsample else ((h + 2 - 3) // stride + 1, (w_ + 2 - 3) // stride + 1)
    _check_epilogue(n_img * ho * wo, w.shape[0], w.shape[0], bias, rowbias, rows_per_group, residual, None)
    _check_out(out, n_img * ho * wo, w.shape[0])
    img = _images(xs, n_img, h, w_)
    if upsample:
        img = F.interpolate(img, scale_factor=2.0, mode="nearest")
    y = _rows(F.conv2d(img, _unpack(w, xs.shape[1], (3, 3)), None, stride=stride, padding=1))
    if _use_carry(carry, w.shape[0], residual):
        return _carry_store(_epilogue(y, bias=bias, rowbias=rowbias, rows_per_group=rows_per_group, residual=residual, alpha=None,
                                      act=MV_ACT_NONE, geglu=False, residual_lo=_res_lo(residual)), out)
    return _store(_epilogue(y, bias=bias, rowbias=rowbias, rows_per_group=rows_per_group, residual=residual, alpha=None,
                            act=MV_ACT_NONE, geglu=False), out)


def tconv3(x, w, b, t, hw, *, bias=None, residual=None, alpha=None, out=None, carry=False):
    _mat(x, "x")
    _mat(w, "w")
    c = x.shape[1]
    _req(x.shape[0] == b * t * hw and w.shape[1] == 3 * c, "tconv3: geometry")
    if STRICT_WIDTHS:
        _req(c % 64 == 0, "conv modes need cin % 64 == 0")
    _check_epilogue(x.shape[0], w.shape[0], w.shape[0], bias, None, 0, residual, alpha)
    _check_out(out, x.shape[0], w.shape[0])
    vol = x.float().reshape(b, t, hw, 1, c).permute(0, 4, 1, 2, 3)
    y = F.conv3d(vol, _unpack(w, c, (3, 1, 1)), None, padding=(1, 0, 0))
    y = y.permute(0, 2, 3, 4, 1).reshape(b * t * hw, -1)
    if _use_carry(carry, w.shape[0], residual):
        return _carry_store(_epilogue(y, bias=bias, rowbias=None, rows_per_group=0, residual=residual, alpha=alpha,
                                      act=MV_ACT_NONE, geglu=False, residual_lo=_res_lo(residual)), out)
    return _store(_epilogue(y, bias=bias, rowbias=None, rows_per_group=0, residual=residual, alpha=alpha,
                            act=MV_ACT_NONE, geglu=False), out)


def groupnorm(x, gamma, beta, n_items, rows, *, eps, silu, x2=None, groups=32, out=None, carry=False):
    _mat(x, "x")
    _req(x.shape[0] == n_items * rows, "groupnorm: x rows != n_items*rows")
    if x2 is not None:
        _mat(x2, "x2")
        _req(x2.shape[0] == x.shape[0] and x2.shape[1] % 8 == 0, "groupnorm: x2")
    xs = x if x2 is None else torch.cat([x, x2], dim=1)
    c = xs.shape[1]
    _req(x.shape[1] % 8 == 0 and c % groups == 0 and c <= 8192, "groupnorm: channels")
    _vec(gamma, "gamma", c)
    _vec(beta, "beta", c)
    _check_out(out, n_items * rows, c, 8)
    xf = xs.float()
    use_carry = bool(carry) and x2 is None and c <= CARRY_MAX_C
    if use_carry and getattr(x, "_mv_lo", None) is not None:
        # the kernel takes the statistics of the hi half and normalises hi + lo
        st = xf.reshape(n_items, rows, groups, c // groups)
        mean = st.mean(dim=(1, 3), keepdim=True)
        var = st.var(dim=(1, 3), unbiased=False, keepdim=True)
        y = ((xf + x._mv_lo.float()).reshape(n_items, rows, groups, c // groups) - mean) * torch.rsqrt(var + eps)
        y = y.reshape(n_items, rows, c) * gamma.float() + beta.float()
        y = y.permute(0, 2, 1)
    else:
        y = F.group_norm(xf.reshape(n_items, rows, c).permute(0, 2, 1), groups, gamma.float(), beta.float(), eps)
    if silu:
        y = F.silu(y)
    y = y.permute(0, 2, 1).reshape(n_items * rows, c)
    return _carry_store(y, out) if use_carry else _store(y, out)


def layernorm(x, gamma, beta, eps=1e-5, out=None):
    _mat(x, "x")
    _req(x.shape[1] % 8 == 0 and x.shape[1] <= 1536, "layernorm: C % 8 == 0 and C <= 1536")
    _vec(gamma, "gamma", x.shape[1])
    _vec(beta, "beta", x.shape[1])
    _check_out(out, x.shape[0], x.shape[1], 8)
    return _store(F.layer_norm(x.float(), (x.shape[1],), gamma.float(), beta.float(), eps), out)


def attention(q, segs, nb, lq, heads, d, scale, *, out=None, accumulate=False, out_scale=1.0, group_scales=None):
    c = heads * d
    _mat(q, "q")
    _req(tuple(q.shape) == (nb * lq, c) and 1 <= len(segs) <= 4, "attention: q shape / segment count")
    _req(nb <= 65535 and nb > 0 and lq > 0, "attention: grid")
    if STRICT_WIDTHS:
        _req(d in (40, 80, 160), f"attention: head dim {d} not in (40, 80, 160)")
    _check_out(out, nb * lq, c)
    _req(not accumulate or out is not None, "attention: accumulate needs out")
    ks, vs = [], []
    for k, v, ln, div, mul, add_ in segs:
        _mat(k, "k")
        _mat(v, "v")
        _req(ln > 0 and div > 0 and k.shape[1] == c and v.shape[1] == c, "attention: bad segment")
        kvb = [(n // div) * mul + add_ for n in range(nb)]
        _req(k.shape[0] >= (max(kvb) + 1) * ln and v.shape[0] >= (max(kvb) + 1) * ln, "attention: segment does not cover its key batches")
        nkb = k.shape[0] // ln
        ks.append(k.float()[: nkb * ln].reshape(nkb, ln, c)[kvb])
        vs.append(v.float()[: nkb * ln].reshape(nkb, ln, c)[kvb])
    qh = q.float().reshape(nb, lq, heads, d).transpose(1, 2)
    if group_scales is None:
        group_scales = [1.0] + [None] * (len(segs) - 1)
    _req(len(group_scales) == len(segs) and group_scales[0] is not None, "attention: group_scales: one entry per segment, the first a weight")
    _req(all(g is None for g in group_scales[1:]) or d in (40, 80) or not STRICT_WIDTHS, "attention: softmax groups need head dim 40 / 80")
    starts = [i for i, g in enumerate(group_scales) if g is not None] + [len(segs)]
    o = 0.0
    for a_, b_ in zip(starts[:-1], starts[1:]):
        kk, vv = torch.cat(ks[a_:b_], dim=1), torch.cat(vs[a_:b_], dim=1)
        kh = kk.reshape(nb, -1, heads, d).transpose(1, 2)
        vh = vv.reshape(nb, -1, heads, d).transpose(1, 2)
        o = o + float(group_scales[a_]) * (torch.softmax((qh @ kh.transpose(-1, -2)) * scale, dim=-1) @ vh).transpose(1, 2).reshape(nb * lq, c)
    if accumulate:
        o = out.float() + out_scale * o
    return _store(o, out)


def temporal_attention(q, k, v, b, t, hw, heads, d, scale, out=None):
    c = heads * d
    for name, m in (("q", q), ("k", k), ("v", v)):
        _mat(m, name)
        _req(tuple(m.shape) == (b * t * hw, c), f"temporal_attention: {name} shape")
    _req(1 <= t <= 32 and d % 8 == 0, "temporal_attention: T in [1, 32], d % 8")
    _check_out(out, b * t * hw, c, 8)

    def seq(x):  # rows (b, t, p) -> [(b p), heads, t, d]
        return x.float().reshape(b, t, hw, heads, d).permute(0, 2, 3, 1, 4).reshape(b * hw, heads, t, d)
    o = torch.softmax((seq(q) @ seq(k).transpose(-1, -2)) * scale, dim=-1) @ seq(v)
    o = o.reshape(b, hw, heads, t, d).permute(0, 3, 1, 2, 4).reshape(b * t * hw, c)
    return _store(o, out)


def geglu(x, out=None):
    _mat(x, "x")
    half = x.shape[1] // 2
    _req(half % 8 == 0, "geglu: half_cols % 8")
    _check_out(out, x.shape[0], half, 8)
    return _store(x.float()[:, :half] * F.gelu(x.float()[:, half:]), out)


def softmax_rows_(x):
    _mat(x, "x")
    _req(x.shape[1] % 8 == 0 and x.shape[1] <= 16384 and x.stride(0) % 8 == 0, "softmax_rows_: cols % 8, cols <= 16384, ld % 8")
    x.copy_(torch.softmax(x.float(), dim=-1).to(torch.float16))
    return x


def silu(x):
    _req(x.is_contiguous() and x.dtype == torch.float16 and x.numel() % 8 == 0 and x.numel() > 0, "silu: contiguous fp16, n % 8")
    return F.silu(x.float()).to(torch.float16)


def add(a, b):
    _req(a.shape == b.shape and a.is_contiguous() and b.is_contiguous() and a.dtype == b.dtype == torch.float16
         and a.numel() % 8 == 0, "add: contiguous fp16 tensors of equal shape, n % 8")
    lo = _res_lo(a)
    if lo is not None:   # a carried stream tensor keeps both halves through the add (mv_add_f16 with a_lo / y_lo)
        return _carry_store((a.float() + lo.float()) + b.float(), None)
    return (a.float() + b.float()).to(torch.float16)


def ffn_fused_applies(c, hidden):
    return c == 320 and hidden == 1280


def ffn_geglu(x, gamma, beta, eps, w1p, b1p, w2, b2, residual, out=None):
    """the kernel's arithmetic: LayerNorm rounded to fp16, fp32 accumulation, the gated activation rounded to fp16, fp32 residual add"""
    _mat(x, "x")
    _mat(residual, "residual")
    M, c = x.shape
    hidden = w2.shape[1]
    _req(c == 320 and hidden == 1280 and tuple(w1p.shape) == (2 * hidden, c) and tuple(w2.shape) == (c, hidden), "ffn_geglu: C = 320, hidden = 1280 only")
    _req(tuple(residual.shape) == (M, c) and w1p.is_contiguous() and w2.is_contiguous(), "ffn_geglu: shapes")
    _vec(gamma, "gamma", c)
    _vec(beta, "beta", c)
    _vec(b1p, "bias1", 2 * hidden)
    _vec(b2, "bias2", c)
    _check_out(out, M, c, 8)
    xn = F.layer_norm(x.float(), (c,), gamma.float(), beta.float(), eps).to(torch.float16)
    acc = xn.float() @ w1p.float().t()
    if b1p is not None:
        acc = acc + b1p.float()
    blk = acc.reshape(M, -1, 2, 16)
    gact = (blk[:, :, 0] * F.gelu(blk[:, :, 1])).reshape(M, -1).to(torch.float16)
    y = gact.float() @ w2.float().t()
    if b2 is not None:
        y = y + b2.float()
    return _store(y + residual.float(), out)


def tsa_fused_applies(c, heads, d, t, hw):
    return c == 320 and heads == 8 and d == 40 and 1 <= t <= 16 and hw % 8 == 0


def pack_tsa_qkv(wq, wk, wv, heads, d):
    c = wq.shape[1]
    out = torch.zeros(heads, 128, c, dtype=torch.float16)
    for i, w in enumerate((wq, wk, wv)):
        out[:, i * d:(i + 1) * d] = w.reshape(heads, d, c).to(torch.float16)
    return out.contiguous()


def pack_tsa_out(wo, heads, d):
    c = wo.shape[0]
    out = torch.zeros(c, heads, 64, dtype=torch.float16)
    out[:, :, :d] = wo.reshape(c, heads, d).to(torch.float16)
    return out.reshape(c, heads * 64).contiguous()


def temporal_attn_block(x, gamma, beta, eps, wqkv_p, wo_p, bias_o, b, t, hw, heads, d, scale, out=None):
    """the kernel's arithmetic: LayerNorm rounded to fp16, q / k / v rounded to fp16, probabilities normalised then rounded to fp16,
    the attention output rounded to fp16, fp32 accumulation and an fp32 residual add, rounded once"""
    _mat(x, "x")
    M, c = x.shape
    _req(c == 320 and heads == 8 and d == 40 and 1 <= t <= 16 and hw % 8 == 0 and M == b * t * hw, "temporal_attn_block: C = 320 = 8 x 40, T <= 16, HW % 8")
    _req(tuple(wqkv_p.shape) == (heads, 128, c) and tuple(wo_p.shape) == (c, heads * 64) and wqkv_p.is_contiguous() and wo_p.is_contiguous(), "temporal_attn_block: packed weights")
    _vec(gamma, "gamma", c)
    _vec(beta, "beta", c)
    _vec(bias_o, "bias_o", c)
    _check_out(out, M, c, 8)
    xn = F.layer_norm(x.float(), (c,), gamma.float(), beta.float(), eps).to(torch.float16).float()
    qkv = (xn @ wqkv_p.float().reshape(heads * 128, c).t()).to(torch.float16).float().reshape(M, heads, 128)

    def seq(y):  # [M, heads, d] rows (b, t, p) -> [(b p), heads, t, d]
        return y.reshape(b, t, hw, heads, d).permute(0, 2, 3, 1, 4).reshape(b * hw, heads, t, d)
    q, k, v = seq(qkv[:, :, :d]), seq(qkv[:, :, d:2 * d]), seq(qkv[:, :, 2 * d:3 * d])
    p = torch.softmax((q @ k.transpose(-1, -2)) * scale, dim=-1).to(torch.float16).float()
    o = (p @ v).to(torch.float16).float()                                             # [(b p), heads, t, d]
    o = o.reshape(b, hw, heads, t, d).permute(0, 3, 1, 2, 4).reshape(M, heads, d)
    y = torch.einsum("mhd,chd->mc", o, wo_p.float().reshape(c, heads, 64)[:, :, :d])
    if bias_o is not None:
        y = y + bias_o.float()
    return _store(y + x.float(), out)



def xab_fused_applies(c, heads, d, n_keys, rows_per_kvb):
    return c == 320 and heads == 8 and d == 40 and 1 <= n_keys <= 80 and rows_per_kvb % 128 == 0


def pack_xab_q(wq, heads, d):
    c = wq.shape[1]
    out = torch.zeros(heads // 2, 2, 64, c, dtype=torch.float16)
    out[:, :, :d] = wq.reshape(heads // 2, 2, d, c).to(torch.float16)
    return out.reshape(heads // 2, 128, c).contiguous()


def xattn_block(x, gamma, beta, eps, wq_p, k, v, n_keys, rows_per_kvb, wo_p, bias_o, heads, d, scale, out=None):
    """the kernel's arithmetic (mv_xattn_block_f16): LayerNorm rounded to fp16, q rounded to fp16, probabilities normalised then rounded to
    fp16, the attention output rounded to fp16, fp32 accumulation and an fp32 residual add, rounded once"""
    _mat(x, "x")
    _mat(k, "k")
    _mat(v, "v")
    M, c = x.shape
    _req(xab_fused_applies(c, heads, d, n_keys, rows_per_kvb), "xattn_block: C = 320 = 8 x 40, <= 80 keys, rows_per_kvb % 128")
    nkvb = (M + rows_per_kvb - 1) // rows_per_kvb
    _req(tuple(wq_p.shape) == (heads // 2, 128, c) and tuple(wo_p.shape) == (c, heads * 64) and wq_p.is_contiguous() and wo_p.is_contiguous(), "xattn_block: packed weights")
    _req(k.shape[0] >= nkvb * n_keys and v.shape[0] >= nkvb * n_keys and k.shape[1] == c and v.shape[1] == c, "xattn_block: keys / values")
    _vec(gamma, "gamma", c)
    _vec(beta, "beta", c)
    _vec(bias_o, "bias_o", c)
    _check_out(out, M, c, 8)
    xn = F.layer_norm(x.float(), (c,), gamma.float(), beta.float(), eps).to(torch.float16).float()
    wq = wq_p.float().reshape(heads // 2, 2, 64, c)[:, :, :d].reshape(heads * d, c)
    q = (xn @ wq.t()).to(torch.float16).float().reshape(M, heads, d)
    kb = torch.arange(M) // rows_per_kvb
    kk = k.float()[:nkvb * n_keys].reshape(nkvb, n_keys, heads, d)[kb]      # [M, keys, heads, d]
    vv = v.float()[:nkvb * n_keys].reshape(nkvb, n_keys, heads, d)[kb]
    s = torch.einsum("mhd,mkhd->mhk", q, kk) * scale
    p = torch.softmax(s, dim=-1).to(torch.float16).float()
    o = torch.einsum("mhk,mkhd->mhd", p, vv).to(torch.float16).float()
    y = torch.einsum("mhd,chd->mc", o, wo_p.float().reshape(c, heads, 64)[:, :, :d])
    if bias_o is not None:
        y = y + bias_o.float()
    return _store(y + x.float(), out)


def conv3x3_cin_small(x, w, bias, n_img, h, w_, add_=None, _carry=False):
    cin = x.shape[1]
    _req(x.dim() == 2 and x.is_contiguous() and x.dtype == torch.float16 and x.shape[0] == n_img * h * w_, "conv_in: x")
    _req(w.dtype == torch.float16 and w.is_contiguous() and w.shape[1] >= 9 * cin and w.shape[0] % 8 == 0 and cin <= 16, "conv_in: w")
    _vec(bias, "bias", w.shape[0])
    if add_ is not None:
        _req(add_.dtype == torch.float16 and add_.is_contiguous() and tuple(add_.shape) == (x.shape[0], w.shape[0]), "conv_in: add")
    y = _rows(F.conv2d(_images(x, n_img, h, w_), _unpack(w[:, : 9 * cin], cin, (3, 3)), None, padding=1))
    if bias is not None:
        y = y + bias.float()
    if add_ is not None:
        y = y + add_.float()
    if _carry and _use_carry(True, w.shape[0], None):
        return _carry_store(y, None)
    return y.to(torch.float16)


def conv3x3_cin_small_gemm(x, w, bias, n_img, h, w_, add_=None, kpad=64):
    _req(9 * x.shape[1] <= kpad and kpad % 8 == 0 and w.shape[1] in (9 * x.shape[1], kpad), "conv3x3_cin_small_gemm: bad shapes")
    return conv3x3_cin_small(x, w, bias, n_img, h, w_, add_, _carry=True)  # (the GEMM form opens the residual stream with a carry)


def pad_cols(w, k):
    out = torch.zeros((w.shape[0], k), dtype=w.dtype)
    out[:, : w.shape[1]] = w
    return out


def conv3x3_cout_small(x, w, bias, n_img, h, w_, out_dtype=torch.float16):
    _mat(x, "x")
    _mat(w, "w")
    _req(x.is_contiguous() and w.shape[1] == 9 * x.shape[1] and x.shape[1] % 8 == 0 and w.shape[0] <= 8, "conv_out: shapes")
    _req(x.shape[0] == n_img * h * w_ and out_dtype in (torch.float16, torch.float32), "conv_out: rows / dtype")
    _vec(bias, "bias", w.shape[0])
    xin = x.float() + x._mv_lo.float() if getattr(x, "_mv_lo", None) is not None else x
    y = _rows(F.conv2d(_images(xin, n_img, h, w_), _unpack(w, x.shape[1], (3, 3)), None, padding=1))
    if bias is not None:
        y = y + bias.float()
    return y.to(out_dtype).contiguous()


def conv3x3_direct(x, w, bias, n_img, h, w_, *, stride=1, act=MV_ACT_NONE):
    _req(x.dim() == 2 and x.is_contiguous() and x.dtype == torch.float16 and x.shape[0] == n_img * h * w_, "conv3x3_direct: x")
    _req(w.dim() == 2 and w.is_contiguous() and w.dtype == torch.float16 and w.shape[1] == 9 * x.shape[1], "conv3x3_direct: w")
    _req(w.shape[0] % 8 == 0 and x.shape[1] <= 455 and stride in (1, 2) and act in (MV_ACT_NONE, MV_ACT_SILU), "conv3x3_direct: args")
    _req(x.storage_offset() % 8 == 0, "conv3x3_direct: 16-byte alignment")
    _vec(bias, "bias", w.shape[0])
    y = _rows(F.conv2d(_images(x, n_img, h, w_), _unpack(w, x.shape[1], (3, 3)), None, stride=stride, padding=1))
    if bias is not None:
        y = y + bias.float()
    if act == MV_ACT_SILU:
        y = F.silu(y)
    return y.to(torch.float16).contiguous()


def timestep_embedding(t, dim):
    half = dim // 2
    freq = torch.exp(-math.log(10000.0) * torch.arange(half, dtype=torch.float32) / half)
    arg = t.float().reshape(-1, 1) * freq[None]
    return torch.cat([torch.cos(arg), torch.sin(arg)], dim=-1).to(torch.float16)  # flip_sin_to_cos, freq_shift 0


def upsample_nearest(x, n_img, h, w_, ho, wo):
    x = _mat(x, "x", 8)
    _req(x.shape[0] == n_img * h * w_ and x.shape[1] % 8 == 0, "upsample_nearest: rows / channels")
    y = F.interpolate(x.float().reshape(n_img, h, w_, -1).permute(0, 3, 1, 2), size=(ho, wo), mode="nearest")
    return y.permute(0, 2, 3, 1).reshape(n_img * ho * wo, -1).to(torch.float16).contiguous()


def zero_rows(x, row_idx):
    _mat(x, "x", 1)
    _req(row_idx.numel() > 0 and int(row_idx.max()) < x.shape[0], "zero_rows: index range")
    x[row_idx.long()] = 0


def bcthw_to_bthwc(x):
    return x.float().permute(0, 2, 3, 4, 1).reshape(-1, x.shape[1]).to(torch.float16).contiguous()


def bthwc_to_bcthw(x, b, t, h, w, dtype=torch.float16):
    _req(x.dim() == 2 and x.is_contiguous() and x.dtype in (torch.float16, torch.float32) and x.shape[0] == b * t * h * w,
         "bthwc_to_bcthw: contiguous 2-D fp16|fp32 rows of b*t*h*w expected")
    return x.reshape(b, t, h, w, -1).permute(0, 4, 1, 2, 3).to(dtype).contiguous()



GN_FOLD_MAX_RATIO = 0.35   # musev_amd.ops.GN_FOLD_MAX_RATIO


def groupnorm_fold_linear(x, gamma, beta, n_items, rows, *, eps, groups, w, bias, rowbias=None, rb_per_item=1, out=None):
    """the contract of ops.groupnorm_fold_linear (mv_groupnorm_cs_fold_linear_f16 + the per-group-weight projection): per item the fp16
    weights W gamma rstd, the bias term in fp32 from the ROUNDED weights; the emulation has no producer statistics, so the fold is taken
    wherever the size rule allows it"""
    _mat(x, "x")
    _mat(w, "w")
    M, c = x.shape
    N = w.shape[0]
    _req(M == n_items * rows and w.shape[1] == c, "groupnorm_fold_linear: shapes")
    if n_items * N > GN_FOLD_MAX_RATIO * M or rows % 32 or rows % rb_per_item or c % 8 or c > 2048 or c % groups:
        return None
    _vec(gamma, "gamma", c)
    _vec(beta, "beta", c)
    if rowbias is not None:
        _req(tuple(rowbias.shape) == (n_items * rb_per_item, N), "groupnorm_fold_linear: rowbias must be [n_items * rb_per_item, N]")
    xf = x.float().reshape(n_items, rows, groups, c // groups)
    mean = xf.mean(dim=(1, 3))                                   # [items, groups]
    rstd = torch.rsqrt(xf.var(dim=(1, 3), unbiased=False) + eps)
    cpg = c // groups
    mean_c, rstd_c = mean.repeat_interleave(cpg, dim=1), rstd.repeat_interleave(cpg, dim=1)   # [items, c]
    wf = (w.float()[None] * (gamma.float()[None] * rstd_c)[:, None, :]).half()                # [items, N, c]
    b = (w.float() @ beta.float())[None] - torch.einsum("inc,ic->in", wf.float(), mean_c)      # [items, N]
    if bias is not None:
        b = b + bias.float()[None]
    b = b.repeat_interleave(rb_per_item, dim=0)
    if rowbias is not None:
        b = b + rowbias.float()
    hi = b.half()
    lo = (b - hi.float()).half()
    y = torch.einsum("irc,inc->irn", x.float().reshape(n_items, rows, c), wf.float()).reshape(M, N)
    y = y + (hi.float() + lo.float())[torch.arange(M) // (rows // rb_per_item)]
    return _store(y, out)


def pack_geglu(w, bias):
    half = w.shape[0] // 2
    idx = torch.arange(half).view(-1, 16)
    perm = torch.cat([idx, idx + half], dim=1).reshape(-1)
    return w.index_select(0, perm).contiguous(), (bias.index_select(0, perm).contiguous() if bias is not None else None)


EMULATED = ["gemm", "ln_fold_applies", "fold_layernorm", "conv3x3", "tconv3", "groupnorm", "groupnorm_fold_linear", "layernorm", "attention", "temporal_attention", "geglu", "silu", "add", "softmax_rows_",
            "conv3x3_cin_small", "conv3x3_cin_small_gemm", "pad_cols", "conv3x3_cout_small", "conv3x3_direct", "timestep_embedding", "zero_rows", "upsample_nearest",
            "bcthw_to_bthwc", "bthwc_to_bcthw", "window_gather", "window_scatter_add", "window_units_reduce", "cfg_ddim_step", "cfg_affine_step",
            "pack_conv_weight", "pack_geglu", "ffn_fused_applies", "ffn_geglu", "tsa_fused_applies", "pack_tsa_qkv", "pack_tsa_out", "temporal_attn_block", "xab_fused_applies", "pack_xab_q", "xattn_block"]


def install(monkeypatch) -> None:
    """route musev_amd.ops.<name> to the emulation for the duration of one test (pytest's monkeypatch restores it)"""
    import sys

    from musev_amd import ops
    me = sys.modules[__name__]
    for name in EMULATED:
        monkeypatch.setattr(ops, name, getattr(me, name))
