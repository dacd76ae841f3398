"""Per-kernel parity cases: each HIP entry point (through the C ABI) against a plain PyTorch fp32 reference of the
same op on the same seeded inputs.  Used by tests/test_kernels_gpu.py (pytest -m gpu) and by tools/gpu_report.py / tools/gpu_gemm_ab.py
(one-shot report that does not stop at the first failure).

Tolerances: inputs/outputs are fp16, accumulation fp32 -> |err| <= atol + rtol*|ref| with rtol = 2^-9 (two fp16
roundings) and an atol scaled to the magnitude of the contraction (stated per case).
"""
from __future__ import annotations

import math
from typing import Callable, Dict, List, Tuple

import torch
import torch.nn.functional as F

DEV = "cuda"


def _rand(shape, seed, scale=1.0, dtype=torch.float16):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(shape, generator=g, dtype=torch.float32) * scale).to(dtype).to(DEV)


def _cmp(name, got, ref, atol, rtol=2e-3):
    got = got.float()
    ref = ref.float()
    err = (got - ref).abs()
    bound = atol + rtol * ref.abs()
    worst = (err - bound).max().item()
    return {
        "name": name,
        "max_abs_err": err.max().item(),
        "ref_absmax": ref.abs().max().item(),
        "atol": atol,
        "rtol": rtol,
        "ok": bool(worst <= 0.0) and bool(torch.isfinite(got).all()),
    }


# ---------------------------------------------------------------------------------------------------
def case_tr16_probe():
    """ds_read_b64_tr_b16 must return, to lane i of a 16-lane group, column i of the 4x16 block whose rows are
    addressed by lanes (4r .. 4r+3)."""
    from musev_amd import ops
    img = torch.arange(1024, dtype=torch.int16, device=DEV)
    out = ops.probe_tr16(img).cpu()
    exp = torch.empty((64, 4), dtype=torch.int16)
    for l in range(64):
        g, i = l >> 4, l & 15
        for j in range(4):
            exp[l, j] = (4 * g + j) * 64 + i  # image[row 4g + j][col i] with row stride 64
    ok = bool((out == exp).all())
    return {"name": "tr16_probe", "ok": ok, "max_abs_err": float((out.int() - exp.int()).abs().max()),
            "got_lane0": out[0].tolist(), "got_lane1": out[1].tolist(), "got_lane17": out[17].tolist(),
            "exp_lane17": exp[17].tolist()}


def case_gemm(M=1000, N=320, K=640, seed=0, two_src=False, epilogue=True):
    from musev_amd import ops
    a = _rand((M, K), seed)
    w = _rand((N, K), seed + 1, 1.0 / math.sqrt(K))
    bias = _rand((N,), seed + 2) if epilogue else None
    groups = 7
    rpg = (M + groups - 1) // groups
    rowbias = _rand((groups, N), seed + 3) if epilogue else None
    res = _rand((M, N), seed + 4) if epilogue else None
    alpha = torch.tensor([-0.37], dtype=torch.float32, device=DEV) if epilogue else None
    if two_src:
        c1 = 384 if K > 384 else K // 2
        got = ops.gemm(a[:, :c1], w, a2=a[:, c1:], bias=bias, rowbias=rowbias, rows_per_group=rpg, residual=res, alpha=alpha)
    else:
        got = ops.gemm(a, w, bias=bias, rowbias=rowbias, rows_per_group=rpg, residual=res, alpha=alpha)
    ref = a.float() @ w.float().t()
    if epilogue:
        ref = ref + bias.float()
        ref = ref + rowbias.float()[torch.arange(M, device=DEV) // rpg]
        ref = ref * 0.37 + res.float()
    return _cmp(f"gemm M{M} N{N} K{K} two_src={two_src} epi={epilogue}", got, ref, atol=4e-3)


def case_gemm_ln(M=1000, N=960, K=320, seed=20, geglu=False, residual=False, offset=0.0, cfg=None):
    """LayerNorm folded into the projection (ops.gemm(ln=)): LayerNorm(x) @ W.T + bias (nn.LayerNorm + Linear in fp32 as the
    reference computes them), optionally the GEGLU gate or a residual behind it; ``offset`` shifts the rows' mean (the folded form
    subtracts mean * colsum from the accumulator: cancellation is exercised with |mean| >> std)."""
    from musev_amd import ops
    x = _rand((M, K), seed) * 1.7 + offset
    gamma = 1.0 + 0.3 * _rand((K,), seed + 1)
    beta = 0.2 * _rand((K,), seed + 2)
    w = _rand((N, K), seed + 3, 1.0 / math.sqrt(K))
    b = _rand((N,), seed + 4, 0.3)
    res = _rand((M, N), seed + 5) if residual else None
    y = F.layer_norm(x.float(), (K,), gamma.float(), beta.float(), 1e-5)
    ref = y @ w.float().t() + b.float()
    if geglu:
        wp, bp = ops.pack_geglu(w, b)
        wf, cs, cb = ops.fold_layernorm(wp, bp, gamma, beta)
        h = ref
        ref = h[:, :N // 2] * F.gelu(h[:, N // 2:])
    else:
        wf, cs, cb = ops.fold_layernorm(w, b, gamma, beta)
    if res is not None:
        ref = ref.half().float() + res.float()
    old = ops.GEMM_CFG
    if cfg is not None:
        ops.GEMM_CFG = cfg
    try:
        got = ops.gemm(x, wf, ln=(cs, cb, 1e-5), geglu=geglu, residual=res)
    finally:
        ops.GEMM_CFG = old
    return _cmp(f"gemm LN-folded M{M} N{N} K{K} geglu={geglu} res={residual} offset={offset} cfg={cfg}", got, ref, atol=5e-3)


def case_gemm_ln_repeatable(M=53248, N=320, K=320, runs=12, seed=30):
    """the folded launch must be bit-reproducible: round 3 found hipcc's packed-fp32 form of the row affine (v_pk_mul_f32 / v_pk_fma_f32
    with op_sel swizzles) returning sporadically wrong LOW results on lanes 48-63 of the MI355X -- one element of one 16 x 16 tile in
    1 of ~5 launches at this size (profiles/r03d_ln_stress.log); the epilogue now uses scalar v_fma_f32 (common.h MV_FMA_SCALAR)"""
    from musev_amd import ops
    x = _rand((M, K), seed) * 1.3
    gamma = 1.0 + 0.1 * _rand((K,), seed + 1)
    beta = 0.1 * _rand((K,), seed + 2)
    w = _rand((N, K), seed + 3, 1.0 / math.sqrt(K))
    b = _rand((N,), seed + 4, 0.1)
    wf, cs, cb = ops.fold_layernorm(w, b, gamma, beta)
    outs = [ops.gemm(x, wf, ln=(cs, cb, 1e-5)) for _ in range(runs)]
    ref = F.layer_norm(x.float(), (K,), gamma.float(), beta.float(), 1e-5) @ w.float().t() + b.float()
    res = _cmp(f"gemm LN-folded x{runs} M{M} N{N} K{K}", outs[-1], ref, atol=5e-3)
    res["ok"] = res["ok"] and all(torch.equal(o, outs[0]) for o in outs)
    res["runs_equal"] = all(torch.equal(o, outs[0]) for o in outs)
    return res


def case_gemm_silu():
    from musev_amd import ops
    a = _rand((26, 320), 5)
    w = _rand((1280, 320), 6, 1.0 / math.sqrt(320))
    b = _rand((1280,), 7)
    got = ops.gemm(a, w, bias=b, act=ops.MV_ACT_SILU)
    ref = F.silu(a.float() @ w.float().t() + b.float())
    return _cmp("gemm silu (time-embedding MLP shape)", got, ref, atol=3e-3)


def case_gemm_geglu(M=777, C=320):
    from musev_amd import ops
    a = _rand((M, C), 8)
    w = _rand((8 * C, C), 9, 1.0 / math.sqrt(C))
    b = _rand((8 * C,), 10, 0.1)
    wp, bp = ops.pack_geglu(w, b)
    got = ops.gemm(a, wp, bias=bp, geglu=True)
    h = a.float() @ w.float().t() + b.float()
    ref = h[:, : 4 * C] * F.gelu(h[:, 4 * C:])
    return _cmp(f"gemm geglu M{M} C{C}", got, ref, atol=4e-3)


def case_conv3x3(n=3, h=16, w=24, c1=64, c2=0, cout=320, stride=1, upsample=False, seed=20):
    from musev_amd import ops
    cin = c1 + c2
    x = _rand((n, cin, h, w), seed)
    wt = _rand((cout, cin, 3, 3), seed + 1, 1.0 / math.sqrt(9 * cin))
    bias = _rand((cout,), seed + 2)
    xr = x.float()
    if upsample:
        xr = F.interpolate(xr, scale_factor=2.0, mode="nearest")
    ref = F.conv2d(xr, wt.float(), bias.float(), stride=stride, padding=1)
    ho, wo = ref.shape[2], ref.shape[3]
    temb = _rand((n, cout), seed + 3)
    res = _rand((n * ho * wo, cout), seed + 4)
    ref = ref + temb.float()[:, :, None, None]
    ref = ref.permute(0, 2, 3, 1).reshape(n * ho * wo, cout) + res.float()
    xl = x.permute(0, 2, 3, 1).reshape(n * h * w, cin).contiguous()
    wp = ops.pack_conv_weight(wt)
    if c2:
        got = ops.conv3x3(xl[:, :c1].contiguous(), wp, n, h, w, x2=xl[:, c1:].contiguous(), stride=stride, upsample=upsample,
                          bias=bias, rowbias=temb, rows_per_group=ho * wo, residual=res)
    else:
        got = ops.conv3x3(xl, wp, n, h, w, stride=stride, upsample=upsample, bias=bias, rowbias=temb,
                          rows_per_group=ho * wo, residual=res)
    return _cmp(f"conv3x3 n{n} {h}x{w} c{c1}+{c2}->{cout} s{stride} up{int(upsample)}", got, ref, atol=5e-3)


def case_tconv3(b=2, t=5, hw=48, c=128, seed=30):
    from musev_amd import ops
    x = _rand((b, c, t, hw, 1), seed)
    wt = _rand((c, c, 3, 1, 1), seed + 1, 1.0 / math.sqrt(3 * c))
    bias = _rand((c,), seed + 2)
    ref = F.conv3d(x.float(), wt.float(), bias.float(), padding=(1, 0, 0))
    alpha = torch.tensor([0.6], dtype=torch.float32, device=DEV)
    ref = x.float() + 0.6 * ref
    ref = ref.permute(0, 2, 3, 4, 1).reshape(b * t * hw, c)
    xl = x.permute(0, 2, 3, 4, 1).reshape(b * t * hw, c).contiguous()
    got = ops.tconv3(xl, ops.pack_conv_weight(wt), b, t, hw, bias=bias, residual=xl, alpha=alpha)
    return _cmp(f"tconv3 b{b} t{t} hw{hw} c{c}", got, ref, atol=4e-3)


def case_carry(kind="linear", n=3, h=16, w=16, cin=64, c=320, cfg=None, seed=600, with_colstats=True):
    """two-fp16 carry of the residual stream (mv_gemm_desc.c_lo / residual_lo): a chain of two stream-producing launches -- the first
    opens the stream (no residual), the second adds to it -- in each of the three modes.  Checked: hi against the fp32 sum to the
    usual fp16 bar; hi + lo against the fp32 sum to 2e-5 (1 + |ref|) -- 25 x tighter than an fp16 tensor can be (2^-11 |ref|): the
    second launch must have picked the first one's lo half up; lo is a rounding remainder (at most half an ulp of hi); the column
    statistics are those of hi."""
    from musev_amd import ops
    hw = h * w
    M = n * hw
    old = ops.GEMM_CFG
    if cfg is not None:
        ops.GEMM_CFG = cfg
    hits = ops.CARRY_HITS
    try:
        # launch 1 (opens the stream, values of magnitude ~12 like the level-0 stream of the noise-predictor weights): linear, bias only
        a0 = _rand((M, 64), seed)
        w0 = _rand((c, 64), seed + 1, 12.0 / 8.0)
        b0 = _rand((c,), seed + 2)
        s1 = ops.gemm(a0, w0, bias=b0, carry=True)
        ref1 = a0.float() @ w0.float().t() + b0.float()
        # launch 2: f(x) + stream
        if kind == "conv":
            x = _rand((M, cin), seed + 3)
            wt = _rand((c, cin, 3, 3), seed + 4, 1.0 / math.sqrt(9 * cin))
            bias = _rand((c,), seed + 5)
            temb = _rand((n, c), seed + 6)
            s2 = ops.conv3x3(x, ops.pack_conv_weight(wt), n, h, w, bias=bias, rowbias=temb, rows_per_group=hw, residual=s1, carry=True)
            f = F.conv2d(x.float().reshape(n, h, w, cin).permute(0, 3, 1, 2), wt.float(), bias.float(), padding=1) + temb.float()[:, :, None, None]
            f = f.permute(0, 2, 3, 1).reshape(M, c)
        elif kind == "tconv":
            x = _rand((M, c), seed + 3)  # (b = 1, t = n)
            wt = _rand((c, c, 3, 1, 1), seed + 4, 1.0 / math.sqrt(3 * c))
            bias = _rand((c,), seed + 5)
            s2 = ops.tconv3(x, ops.pack_conv_weight(wt), 1, n, hw, bias=bias, residual=s1,
                            alpha=torch.tensor([-0.7], dtype=torch.float32, device=DEV), carry=True)
            vol = x.float().reshape(1, n, hw, 1, c).permute(0, 4, 1, 2, 3)
            f = 0.7 * F.conv3d(vol, wt.float(), bias.float(), padding=(1, 0, 0)).permute(0, 2, 3, 4, 1).reshape(M, c)
        else:
            x = _rand((M, cin), seed + 3)
            wt = _rand((c, cin), seed + 4, 1.0 / math.sqrt(cin))
            bias = _rand((c,), seed + 5)
            s2 = ops.gemm(x, wt, bias=bias, residual=s1, colstats=with_colstats, carry=True)
            f = x.float() @ wt.float().t() + bias.float()
    finally:
        ops.GEMM_CFG = old
    name = f"carry {kind} cfg{cfg} M{M} c{c}"
    if ops.CARRY_HITS != hits + 2 or getattr(s1, "_mv_lo", None) is None or getattr(s2, "_mv_lo", None) is None:
        return {"name": name, "ok": False, "max_abs_err": float("nan"), "detail": "a launch did not take the carry path"}
    ref2 = f + ref1
    results = [_cmp(name + " hi1", s1, ref1, atol=4e-3), _cmp(name + " hi2", s2, ref2, atol=6e-3)]
    for nm, t, ref in (("1", s1, ref1), ("2", s2, ref2)):
        both = t.float() + t._mv_lo.float()
        results.append(_cmp(name + f" hi{nm} + lo{nm}", both, ref, atol=2e-5, rtol=2e-5))
        # lo is a rounding remainder: at most half an ulp of hi (|hi| 2^-11; its own rounding can land it exactly on the half)
        worst = (t._mv_lo.float().abs() - (t.float().abs() * 2.0 ** -11 + 2.0 ** -25)).max().item()
        results.append({"name": name + f" |lo{nm}| <= ulp(hi{nm}) / 2", "ok": worst <= 0.0, "max_abs_err": max(worst, 0.0)})
    if kind == "linear" and with_colstats:
        cs = getattr(s2, "_mv_colstats", None)
        if cs is None:
            return {"name": name, "ok": False, "max_abs_err": float("nan"), "detail": "no column statistics next to the carry"}
        buf, rpt = cs[:2]
        tiles = (M + rpt - 1) // rpt
        tf = s2.float()
        if tiles * rpt != M:
            tf = torch.cat([tf, tf.new_zeros(tiles * rpt - M, c)])
        tf = tf.reshape(tiles, rpt, -1)
        results.append(_cmp(name + " colstats of hi", buf.reshape(tiles, -1, 2), torch.stack([tf.sum(1), (tf * tf).sum(1)], dim=-1),
                            atol=2e-3 * rpt * 16, rtol=1e-4))
    # a residual WITHOUT a lo half (a plain fp16 tensor) under a carry launch, and the knob off
    plain_res = s1.clone()
    s3 = ops.gemm(a0, w0, bias=b0, residual=plain_res, carry=True)
    results.append(_cmp(name + " residual without lo", s3.float() + s3._mv_lo.float(), ref1 + plain_res.float(), atol=2e-5, rtol=2e-5))
    return _all_ok(results)


def case_ffn_fused(M=300, seed=700, offset=0.0, with_bias=True):
    """the level-0 feed-forward as one launch (mv_ffn_geglu_f16): residual + GEGLU(LayerNorm(x) W1^T + b1) W2^T + b2, C = 320, hidden =
    1280, against the torch fp32 expression of the chain (and, as a second check, against the three-launch form it replaces)"""
    from musev_amd import ops
    c, hd = 320, 1280
    x = (_rand((M, c), seed, 1.5).float() + offset).half()
    gamma = (_rand((c,), seed + 1, 0.2).float() + 1.0).half()
    beta = _rand((c,), seed + 2, 0.2)
    w1 = _rand((2 * hd, c), seed + 3, 1.0 / math.sqrt(c))
    b1 = _rand((2 * hd,), seed + 4, 0.3) if with_bias else None
    w2 = _rand((c, hd), seed + 5, 1.0 / math.sqrt(hd))
    b2 = _rand((c,), seed + 6, 0.3) if with_bias else None
    res = _rand((M, c), seed + 7, 2.0)
    w1p, b1p = ops.pack_geglu(w1, b1)
    hits = ops.FFN_FUSED_HITS
    got = ops.ffn_geglu(x, gamma, beta, 1e-5, w1p, b1p, w2, b2, res)
    xn = F.layer_norm(x.float(), (c,), gamma.float(), beta.float(), 1e-5)
    h = xn @ w1.float().t() + (b1.float() if with_bias else 0.0)
    ref = (h[:, :hd] * F.gelu(h[:, hd:])) @ w2.float().t() + (b2.float() if with_bias else 0.0) + res.float()
    results = [_cmp(f"ffn fused M{M} offset{offset}", got, ref, atol=6e-3)]
    if ops.FFN_FUSED_HITS != hits + 1:
        return {"name": f"ffn fused M{M}", "ok": False, "max_abs_err": float("nan"), "detail": "the fused launch was not taken"}
    three = ops.gemm(ops.gemm(ops.layernorm(x, gamma, beta, 1e-5), w1p, bias=b1p, geglu=True), w2, bias=b2, residual=res)
    results.append(_cmp(f"ffn fused vs three launches M{M}", got, three.float(), atol=6e-3))
    return _all_ok(results)


def case_tsa_block(b=1, t=13, hw=24, seed=720, offset=0.0, with_bias=True):
    """one temporal self-attention sub-block as one launch (mv_temporal_attn_block_f16): x + to_out(softmax_T(q k^T scale) v) with
    [q | k | v] = LayerNorm(x) Wqkv^T over the t frames of every pixel (C = 320, 8 heads x 40), against the torch fp32 expression of
    the chain and, as a second check, against the three launches it replaces"""
    from musev_amd import ops
    c, heads, d = 320, 8, 40
    M = b * t * hw
    x = (_rand((M, c), seed, 1.5).float() + offset).half()
    gamma = (_rand((c,), seed + 1, 0.2).float() + 1.0).half()
    beta = _rand((c,), seed + 2, 0.2)
    wq, wk, wv = (_rand((c, c), seed + 3 + i, 1.6 / math.sqrt(c)) for i in range(3))
    wo = _rand((c, c), seed + 6, 1.0 / math.sqrt(c))
    bo = _rand((c,), seed + 7, 0.3) if with_bias else None
    scale = d ** -0.5
    got = ops.temporal_attn_block(x, gamma, beta, 1e-5, ops.pack_tsa_qkv(wq, wk, wv, heads, d), ops.pack_tsa_out(wo, heads, d), bo,
                                  b, t, hw, heads, d, scale)
    xn = F.layer_norm(x.float(), (c,), gamma.float(), beta.float(), 1e-5)

    def seq(y):  # [(b t hw), c] -> [(b hw), t, c]
        return y.reshape(b, t, hw, c).permute(0, 2, 1, 3).reshape(b * hw, t, c)
    att = _attn_ref(seq(xn @ wq.float().t()), seq(xn @ wk.float().t()), seq(xn @ wv.float().t()), heads, d, scale)
    att = att.reshape(b, hw, t, c).permute(0, 2, 1, 3).reshape(M, c)
    ref = x.float() + att @ wo.float().t() + (bo.float() if with_bias else 0.0)
    results = [_cmp(f"tsa block b{b} t{t} hw{hw} offset{offset}", got, ref, atol=6e-3)]
    # the three launches: LayerNorm-folded (or plain) q / k / v projection, mv_temporal_attention_f16, to_out + residual
    xln = ops.layernorm(x, gamma, beta, 1e-5)
    qkv = ops.gemm(xln, torch.cat([wq, wk, wv], 0).contiguous())
    a3 = ops.temporal_attention(qkv[:, :c], qkv[:, c:2 * c], qkv[:, 2 * c:], b, t, hw, heads, d, scale)
    three = ops.gemm(a3, wo, bias=bo, residual=x)
    results.append(_cmp(f"tsa block vs three launches b{b} t{t} hw{hw}", got, three.float(), atol=6e-3))
    return _all_ok(results)


def case_xab_block(nkvb=2, rows_per_kvb=256, n_keys=77, seed=760, offset=0.0, with_bias=True, ragged=0):
    """the text cross-attention sub-block as one launch (mv_xattn_block_f16): x + to_out(softmax(LN(x) Wq^T K^T scale) V) against torch fp32 and
    against the three-launch HIP form (LayerNorm + projection, mv_attention_f16, to_out + residual).  ``ragged``: rows cut off the last block."""
    from musev_amd import ops
    c, heads, d = 320, 8, 40
    M = nkvb * rows_per_kvb - ragged
    x = _rand((M, c), seed) + offset
    gamma = _rand((c,), seed + 1) * 0.2 + 1.0
    beta = _rand((c,), seed + 2, 0.2)
    wq = _rand((c, c), seed + 3, 1.0 / math.sqrt(c))
    wo = _rand((c, c), seed + 4, 1.0 / math.sqrt(c))
    bo = _rand((c,), seed + 5, 0.3) if with_bias else None
    kv = _rand((nkvb * n_keys, 2 * c), seed + 6)          # the fused K | V projection of the prompt, as the model hands it over
    k, v = kv[:, :c], kv[:, c:]
    scale = d ** -0.5
    got = ops.xattn_block(x, gamma, beta, 1e-5, ops.pack_xab_q(wq, heads, d), k, v, n_keys, rows_per_kvb, ops.pack_tsa_out(wo, heads, d), bo, heads, d, scale)
    xn = F.layer_norm(x.float(), (c,), gamma.float(), beta.float(), 1e-5)
    q = (xn @ wq.float().t()).reshape(M, heads, d)
    kb = torch.arange(M, device=x.device) // rows_per_kvb
    kk = k.float().reshape(nkvb, n_keys, heads, d)[kb]
    vv = v.float().reshape(nkvb, n_keys, heads, d)[kb]
    p = torch.softmax(torch.einsum("mhd,mkhd->mhk", q, kk) * scale, dim=-1)
    o = torch.einsum("mhk,mkhd->mhd", p, vv).reshape(M, c)
    ref = x.float() + o @ wo.float().t() + (bo.float() if bo is not None else 0.0)
    res = [_cmp(f"xab_block {nkvb}x{rows_per_kvb}-{ragged} keys {n_keys} vs torch fp32", got, ref, atol=6e-3)]
    if ragged == 0:
        # the three-launch form on the same inputs (frames of rows_per_kvb rows, one key batch each)
        xh = ops.layernorm(x, gamma, beta, 1e-5)
        qh = ops.gemm(xh, wq)
        att = ops.attention(qh, [(k, v, n_keys, 1, 1, 0)], nkvb, rows_per_kvb, heads, d, scale)
        plain = ops.gemm(att, wo, bias=bo, residual=x)
        res.append(_cmp(f"xab_block {nkvb}x{rows_per_kvb} keys {n_keys} vs the three-launch form", got, plain.float(), atol=6e-3))
    return _all_ok(res)


def case_tail_carry(n=3, h=16, w=16, c=320, seed=800):
    """the network's tail on a carried stream: conv_norm_out reads hi + lo (statistics of hi, from the producer's column statistics or
    its own pass), normalises in fp32 and hands conv_out two fp16 halves; conv_out (320 -> 4, fp32 out) reads both.  Against the
    torch fp32 expression on the fp32 stream: 4 x tighter than the bar of the fp16 tail."""
    from musev_amd import ops
    hw = h * w
    s32 = (_rand((n * hw, c), seed, 6.0).float() + 0.37 * _rand((n * hw, c), seed + 1).float())
    hi = s32.half()
    ops._set_lo(hi, (s32 - hi.float()).half())   # (the lo half rides on the hi tensor with its torch version, as a producing launch leaves it)
    gamma = (_rand((c,), seed + 2, 0.2).float() + 1.0).half()
    beta = _rand((c,), seed + 3, 0.2)
    wt = _rand((4, c, 3, 3), seed + 4, 1.0 / math.sqrt(9 * c))
    bias = _rand((4,), seed + 5)
    y = ops.groupnorm(hi, gamma, beta, n, hw, eps=1e-5, silu=True, carry=True)
    if getattr(y, "_mv_lo", None) is None:
        return {"name": "tail carry", "ok": False, "max_abs_err": float("nan"), "detail": "groupnorm(carry=True) returned no lo half"}
    ref_n = F.silu(F.group_norm(s32.reshape(n, hw, c).permute(0, 2, 1), 32, gamma.float(), beta.float(), 1e-5))
    results = [_cmp("tail carry: norm hi + lo", y.float() + y._mv_lo.float(), ref_n.permute(0, 2, 1).reshape(n * hw, c), atol=1.5e-4, rtol=1e-4)]
    out = ops.conv3x3_cout_small(y, ops.pack_conv_weight(wt), bias, n, h, w, out_dtype=torch.float32)
    ref = F.conv2d(ref_n.reshape(n, c, h, w), wt.float(), bias.float(), padding=1).permute(0, 2, 3, 1).reshape(n * hw, 4)
    results.append(_cmp("tail carry: conv_out", out, ref, atol=4e-4, rtol=1e-4))
    return _all_ok(results)


def case_groupnorm(n=3, rows=200, c1=320, c2=0, silu=True, eps=1e-5, seed=40):
    from musev_amd import ops
    c = c1 + c2
    x = _rand((n, rows, c), seed) * 1.5 + 0.3
    gamma = _rand((c,), seed + 1) * 0.2 + 1.0
    beta = _rand((c,), seed + 2, 0.2)
    xr = x.float().permute(0, 2, 1)  # [n, c, rows]
    ref = F.group_norm(xr, 32, gamma.float(), beta.float(), eps)
    if silu:
        ref = F.silu(ref)
    ref = ref.permute(0, 2, 1).reshape(n * rows, c)
    xl = x.reshape(n * rows, c)
    if c2:
        got = ops.groupnorm(xl[:, :c1].contiguous(), gamma, beta, n, rows, eps=eps, silu=silu, x2=xl[:, c1:].contiguous())
    else:
        got = ops.groupnorm(xl, gamma, beta, n, rows, eps=eps, silu=silu)
    return _cmp(f"groupnorm n{n} rows{rows} c{c1}+{c2} silu{int(silu)}", got, ref, atol=4e-3)


def case_colstats_groupnorm(kind="conv", n=3, h=16, w=16, cin=64, c=320, c2=0, cfg=None, seed=300, rows_mul=1):
    """producer-side GroupNorm statistics: the conv3x3 / tconv3 / linear epilogue leaves {sum, sum of squares} per (row tile, column)
    behind (mv_gemm_desc.colstats) and ops.groupnorm folds those instead of reading the tensor (mv_groupnorm_cs_f16).  Checked:
    the column statistics themselves against torch sums of the STORED fp16 output, and the GroupNorm (optionally over two
    concatenated producers, groups straddling the seam) against F.group_norm of that output."""
    from musev_amd import ops
    hw = h * w
    old = ops.GEMM_CFG
    if cfg is not None:
        ops.GEMM_CFG = cfg
    try:
        def produce(cout, sd):
            if kind == "conv":
                x = _rand((n * hw, cin), sd)
                wt = _rand((cout, cin, 3, 3), sd + 1, 1.0 / math.sqrt(9 * cin))
                res = _rand((n * hw, cout), sd + 2)
                return ops.conv3x3(x, ops.pack_conv_weight(wt), n, h, w, bias=_rand((cout,), sd + 3), residual=res)
            if kind == "tconv":
                x = _rand((n * hw, cout), sd)   # (b = 1, t = n)
                wt = _rand((cout, cout, 3, 1, 1), sd + 1, 1.0 / math.sqrt(3 * cout))
                return ops.tconv3(x, ops.pack_conv_weight(wt), 1, n, hw, bias=_rand((cout,), sd + 3), residual=x,
                                  alpha=torch.tensor([0.7], dtype=torch.float32, device=DEV))
            x = _rand((n * hw, cin), sd)
            wt = _rand((cout, cin), sd + 1, 1.0 / math.sqrt(cin))
            return ops.gemm(x, wt, bias=_rand((cout,), sd + 3), residual=_rand((n * hw, cout), sd + 2), colstats=True)
        a = produce(c, seed)
        b = produce(c2, seed + 10) if c2 else None
    finally:
        ops.GEMM_CFG = old
    results = []
    for t in (a, b):
        if t is None:
            continue
        cs = getattr(t, "_mv_colstats", None)
        if cs is None:
            return {"name": f"colstats {kind} cfg{cfg}", "ok": False, "max_abs_err": float("nan"), "detail": "producer emitted no column statistics"}
        buf, rpt = cs[:2]
        tiles = (t.shape[0] + rpt - 1) // rpt
        tf = t.float()
        pad = tiles * rpt - t.shape[0]
        if pad:
            tf = torch.cat([tf, tf.new_zeros(pad, tf.shape[1])])
        tf = tf.reshape(tiles, rpt, -1)
        ref = torch.stack([tf.sum(1), (tf * tf).sum(1)], dim=-1)
        results.append(_cmp(f"colstats {kind} cfg{cfg} rpt{rpt} c{t.shape[1]}", buf.reshape(tiles, -1, 2), ref, atol=2e-3 * rpt, rtol=1e-4))
    ctot = c + c2
    gamma = _rand((ctot,), seed + 20) * 0.2 + 1.0
    beta = _rand((ctot,), seed + 21, 0.2)
    n_items, rows = (1, n * hw) if kind == "tconv" else (n // rows_mul, hw * rows_mul)
    full = a if b is None else torch.cat([a, b], dim=1)
    ref = F.group_norm(full.float().reshape(n_items, rows, ctot).permute(0, 2, 1), 32, gamma.float(), beta.float(), 1e-5)
    ref = F.silu(ref).permute(0, 2, 1).reshape(n_items * rows, ctot)
    hits = ops.COLSTATS_HITS
    got = ops.groupnorm(a, gamma, beta, n_items, rows, eps=1e-5, silu=True, x2=b)
    results.append(_cmp(f"groupnorm from colstats {kind} cfg{cfg} c{c}+{c2}", got, ref, atol=4e-3))
    if ops.COLSTATS_HITS != hits + 1:
        return {"name": f"colstats {kind} cfg{cfg}", "ok": False, "max_abs_err": float("nan"), "detail": "groupnorm did not take the column-statistics path"}
    # and the same through the statistics pass (a tensor object without the attribute): the two must agree closely
    plain = ops.groupnorm(a.clone(), gamma, beta, n_items, rows, eps=1e-5, silu=True, x2=None if b is None else b.clone())
    results.append(_cmp(f"groupnorm colstats vs statistics pass {kind} cfg{cfg}", got, plain.float(), atol=2e-3))
    return _all_ok(results)


def case_gn_fold_linear(kind="spatial", n=4, h=16, w=16, cin=64, c=320, n_out=320, cfg=None, seed=900, offset=0.0):
    """GroupNorm (no activation) folded into the projection behind it (ops.groupnorm_fold_linear: mv_groupnorm_cs_fold_linear_f16 + the
    per-group-weight GEMM with a two-half row bias).  spatial: one item per frame (Transformer2DModel.norm -> proj_in); temporal: ONE
    item over all frames + a per-frame row bias (TransformerTemporalModel.norm -> proj_in + frame embedding).  Checked against torch
    fp32 Linear(group_norm(x)) of the STORED producer output, and against the unfolded HIP path (groupnorm + gemm)."""
    from musev_amd import ops
    hw = h * w
    x_in = _rand((n * hw, cin), seed)
    wt = _rand((c, cin), seed + 1, 1.0 / math.sqrt(cin))
    x = ops.gemm(x_in, wt, bias=_rand((c,), seed + 2) + offset, residual=_rand((n * hw, c), seed + 3), colstats=True)   # producer with column statistics
    gamma = _rand((c,), seed + 4) * 0.2 + 1.0
    beta = _rand((c,), seed + 5, 0.2)
    wp = _rand((n_out, c), seed + 6, 1.0 / math.sqrt(c))
    bp = _rand((n_out,), seed + 7, 0.3)
    if kind == "spatial":
        n_items, rows, rb, rbpi = n, hw, None, 1
    else:
        n_items, rows, rbpi = 1, n * hw, n
        rb = _rand((n, n_out + 8), seed + 8, 0.5)[:, :n_out]   # a column slice of a wider tensor, as the batched embedding projection hands it over
    old, old_ratio = ops.GEMM_CFG, ops.GN_FOLD_MAX_RATIO
    ops.GN_FOLD_MAX_RATIO = 1e9   # (the size rule is the caller's business: here the fold itself is under test)
    if cfg is not None:
        ops.GEMM_CFG = cfg
    try:
        hits = ops.GN_FOLD_HITS
        got = ops.groupnorm_fold_linear(x, gamma, beta, n_items, rows, eps=1e-6, groups=32, w=wp, bias=bp, rowbias=rb, rb_per_item=rbpi)
    finally:
        ops.GEMM_CFG, ops.GN_FOLD_MAX_RATIO = old, old_ratio
    if got is None or ops.GN_FOLD_HITS != hits + 1:
        return {"name": f"gn_fold_linear {kind}", "ok": False, "max_abs_err": float("nan"), "detail": "the fold was not taken"}
    g = F.group_norm(x.float().reshape(n_items, rows, c).permute(0, 2, 1), 32, gamma.float(), beta.float(), 1e-6).permute(0, 2, 1).reshape(n_items * rows, c)
    ref = g @ wp.float().t() + bp.float()
    if rb is not None:
        ref = ref + rb.float()[torch.arange(n * hw, device=ref.device) // hw]
    res = [_cmp(f"gn_fold_linear {kind} cfg{cfg} vs torch fp32", got, ref, atol=6e-3)]
    gh = ops.groupnorm(x.clone(), gamma, beta, n_items, rows, eps=1e-6, silu=False)
    plain = ops.gemm(gh, wp, bias=bp, rowbias=rb, rows_per_group=hw if rb is not None else 0)
    res.append(_cmp(f"gn_fold_linear {kind} cfg{cfg} vs groupnorm + gemm", got, plain.float(), atol=8e-3))
    return _all_ok(res)


def case_layernorm(rows=999, c=640, seed=50):
    from musev_amd import ops
    x = _rand((rows, c), seed) * 2.0 + 0.5
    gamma = _rand((c,), seed + 1) * 0.2 + 1.0
    beta = _rand((c,), seed + 2, 0.2)
    got = ops.layernorm(x, gamma, beta, 1e-5)
    ref = F.layer_norm(x.float(), (c,), gamma.float(), beta.float(), 1e-5)
    return _cmp(f"layernorm rows{rows} c{c}", got, ref, atol=4e-3)


def _attn_ref(q, ks, vs, heads, d, scale):
    # q [nb, lq, C]; ks/vs [nb, lk, C] already gathered per query batch
    nb, lq, c = q.shape
    qh = q.float().view(nb, lq, heads, d).transpose(1, 2)
    kh = ks.float().view(nb, -1, heads, d).transpose(1, 2)
    vh = vs.float().view(nb, -1, heads, d).transpose(1, 2)
    s = (qh @ kh.transpose(-1, -2)) * scale
    o = torch.softmax(s, dim=-1) @ vh
    return o.transpose(1, 2).reshape(nb * lq, c)


def case_attention_self(d=40, b=2, t=3, lq=200, cond_idx=0, seed=60, qscale=1.0):
    """reference-only self attention: segments = [own frame, vision-condition frame of the same batch item]."""
    from musev_amd import ops
    heads = 8
    c = heads * d
    nb = b * t
    qkv = _rand((nb * lq, 3 * c), seed, qscale)
    q, k, v = qkv[:, :c], qkv[:, c:2 * c], qkv[:, 2 * c:]
    scale = d ** -0.5
    got = ops.attention(q, [(k, v, lq, 1, 1, 0), (k, v, lq, t, t, cond_idx)], nb, lq, heads, d, scale)
    k3, v3 = k.reshape(nb, lq, c), v.reshape(nb, lq, c)
    cond = [(n // t) * t + cond_idx for n in range(nb)]
    ks = torch.cat([k3, k3[cond]], dim=1)
    vs = torch.cat([v3, v3[cond]], dim=1)
    ref = _attn_ref(q.reshape(nb, lq, c), ks, vs, heads, d, scale)
    # d = 40 / 80 fold scale * log2(e) into the fp16 query fragment (one more fp16 rounding on the score path, the size of the
    # rounding the QKV projection already left in q): the absolute bound scales with |v| (= qscale) and with the score spread
    return _cmp(f"attention self+cond d{d} nb{nb} lq{lq} qscale{qscale}", got, ref, atol=3e-3 * qscale * qscale)


def case_attention_self_ref(d=40, b=2, t=3, lq=150, lr=90, cond_idx=1, seed=65):
    """reference-only self-attention with ReferenceNet tokens: segments = [own frame, vision-condition frame, reference tokens of the
    batch item] under ONE softmax (attention_processor.py:431-491).  For the condition frame itself the second segment repeats the
    first: the kernel does not walk it and counts segment 0 twice instead (exp2(score + 1)) -- the third segment keeps its weight."""
    from musev_amd import ops
    heads = 8
    c = heads * d
    nb = b * t
    qkv = _rand((nb * lq, 3 * c), seed)
    q, k, v = qkv[:, :c], qkv[:, c:2 * c], qkv[:, 2 * c:]
    rkv = _rand((b * lr, 2 * c), seed + 1)
    scale = d ** -0.5
    got = ops.attention(q, [(k, v, lq, 1, 1, 0), (k, v, lq, t, t, cond_idx), (rkv[:, :c], rkv[:, c:], lr, t, 1, 0)], nb, lq, heads, d, scale)
    k3, v3 = k.reshape(nb, lq, c), v.reshape(nb, lq, c)
    cond = [(n // t) * t + cond_idx for n in range(nb)]
    bi = [n // t for n in range(nb)]
    ks = torch.cat([k3, k3[cond], rkv[:, :c].reshape(b, lr, c)[bi]], dim=1)
    vs = torch.cat([v3, v3[cond], rkv[:, c:].reshape(b, lr, c)[bi]], dim=1)
    ref = _attn_ref(q.reshape(nb, lq, c), ks, vs, heads, d, scale)
    return _cmp(f"attention self+cond+ref d{d} nb{nb} lq{lq} lr{lr}", got, ref, atol=3e-3)


def case_attention_cross(d=80, nb=6, t=3, lq=130, lk=77, seed=70, with_ip=True):
    """text cross attention (keys per batch item, shared by its t frames) + IP-Adapter second attention."""
    from musev_amd import ops
    heads = 8
    c = heads * d
    b = nb // t
    q = _rand((nb * lq, c), seed)
    kv = _rand((b * lk, 2 * c), seed + 1)
    k, v = kv[:, :c], kv[:, c:]
    scale = d ** -0.5
    got = ops.attention(q, [(k, v, lk, t, 1, 0)], nb, lq, heads, d, scale)
    bidx = [n // t for n in range(nb)]
    ref = _attn_ref(q.reshape(nb, lq, c), k.reshape(b, lk, c)[bidx], v.reshape(b, lk, c)[bidx], heads, d, scale)
    if with_ip:
        kvi = _rand((b * 4, 2 * c), seed + 2)
        ki, vi = kvi[:, :c], kvi[:, c:]
        got = ops.attention(q, [(ki, vi, 4, t, 1, 0)], nb, lq, heads, d, scale, out=got, accumulate=True, out_scale=0.7)
        ref_ip = _attn_ref(q.reshape(nb, lq, c), ki.reshape(b, 4, c)[bidx], vi.reshape(b, 4, c)[bidx], heads, d, scale)
        ref = ref.half().float() + 0.7 * ref_ip
    return _cmp(f"attention cross d{d} nb{nb} lq{lq} lk{lk} ip{int(with_ip)}", got, ref, atol=3e-3)


def case_attention_groups(d=40, nb=6, t=3, lq=130, lk=77, seed=75, spike=False):
    """text cross attention + 0.7 x IP-Adapter attention (4 tokens) + 0.4 x FaceID attention (16 tokens) of the same queries in ONE
    launch (softmax groups: every term its own normalisation), the text keys as TWO segments of one group; against the sum of the
    three softmax attentions in fp32.  ``spike``: a late key of the text group dominates (rescale inside a group) and the IP group's
    scores sit far below the text group's (the reference must restart with the group)."""
    from musev_amd import ops
    heads = 8
    c = heads * d
    b = nb // t
    q = _rand((nb * lq, c), seed)
    kv = _rand((b * lk, 2 * c), seed + 1)
    if spike:
        kv[lk - 3, :c] *= 6.0
    k, v = kv[:, :c], kv[:, c:]
    kvi = _rand((b * 4, 2 * c), seed + 2) * (0.05 if spike else 1.0)
    kvf = _rand((b * 16, 2 * c), seed + 3)
    scale = d ** -0.5
    bidx = [n // t for n in range(nb)]
    l1 = 40  # the text keys in two segments: rows [0, 40) and [40, lk) of every key batch
    k3, v3 = k.reshape(b, lk, c), v.reshape(b, lk, c)
    ka, va = k3[:, :l1].reshape(b * l1, c).contiguous(), v3[:, :l1].reshape(b * l1, c).contiguous()
    kb, vb = k3[:, l1:].reshape(b * (lk - l1), c).contiguous(), v3[:, l1:].reshape(b * (lk - l1), c).contiguous()
    segs = [(ka, va, l1, t, 1, 0), (kb, vb, lk - l1, t, 1, 0), (kvi[:, :c], kvi[:, c:], 4, t, 1, 0), (kvf[:, :c], kvf[:, c:], 16, t, 1, 0)]
    got = ops.attention(q, segs, nb, lq, heads, d, scale, group_scales=[1.0, None, 0.7, 0.4])
    q3 = q.reshape(nb, lq, c)
    ref = (_attn_ref(q3, k3[bidx], v3[bidx], heads, d, scale)
           + 0.7 * _attn_ref(q3, kvi[:, :c].reshape(b, 4, c)[bidx], kvi[:, c:].reshape(b, 4, c)[bidx], heads, d, scale)
           + 0.4 * _attn_ref(q3, kvf[:, :c].reshape(b, 16, c)[bidx], kvf[:, c:].reshape(b, 16, c)[bidx], heads, d, scale))
    return _cmp(f"attention groups d{d} nb{nb} lq{lq} lk{lk} spike{int(spike)}", got, ref, atol=3e-3)


def case_gemm_weight_stationary(M=200, N=1280, K=640, seed=880, splitk=0, cfg=17):
    """mv_gemm_desc.tile_order = 1 (ops.GEMM_WEIGHT_STATIONARY): the n-major workgroup order of the small-M levels is a bijection onto
    the same tiles -- bit-identical output, with and without a K split; conv3x3 and the linear mode.  The tile is forced (64 x 160:
    4 x 8 tiles here, the grid shape of the 8 x 8-latent level) so that the problem is one where the library's fetch model takes the
    order; the first GPU run of this case (r05a) used the table's 128 x 128 tile, for which the default order already is
    weight-stationary, and reported exactly that."""
    from musev_amd import ops
    x = _rand((M, K), seed)
    w = _rand((N, K), seed + 1, 1.0 / math.sqrt(K))
    bias = _rand((N,), seed + 2)
    n_img, h, wd, cin = 2, 8, 8, 64
    xc = _rand((n_img * h * wd, cin), seed + 3)
    wc = ops.pack_conv_weight(_rand((N, cin, 3, 3), seed + 4, 1.0 / math.sqrt(9 * cin)))
    flag, sk, cf = ops.GEMM_WEIGHT_STATIONARY, ops.GEMM_SPLITK, ops.GEMM_CFG
    try:
        ops.GEMM_SPLITK, ops.GEMM_CFG = splitk, cfg
        ops.GEMM_WEIGHT_STATIONARY = False
        a0, c0 = ops.gemm(x, w, bias=bias), ops.conv3x3(xc, wc, n_img, h, wd)
        ops.GEMM_WEIGHT_STATIONARY = True
        a1, c1 = ops.gemm(x, w, bias=bias), ops.conv3x3(xc, wc, n_img, h, wd)
    finally:
        ops.GEMM_WEIGHT_STATIONARY, ops.GEMM_SPLITK, ops.GEMM_CFG = flag, sk, cf
    ref = x.float() @ w.float().t() + bias.float()
    # the linear problem must really run in the n-major order (the library takes it only where its fetch model prefers it)
    import ctypes as C
    from musev_amd import _lib
    d = _lib.GemmDesc()
    d.a, d.w, d.c, d.M, d.N, d.K, d.lda, d.ldc, d.c1 = 0x10000, 0x20000, 0x30000, M, N, K, K, N, K
    d.mode, d.cfg, d.splitk, d.tile_order = 0, cfg, splitk, 1
    if _lib.load().mv_gemm_weight_stationary(C.byref(d)) != 1:
        return {"name": "gemm weight-stationary", "ok": False, "max_abs_err": float("nan"), "detail": "the case's problem does not take the weight-stationary order"}
    return _all_ok([_cmp(f"gemm weight-stationary M{M} N{N} K{K} split{splitk}", a1, ref, atol=4e-3),
                    {"name": "weight-stationary order == default order (linear)", "ok": bool(torch.equal(a0, a1)), "max_abs_err": (a0.float() - a1.float()).abs().max().item()},
                    {"name": "weight-stationary order == default order (conv3x3)", "ok": bool(torch.equal(c0, c1)), "max_abs_err": (c0.float() - c1.float()).abs().max().item()}])


def case_attention_resident(d=40, nb=6, t=3, lq=130, lk=77, seed=77, groups=True, heads=8, face=True, rows=True):
    """the resident-K/V kernel (mv_attn_desc.resident_kv, ops.XATTN_RESIDENT): text cross-attention (two segments of one softmax
    group) [+ 0.7 x IP-Adapter (4 tokens) + 0.4 x FaceID (16 tokens) as further groups] against the fp32 sum of softmax attentions,
    and against the tiled kernel on the same operands"""
    from musev_amd import ops
    c = heads * d
    b = nb // t
    q = _rand((nb * lq, c), seed)
    kv = _rand((b * lk, 2 * c), seed + 1)
    kv[lk - 3, :c] *= 4.0   # one dominant late key
    k, v = kv[:, :c], kv[:, c:]
    kvi = _rand((b * 4, 2 * c), seed + 2)
    kvf = _rand((b * 16, 2 * c), seed + 3)
    scale = d ** -0.5
    bidx = [n // t for n in range(nb)]
    k3, v3 = k.reshape(b, lk, c), v.reshape(b, lk, c)
    q3 = q.reshape(nb, lq, c)
    ref = _attn_ref(q3, k3[bidx], v3[bidx], heads, d, scale)
    if groups:
        l1 = 40   # the text keys as two segments of one group (d = 80: one segment and no FaceID group -- the LDS image of V of
        ka, va = k3[:, :l1].reshape(b * l1, c).contiguous(), v3[:, :l1].reshape(b * l1, c).contiguous()   # 8 key tiles x 640 columns
        kb, vb = k3[:, l1:].reshape(b * (lk - l1), c).contiguous(), v3[:, l1:].reshape(b * (lk - l1), c).contiguous()   # exceeds 160 KB)
        segs = ([(ka, va, l1, t, 1, 0), (kb, vb, lk - l1, t, 1, 0)] if d == 40 else [(k, v, lk, t, 1, 0)]) + [(kvi[:, :c], kvi[:, c:], 4, t, 1, 0)]
        gs = ([1.0, None] if d == 40 else [1.0]) + [0.7]
        ref = ref + 0.7 * _attn_ref(q3, kvi[:, :c].reshape(b, 4, c)[bidx], kvi[:, c:].reshape(b, 4, c)[bidx], heads, d, scale)
        if d == 40 and face:   # (without it: two groups, the second one the last key tile -- the kernel's compile-time layout)
            segs.append((kvf[:, :c], kvf[:, c:], 16, t, 1, 0))
            gs.append(0.4)
            ref = ref + 0.4 * _attn_ref(q3, kvf[:, :c].reshape(b, 16, c)[bidx], kvf[:, c:].reshape(b, 16, c)[bidx], heads, d, scale)
    else:
        segs, gs = [(k, v, lk, t, 1, 0)], None
    flag, hits, max_d = ops.XATTN_RESIDENT, ops.XATTN_RESIDENT_HITS, ops.XATTN_RESIDENT_MAX_D
    try:
        ops.XATTN_RESIDENT = rows   # True / 1: the launcher's rows per block; >= 16: that many
        ops.XATTN_RESIDENT_MAX_D = 80   # (the model takes the kernel at d = 40 only since round 6; the d = 80 form stays in the library and is tested)
        got = ops.attention(q, segs, nb, lq, heads, d, scale, group_scales=gs)
        took = ops.XATTN_RESIDENT_HITS == hits + 1
        ops.XATTN_RESIDENT = False
        tiled = ops.attention(q, segs, nb, lq, heads, d, scale, group_scales=gs)
    finally:
        ops.XATTN_RESIDENT, ops.XATTN_RESIDENT_MAX_D = flag, max_d
    name = f"attention resident d{d} nb{nb} lq{lq} lk{lk} groups{int(groups)}"
    if not took:
        return {"name": name, "ok": False, "max_abs_err": float("nan"), "detail": "the launch did not take the resident-K/V kernel"}
    return _all_ok([_cmp(name, got, ref, atol=3e-3), _cmp(name + " vs tiled", got, tiled.float(), atol=3e-3)])


def case_attention_spike(d=40):
    """forces online-softmax rescales: one key per 64-key tile has a much larger score than everything before it."""
    from musev_amd import ops
    heads, lq, lk, nb = 8, 64, 320, 1
    c = heads * d
    q = _rand((nb * lq, c), 80)
    k = _rand((nb * lk, c), 81)
    v = _rand((nb * lk, c), 82)
    for tile in range(5):
        k[tile * 64 + 7 * tile + 3] *= (2.0 + 1.5 * tile)  # growing spikes -> the running max jumps in every tile
    scale = d ** -0.5
    got = ops.attention(q, [(k, v, lk, 1, 1, 0)], nb, lq, heads, d, scale)
    ref = _attn_ref(q.reshape(nb, lq, c), k.reshape(nb, lk, c), v.reshape(nb, lk, c), heads, d, scale)
    return _cmp(f"attention spike d{d}", got, ref, atol=3e-3)


def case_temporal_attention(b=2, t=13, hw=70, d=40, seed=90):
    from musev_amd import ops
    heads = 8
    c = heads * d
    qkv = _rand((b * t * hw, 3 * c), seed)
    q, k, v = qkv[:, :c], qkv[:, c:2 * c], qkv[:, 2 * c:]
    scale = d ** -0.5
    got = ops.temporal_attention(q, k, v, b, t, hw, heads, d, scale)

    def seq(x):  # [(b t hw), c] -> [(b hw), t, c]
        return x.reshape(b, t, hw, c).permute(0, 2, 1, 3).reshape(b * hw, t, c)
    ref = _attn_ref(seq(q), seq(k), seq(v), heads, d, scale)  # [(b hw t), c]
    ref = ref.reshape(b, hw, t, c).permute(0, 2, 1, 3).reshape(b * t * hw, c)
    return _cmp(f"temporal_attention b{b} t{t} hw{hw} d{d}", got, ref, atol=3e-3)


def case_geglu():
    from musev_amd import ops
    x = _rand((333, 2560), 100)
    got = ops.geglu(x)
    ref = x.float()[:, :1280] * F.gelu(x.float()[:, 1280:])
    return _cmp("geglu", got, ref, atol=2e-3)


def case_conv_in_out():
    from musev_amd import ops
    n, h, w = 3, 16, 24
    x = _rand((n, 4, h, w), 110)
    wt = _rand((320, 4, 3, 3), 111, 1.0 / 6.0)
    b = _rand((320,), 112)
    ref = F.conv2d(x.float(), wt.float(), b.float(), padding=1).permute(0, 2, 3, 1).reshape(n * h * w, 320)
    xl = x.permute(0, 2, 3, 1).reshape(n * h * w, 4).contiguous()
    got = ops.conv3x3_cin_small(xl, ops.pack_conv_weight(wt), b, n, h, w)
    r1 = _cmp("conv_in 4->320", got, ref, atol=3e-3)
    got_g = ops.conv3x3_cin_small_gemm(xl, ops.pack_conv_weight(wt), b, n, h, w)  # im2col + MFMA path (the product path)
    r1g = _cmp("conv_in 4->320 (im2col + gemm)", got_g, ref, atol=3e-3)
    r1["ok"] = r1["ok"] and r1g["ok"]
    r1["max_abs_err"] = max(r1["max_abs_err"], r1g["max_abs_err"])
    y = _rand((n, 320, h, w), 113)
    wo = _rand((4, 320, 3, 3), 114, 1.0 / math.sqrt(2880))
    bo = _rand((4,), 115)
    ref2 = F.conv2d(y.float(), wo.float(), bo.float(), padding=1).permute(0, 2, 3, 1).reshape(n * h * w, 4)
    yl = y.permute(0, 2, 3, 1).reshape(n * h * w, 320).contiguous()
    got2 = ops.conv3x3_cout_small(yl, ops.pack_conv_weight(wo), bo, n, h, w)
    r2 = _cmp("conv_out 320->4", got2, ref2, atol=3e-3)
    got3 = ops.conv3x3_cout_small(yl, ops.pack_conv_weight(wo), bo, n, h, w, out_dtype=torch.float32)  # unrounded accumulator
    r3 = _cmp("conv_out 320->4 fp32 out", got3, ref2, atol=2e-4)
    r2["ok"] = r2["ok"] and r3["ok"] and got3.dtype == torch.float32
    r1["ok"] = r1["ok"] and r2["ok"]
    r1["name"] = "conv_in/conv_out"
    r1["max_abs_err"] = max(r1["max_abs_err"], r2["max_abs_err"])
    return r1


def case_conv3x3_direct():
    """the PoseGuider conv shapes: odd Cin (3), Cin % 8 == 0 (16, 96), stride 1 and 2 (even and odd sizes), with and
    without the fused SiLU"""
    from musev_amd import ops
    worst = {"ok": True, "max_abs_err": 0.0, "parts": {}}
    for i, (cin, cout, h, w, stride, act) in enumerate([(3, 16, 20, 28, 1, True), (16, 32, 20, 28, 2, True), (16, 16, 9, 11, 2, True),
                                                         (96, 256, 12, 10, 2, True), (32, 96, 7, 9, 1, False), (256, 320, 8, 8, 1, False)]):
        n = 3
        x = _rand((n, cin, h, w), 300 + i)
        wt = _rand((cout, cin, 3, 3), 310 + i, 1.0 / math.sqrt(9 * cin))
        b = _rand((cout,), 320 + i)
        ref = F.conv2d(x.float(), wt.float(), b.float(), stride=stride, padding=1)
        if act:
            ref = F.silu(ref)
        ref = ref.permute(0, 2, 3, 1).reshape(-1, cout)
        xl = x.permute(0, 2, 3, 1).reshape(n * h * w, cin).contiguous()
        got = ops.conv3x3_direct(xl, ops.pack_conv_weight(wt), b, n, h, w, stride=stride, act=ops.MV_ACT_SILU if act else ops.MV_ACT_NONE)
        r = _cmp(f"conv3x3_direct {cin}->{cout} {h}x{w} s{stride} silu{int(act)}", got, ref, atol=3e-3)
        worst["ok"] = worst["ok"] and r["ok"] and tuple(got.shape) == tuple(ref.shape)
        worst["max_abs_err"] = max(worst["max_abs_err"], r["max_abs_err"])
        worst["parts"][r["name"]] = r["ok"]
    worst["name"] = "conv3x3_direct"
    return worst


def case_timestep_embedding():
    from musev_amd import ops
    t = torch.tensor([951.0, 1.0, 0.0, 8.0, 96.0], device=DEV)
    got = ops.timestep_embedding(t, 320)
    half = 160
    freq = torch.exp(-math.log(10000.0) * torch.arange(half, device=DEV, dtype=torch.float32) / half)
    arg = t[:, None] * freq[None]
    ref = torch.cat([torch.cos(arg), torch.sin(arg)], dim=-1)
    return _cmp("timestep_embedding", got, ref, atol=2e-3)


def case_layout_and_misc():
    from musev_amd import ops
    x = _rand((2, 4, 5, 6, 7), 120, dtype=torch.float32)
    y = ops.bcthw_to_bthwc(x)
    ref = x.permute(0, 2, 3, 4, 1).reshape(-1, 4)
    r = _cmp("bcthw_to_bthwc", y, ref, atol=1e-3)
    back = ops.bthwc_to_bcthw(y, 2, 5, 6, 7, dtype=torch.float32)
    r2 = _cmp("bthwc_to_bcthw", back, x.half().float(), atol=1e-6)
    back32 = ops.bthwc_to_bcthw(ref.contiguous(), 2, 5, 6, 7, dtype=torch.float32)  # fp32 rows -> fp32 b c t h w: exact
    r2b = _cmp("bthwc_to_bcthw fp32 src", back32, x, atol=0.0)
    r2["ok"] = r2["ok"] and r2b["ok"]
    a = _rand((64, 40), 121)
    s = ops.silu(a)
    r3 = _cmp("silu", s, F.silu(a.float()), atol=1e-3)
    z = a.clone()
    ops.zero_rows(z, torch.tensor([3, 17], device=DEV))
    refz = a.clone()
    refz[[3, 17]] = 0
    r4 = _cmp("zero_rows", z, refz, atol=0.0, rtol=0.0)
    r5 = _cmp("add", ops.add(a, s), a.float() + s.float(), atol=1e-3)
    # a carried stream tensor (hi + lo) keeps both halves through the add: hi' + lo' = (hi + lo) + b to ~22 bits
    s32 = (_rand((64, 40), 122, 6.0).float() + 0.37 * _rand((64, 40), 123).float())
    hi = s32.half()
    ops._set_lo(hi, (s32 - hi.float()).half())
    y = ops.add(hi, s)
    ylo = ops._lo_of(y)
    r6 = {"name": "add (carried)", "ok": False, "max_abs_err": float("nan")} if ylo is None else \
        _cmp("add (carried)", y.float() + ylo.float(), s32 + s.float(), atol=2e-5, rtol=1e-6)
    parts = (r, r2, r3, r4, r5, r6)
    ok = all(t["ok"] for t in parts)
    return {"name": "layout+misc", "ok": ok, "max_abs_err": max(t["max_abs_err"] for t in parts),
            "parts": {t["name"]: t["ok"] for t in parts}}


def case_upsample_nearest():
    """mv_upsample_nearest_f16 against F.interpolate(size=..., mode="nearest") (diffusers Upsample2D with output_size; the sizes are the
    3 -> 5, 5 -> 10, 9 -> 17 / 13 steps a latent that is not a multiple of 8 produces, plus a down-size and a strided source view): exact"""
    from musev_amd import ops
    parts = []
    for k, (n, h, w_, ho, wo, c) in enumerate([(3, 3, 3, 5, 5, 40), (2, 5, 5, 10, 10, 64), (2, 9, 7, 17, 13, 24), (1, 8, 6, 5, 4, 16), (26, 8, 8, 15, 15, 1280)]):
        x = _rand((n * h * w_, c + 8), 140 + k)[:, :c]   # (a view with a leading dimension of c + 8)
        got = ops.upsample_nearest(x, n, h, w_, ho, wo)
        ref = F.interpolate(x.float().reshape(n, h, w_, c).permute(0, 3, 1, 2), size=(ho, wo), mode="nearest").permute(0, 2, 3, 1).reshape(n * ho * wo, c)
        parts.append(_cmp(f"upsample_nearest {h}x{w_}->{ho}x{wo} c{c}", got, ref, atol=0.0, rtol=0.0))
    ok = all(t["ok"] for t in parts)
    return {"name": "upsample_nearest", "ok": ok, "max_abs_err": max(t["max_abs_err"] for t in parts), "parts": {t["name"]: t["ok"] for t in parts}}


def case_window_loop():
    """gather / scatter-add / CFG + DDIM step against the torch expressions of pipeline_controlnet.py:1902-2117."""
    from musev_amd import ops
    c, t_total, hw, n_cond = 4, 20, 48, 1
    g = torch.Generator().manual_seed(130)
    lat = torch.randn((c, t_total, hw), generator=g).to(DEV)
    cond = torch.randn((c, n_cond, hw), generator=g).to(DEV)
    idx = torch.tensor([16, 17, 18, 19, 0, 1], dtype=torch.int32, device=DEV)
    win = idx.numel()
    inp = ops.window_gather(lat, cond, idx, n_cond, 2)
    frames = torch.cat([cond, lat[:, idx.long()]], dim=1)  # [c, n_cond+win, hw]
    ref_in = frames.permute(1, 2, 0).reshape(-1, c)
    ref_in = torch.cat([ref_in, ref_in], dim=0)
    r1 = _cmp("window_gather", inp, ref_in.half().float(), atol=1e-6)
    both = ops.window_gather(lat, cond, idx, n_cond, 2, hi_lo=True)   # rows [hi | lo]: the fp32 values as two fp16 halves
    r1b = _all_ok([_cmp("window_gather hi_lo: hi", both[:, :c], ref_in.half().float(), atol=0.0, rtol=0.0),
                   _cmp("window_gather hi_lo: hi + lo", both[:, :c].float() + both[:, c:].float(), ref_in.float(), atol=1e-6, rtol=4e-6)])
    r1["ok"] = r1["ok"] and r1b["ok"]
    # condition frames at given slots (the reference's vision_condition_latent_index, data_util.py:242-268 as called at
    # pipeline_controlnet.py:1939-1946): two condition frames named at slots [0, n_cond + win - 1] -> slot 0 = cond 0, slot 1 zeros,
    # the tail slot overwritten by the window's last frame; and [1, 0] -> swapped in front
    cond2 = torch.randn((c, 2, hw), generator=g).to(DEV)
    for slots in ([0, 2 + win - 1], [1, 0], [0, 1]):
        got = ops.window_gather(lat, cond2, idx, 2, 2, cond_slot=torch.tensor(slots, dtype=torch.int32, device=DEV))
        full = torch.zeros((c, 2 + win, hw), device=DEV)
        full.index_copy_(1, torch.tensor(slots, device=DEV), cond2)
        full[:, 2:] = lat[:, idx.long()]
        ref_s = full.permute(1, 2, 0).reshape(-1, c)
        rs = _cmp(f"window_gather cond_slot={slots}", got, torch.cat([ref_s, ref_s], dim=0).half().float(), atol=0.0, rtol=0.0)
        r1["ok"] = r1["ok"] and rs["ok"]
    eps_win = _rand((2 * (n_cond + win) * hw, c), 131)
    acc = torch.zeros((2, c, t_total, hw), device=DEV)
    cnt = torch.zeros((t_total,), device=DEV)
    ops.window_scatter_add(eps_win, idx, n_cond, 2, 0, acc, cnt, True)
    ops.window_scatter_add(eps_win, idx, n_cond, 2, 0, acc, cnt, True)
    e = eps_win.float().reshape(2, n_cond + win, hw, c)[:, n_cond:].permute(0, 3, 1, 2)  # [2, c, win, hw]
    ref_acc = torch.zeros_like(acc)
    ref_acc[:, :, idx.long()] += 2 * e
    ref_cnt = torch.zeros_like(cnt)
    ref_cnt[idx.long()] += 2
    r2 = _cmp("window_scatter_add", acc, ref_acc, atol=1e-6)
    acc32 = torch.zeros_like(acc)
    eps32 = torch.randn((2 * (n_cond + win) * hw, c), generator=g).to(DEV)  # fp32 predictions (the product path)
    ops.window_scatter_add(eps32, idx, n_cond, 2, 0, acc32, cnt.clone(), False)
    ref32 = torch.zeros_like(acc)
    ref32[:, :, idx.long()] += eps32.reshape(2, n_cond + win, hw, c)[:, n_cond:].permute(0, 3, 1, 2)
    r2b = _cmp("window_scatter_add fp32", acc32, ref32, atol=0.0)
    r2["ok"] = r2["ok"] and r2b["ok"]
    r3 = _cmp("counter", cnt, ref_cnt, atol=0.0)
    # step on the covered frames only (others have counter 0): fill counter to avoid 0-division in the check
    cnt2 = torch.where(cnt > 0, cnt, torch.ones_like(cnt))
    lat2 = lat.clone()
    a_t, a_p, gs = 0.3, 0.45, 3.5
    ops.cfg_ddim_step(lat2, acc, cnt2, gs, a_t, a_p)
    eps = acc / cnt2[None, None, :, None]
    eps = eps[0] + gs * (eps[1] - eps[0])
    x0 = (lat - math.sqrt(1 - a_t) * eps) / math.sqrt(a_t)
    ref_lat = math.sqrt(a_p) * x0 + math.sqrt(1 - a_p) * eps
    r4 = _cmp("cfg_ddim_step", lat2, ref_lat, atol=1e-5, rtol=1e-5)
    parts = (r1, r2, r3, r4)
    return {"name": "window loop glue", "ok": all(p["ok"] for p in parts), "max_abs_err": max(p["max_abs_err"] for p in parts),
            "parts": {p["name"]: p["ok"] for p in parts}}


def case_window_units_reduce():
    """the multi-rank accumulation (mv_window_units_reduce) against the scatter-add formulation: config-4-like table (3 ranks x 3
    slots, one unused; windows of 6 and a short one of 3 frames; a wrap-around window), exact in fp32"""
    from musev_amd import ops
    c, t_total, hw, halves, win_max = 4, 14, 40, 2, 6
    wins = [[0, 1, 2, 3, 4, 5], [4, 5, 6, 7, 8, 9], [8, 9, 10, 11, 12, 13], [12, 13, 0]]
    units = [(wi, hf) for wi in range(len(wins)) for hf in range(halves)]           # 8 units over 3 ranks: 3 + 3 + 2
    world, max_units = 3, 3
    shards = [units[0:3], units[3:6], units[6:8]]
    g = torch.Generator().manual_seed(140)
    recv = torch.randn((world * max_units, win_max * hw, c), generator=g).to(DEV)
    pairs = [[[] for _ in range(t_total)] for _ in range(halves)]
    ref = torch.zeros((halves, c, t_total, hw), device=DEV)
    for r in range(world):
        for k, (wi, hf) in enumerate(shards[r]):
            slot = r * max_units + k
            for j, f in enumerate(wins[wi]):
                pairs[hf][f].append((slot, j))
                ref[hf, :, f] += recv[slot, j * hw:(j + 1) * hw].t()
    maxc = max(len(e) for hp in pairs for e in hp)
    tab = torch.full((halves, t_total, maxc, 2), -1, dtype=torch.int32)
    for hf in range(halves):
        for f in range(t_total):
            for q, (slot, j) in enumerate(pairs[hf][f]):
                tab[hf, f, q, 0], tab[hf, f, q, 1] = slot, j
    acc = torch.full((halves, c, t_total, hw), 123.0, device=DEV)  # must be overwritten, not accumulated into
    ops.window_units_reduce(recv, tab.to(DEV), acc)
    return _cmp("window_units_reduce", acc, ref, atol=0.0, rtol=0.0)


def case_softmax_rows(rows=300, cols=1024, seed=150):
    """row softmax in place (VAE mid-block attention): strided rows, large-magnitude scores"""
    from musev_amd import ops
    buf = _rand((rows, cols + 64), seed, 6.0)
    x = buf[:, :cols]
    ref = torch.softmax(x.float(), dim=-1)
    ops.softmax_rows_(x)
    r = _cmp(f"softmax_rows {rows}x{cols}", x, ref, atol=2e-4)
    r["ok"] = r["ok"] and bool((buf[:, cols:].float().abs() > 0).any())  # the tail of the strided rows is untouched noise
    return r


def case_cfg_affine_step():
    """mv_cfg_affine_step against the Euler-discrete step (scheduling_euler_discrete.py:146-162 with gamma = 0):
    x + (sigma_next - sigma) * CFG(acc / counter)."""
    from musev_amd import ops
    c, t_total, hw = 4, 20, 48
    g = torch.Generator().manual_seed(140)
    lat = (14.6 * torch.randn((c, t_total, hw), generator=g)).to(DEV)
    acc = torch.randn((2, c, t_total, hw), generator=g).to(DEV)
    cnt = torch.tensor([1.0, 2.0] * (t_total // 2), device=DEV)
    gs, sig, sig_next = 3.5, 14.6146, 11.8927
    got = lat.clone()
    ops.cfg_affine_step(got, acc, cnt, gs, 1.0, sig_next - sig)
    eps = acc / cnt[None, None, :, None]
    eps = eps[0] + gs * (eps[1] - eps[0])
    r1 = _cmp("cfg_affine_step euler", got, lat + (sig_next - sig) * eps, atol=1e-4, rtol=1e-5)
    got1 = lat.clone()
    ops.cfg_affine_step(got1, acc[:1].contiguous(), cnt, 1.0, 0.5, -2.0)  # one half (no CFG), general cx / ce
    r2 = _cmp("cfg_affine_step single half", got1, 0.5 * lat - 2.0 * (acc[0] / cnt[None, :, None]), atol=1e-4, rtol=1e-5)
    return {"name": "cfg_affine_step", "ok": r1["ok"] and r2["ok"], "max_abs_err": max(r1["max_abs_err"], r2["max_abs_err"])}


def _all_ok(results):
    """one verdict for a list of case results (the first failing one is reported)"""
    bad = [r for r in results if not r["ok"]]
    out = dict((bad or results)[0])
    out["max_abs_err"] = max(r.get("max_abs_err", 0.0) for r in results)
    out["name"] = f"{len(results)} cases: " + results[0]["name"]
    return out


ALL_CASES: List[Tuple[str, Callable[[], Dict]]] = [
    ("tr16_probe", case_tr16_probe),
    ("gemm_plain", lambda: case_gemm(M=1000, N=320, K=640, epilogue=False)),
    ("gemm_epilogue", lambda: case_gemm(M=1000, N=320, K=640)),
    ("gemm_ragged", lambda: case_gemm(M=333, N=196, K=200, seed=3)),
    ("gemm_two_src", lambda: case_gemm(M=517, N=640, K=960, two_src=True, seed=4)),
    ("gemm_big", lambda: case_gemm(M=4096, N=1280, K=1280, seed=5)),
    ("gemm_text_kv", lambda: case_gemm(M=154, N=640, K=768, epilogue=False, seed=6)),
    ("gemm_silu", case_gemm_silu),
    ("gemm_geglu", case_gemm_geglu),
    ("gemm_ln", case_gemm_ln),
    ("gemm_ln_residual_ragged_m", lambda: case_gemm_ln(M=777, N=320, K=640, residual=True, seed=21)),
    ("gemm_ln_geglu", lambda: case_gemm_ln(M=500, N=2560, K=320, geglu=True, seed=22)),
    ("gemm_ln_large_mean", lambda: case_gemm_ln(M=300, N=640, K=1280, offset=12.0, seed=23)),
    ("gemm_ln_every_tile", lambda: _all_ok([case_gemm_ln(M=300, N=640, K=320, seed=24 + c, cfg=c) for c in range(19)])),
    ("conv3x3", case_conv3x3),
    ("conv3x3_two_src", lambda: case_conv3x3(c1=128, c2=64, cout=160, seed=21)),
    ("conv3x3_stride2", lambda: case_conv3x3(stride=2, seed=22)),
    ("conv3x3_upsample", lambda: case_conv3x3(h=8, w=12, upsample=True, seed=23)),
    ("conv3x3_wide", lambda: case_conv3x3(n=2, h=8, w=8, c1=1280, c2=1280, cout=1280, seed=24)),
    ("tconv3", case_tconv3),
    ("tconv3_t13", lambda: case_tconv3(b=2, t=13, hw=64, c=320, seed=31)),
    ("groupnorm", case_groupnorm),
    ("groupnorm_concat", lambda: case_groupnorm(n=2, rows=130, c1=640, c2=320, seed=41)),
    ("groupnorm_nosilu_eps6", lambda: case_groupnorm(n=2, rows=4096, c1=320, silu=False, eps=1e-6, seed=42)),
    ("groupnorm_c2560", lambda: case_groupnorm(n=2, rows=64, c1=1280, c2=1280, seed=43)),
    ("groupnorm_one_launch_c1280", lambda: case_groupnorm(n=5, rows=256, c1=1280, seed=44)),
    ("groupnorm_one_launch_reread", lambda: case_groupnorm(n=2, rows=3328, c1=1280, silu=False, seed=45)),   # temporal, 16x16 level
    ("groupnorm_seam_inside_group", lambda: case_groupnorm(n=2, rows=256, c1=1280, c2=640, seed=46)),          # 60-channel groups: three launches
    ("colstats_conv_groupnorm", case_colstats_groupnorm),
    ("colstats_conv_two_src_seam", lambda: case_colstats_groupnorm(n=2, c=640, c2=320, seed=310)),
    ("colstats_tconv_groupnorm", lambda: case_colstats_groupnorm(kind="tconv", n=5, h=8, w=16, c=320, seed=320)),
    ("colstats_linear_ragged_n", lambda: case_colstats_groupnorm(kind="linear", n=2, cin=320, c=640, seed=330)),
    ("colstats_every_tile", lambda: _all_ok([case_colstats_groupnorm(n=2, c=320, seed=340 + c, cfg=c) for c in range(19)])),
    ("gn_fold_linear_spatial", case_gn_fold_linear),
    ("gn_fold_linear_temporal", lambda: case_gn_fold_linear(kind="temporal", n=5, h=8, w=16, seed=910)),
    ("gn_fold_linear_large_mean_c640", lambda: case_gn_fold_linear(n=3, h=16, w=16, c=640, n_out=640, seed=920, offset=3.0)),
    ("gn_fold_linear_64_row_items", lambda: case_gn_fold_linear(n=6, h=8, w=8, c=1280, n_out=1280, cin=128, seed=930)),   # a 256-row tile would straddle two items
    ("gn_fold_linear_every_tile", lambda: _all_ok([case_gn_fold_linear(n=2, h=16, w=16, seed=940 + c_, cfg=c_) for c_ in range(19)])),
    ("carry_linear", case_carry),
    ("carry_conv", lambda: case_carry(kind="conv", seed=610)),
    ("carry_tconv", lambda: case_carry(kind="tconv", n=5, h=8, w=16, seed=620)),
    ("carry_ragged", lambda: case_carry(kind="linear", n=1, h=9, w=37, cin=200, c=200, seed=630)),
    ("carry_every_tile", lambda: _all_ok([case_carry(kind=("linear", "conv", "tconv")[c_ % 3], n=2, seed=640 + c_, cfg=c_) for c_ in range(19)])),
    ("carry_256x320", lambda: _all_ok([case_carry(kind=k_, n=3, seed=660 + i_, cfg=6) for i_, k_ in enumerate(("linear", "conv", "tconv"))])),
    ("ffn_fused", case_ffn_fused),
    ("ffn_fused_ragged_large_mean", lambda: case_ffn_fused(M=1111, seed=710, offset=6.0)),
    ("ffn_fused_no_bias_one_row_block", lambda: case_ffn_fused(M=128, seed=720, with_bias=False)),
    ("tail_carry", case_tail_carry),
    ("layernorm_320", lambda: case_layernorm(c=320)),
    ("layernorm_640", case_layernorm),
    ("layernorm_1280", lambda: case_layernorm(c=1280, seed=51)),
    ("attention_self_d40", case_attention_self),
    ("attention_self_d80", lambda: case_attention_self(d=80, lq=100, seed=61)),
    ("attention_self_d160", lambda: case_attention_self(d=160, lq=64, seed=62)),
    ("attention_self_d40_big", lambda: case_attention_self(d=40, b=1, t=2, lq=1024, cond_idx=1, seed=63, qscale=2.0)),
    ("attention_self_ref_d40", case_attention_self_ref),
    ("attention_self_ref_d80", lambda: case_attention_self_ref(d=80, lq=100, lr=70, cond_idx=0, seed=66)),
    ("attention_cross_d80", case_attention_cross),
    ("attention_cross_d40", lambda: case_attention_cross(d=40, seed=71)),
    ("attention_cross_d160", lambda: case_attention_cross(d=160, lq=64, seed=72)),
    ("attention_groups_d40", case_attention_groups),
    ("attention_groups_d80", lambda: case_attention_groups(d=80, lq=100, seed=76)),
    ("attention_groups_d40_spike", lambda: case_attention_groups(d=40, lq=300, lk=200, seed=77, spike=True)),
    ("gemm_weight_stationary", case_gemm_weight_stationary),
    ("gemm_weight_stationary_split", lambda: case_gemm_weight_stationary(M=300, K=2560, splitk=4, seed=885)),
    ("attention_resident_d40_groups", case_attention_resident),
    ("attention_resident_d40_text_ip", lambda: case_attention_resident(d=40, nb=4, t=2, lq=200, groups=True, face=False, seed=99)),
    ("attention_resident_d40_text", lambda: case_attention_resident(d=40, nb=4, t=2, lq=1000, groups=False, seed=95)),
    ("attention_resident_d80_groups", lambda: case_attention_resident(d=80, nb=4, t=2, lq=260, groups=True, seed=91)),
    ("attention_resident_rows_per_block", lambda: _all_ok([case_attention_resident(d=40, nb=4, t=2, lq=300, groups=False, seed=101, rows=r) for r in (16, 48, 512)])),
    ("attention_resident_128_keys", lambda: case_attention_resident(d=40, nb=2, t=1, lq=300, lk=128, groups=False, seed=94)),
    ("attention_resident_5_heads", lambda: case_attention_resident(d=40, nb=3, t=3, lq=17, lk=5, groups=False, seed=93, heads=5)),
    ("tsa_block", case_tsa_block),
    ("tsa_block_two_items_t5", lambda: case_tsa_block(b=2, t=5, hw=16, seed=730, offset=0.7)),
    ("tsa_block_t16_no_bias", lambda: case_tsa_block(b=1, t=16, hw=8, seed=740, with_bias=False)),
    ("xab_block", case_xab_block),
    ("xab_block_ragged_large_mean", lambda: case_xab_block(nkvb=3, rows_per_kvb=128, seed=770, offset=0.7, ragged=37)),
    ("xab_block_5_keys_no_bias", lambda: case_xab_block(nkvb=2, rows_per_kvb=384, n_keys=5, seed=780, with_bias=False)),
    ("xab_block_80_keys", lambda: case_xab_block(nkvb=1, rows_per_kvb=4096, n_keys=80, seed=790)),
    ("attention_spike", case_attention_spike),
    ("temporal_attention", case_temporal_attention),
    ("temporal_attention_d160_t4", lambda: case_temporal_attention(b=1, t=4, hw=64, d=160, seed=91)),
    ("temporal_attention_t20", lambda: case_temporal_attention(b=1, t=20, hw=16, d=80, seed=92)),
    ("temporal_attention_d80_t13", lambda: case_temporal_attention(b=2, t=13, hw=33, d=80, seed=93)),
    ("temporal_attention_d40_ragged", lambda: case_temporal_attention(b=1, t=16, hw=71, d=40, seed=94)),
    ("temporal_attention_d160_t13", lambda: case_temporal_attention(b=2, t=13, hw=9, d=160, seed=95)),
    ("geglu", case_geglu),
    ("conv_in_out", case_conv_in_out),
    ("timestep_embedding", case_timestep_embedding),
    ("layout_misc", case_layout_and_misc),
    ("upsample_nearest", case_upsample_nearest),
    ("window_loop", case_window_loop),
    ("window_units_reduce", case_window_units_reduce),
    ("softmax_rows", case_softmax_rows),
    ("softmax_rows_4096", lambda: case_softmax_rows(rows=64, cols=4096, seed=151)),
]


# ---- BASELINE-size cases (`-m gpu` only): every (mode, M, N, K, epilogue) class of the config-2 forward at
# its real size, so that each tile configuration the measured table / the rules select there -- the 256x320 / 256x256 8-wave
# tiles, the 256x160 three-stage counted-wait ring of the M = 6656 level, the split-K path of the 8x8-latent level -- is checked
# against torch on the exact grid shapes the benchmark launches; and the level-0 reference-only self-attention at Lq 4096 x
# Lkv 8192, d = 40 (88 % of the attention FLOPs).
def case_gemm_choice(M, N, K, mode=0):
    """which (tile configuration, K slices) the library picks for a problem (introspection, launches nothing)"""
    import ctypes as C
    from musev_amd import _lib
    d = _lib.GemmDesc()
    d.a = d.w = d.c = 16
    d.M, d.N, d.K, d.mode, d.cfg = M, N, K, mode, -1
    taps = {0: 1, 1: 9, 2: 3}[mode]
    d.c1 = d.lda = K // taps
    d.ldc = N
    d.stride, d.hin, d.win, d.hout, d.wout, d.t, d.hw = 1, 1, 1, 1, 1, 1, 1
    cfg, ns = C.c_int32(), C.c_int32()
    assert _lib.load().mv_gemm_choice(C.byref(d), C.byref(cfg), C.byref(ns)) == 0
    return cfg.value, ns.value


def case_attention_level0(frames=2):
    """level-0 reference-only self-attention of config 2: Lq 4096, K/V = [own frame | vision-condition frame] = 8192 keys, d 40"""
    return case_attention_self(d=40, b=1, t=frames, lq=4096, cond_idx=0, seed=64)


AT_SIZE_CASES: List[Tuple[str, Callable[[], Dict]]] = [
    ("gemm_l0_out_res", lambda: case_gemm(M=106496, N=320, K=320, seed=200)),                    # to_out / proj_out + residual
    ("gemm_l0_qkv", lambda: case_gemm(M=106496, N=960, K=320, epilogue=False, seed=201)),        # fused QKV
    ("gemm_l0_ff2", lambda: case_gemm(M=106496, N=320, K=1280, seed=202)),
    ("gemm_l0_geglu", lambda: case_gemm_geglu(M=106496, C=320)),                                 # N 2560 (packed), 256x256 tiles
    ("gemm_l0_ln_qkv", lambda: case_gemm_ln(M=106496, N=960, K=320, seed=240)),                  # norm1 folded into the fused QKV
    ("gemm_l0_ln_geglu", lambda: case_gemm_ln(M=106496, N=2560, K=320, geglu=True, seed=241)),   # norm3 folded into FF1
    ("gemm_l0_ln_q_half", lambda: case_gemm_ln(M=53248, N=320, K=320, seed=242)),                # norm2 -> to_q, one CFG half
    ("gemm_l1_ln_qkv", lambda: case_gemm_ln(M=26624, N=1920, K=640, seed=243)),
    ("gemm_l0_ln_repeatable", case_gemm_ln_repeatable),
    ("gemm_l0_ln_repeatable_batch2", lambda: case_gemm_ln_repeatable(M=106496, runs=12, seed=31)),
    ("colstats_l0_conv_half", lambda: case_colstats_groupnorm(n=13, h=64, w=64, cin=320, c=320, seed=480)),
    ("colstats_l0_conv_two_src_half", lambda: case_colstats_groupnorm(n=13, h=64, w=64, cin=320, c=640, c2=320, seed=481)),
    ("colstats_l0_tconv_half", lambda: case_colstats_groupnorm(kind="tconv", n=13, h=64, w=64, c=320, seed=482)),
    ("colstats_l1_linear_half", lambda: case_colstats_groupnorm(kind="linear", n=13, h=32, w=32, cin=640, c=640, seed=483)),
    ("carry_l0_linear_half", lambda: case_carry(kind="linear", n=13, h=64, w=64, cin=320, c=320, seed=650)),
    ("carry_l0_conv_half", lambda: case_carry(kind="conv", n=13, h=64, w=64, cin=320, c=320, seed=651)),
    ("carry_l0_tconv_half", lambda: case_carry(kind="tconv", n=13, h=64, w=64, c=320, seed=652)),
    ("ffn_fused_l0_half", lambda: case_ffn_fused(M=53248, seed=730)),
    ("tail_carry_l0_half", lambda: case_tail_carry(n=13, h=64, w=64, seed=810)),
    ("gemm_l1_geglu", lambda: case_gemm_geglu(M=26624, C=640)),
    ("gemm_l1_out_res", lambda: case_gemm(M=26624, N=640, K=640, seed=203)),
    ("gemm_l2_out_res", lambda: case_gemm(M=6656, N=1280, K=1280, seed=204)),                    # 256x160 three-stage ring (26 x 8 blocks)
    ("gemm_l2_qkv", lambda: case_gemm(M=6656, N=3840, K=1280, epilogue=False, seed=205)),
    ("gemm_l2_half", lambda: case_gemm(M=3328, N=1280, K=1280, seed=206)),                       # one CFG half (two-stream default path)
    ("gemm_l3_out_res", lambda: case_gemm(M=1664, N=1280, K=1280, seed=207)),                    # split-K rule
    ("gemm_l3_ff2_half", lambda: case_gemm(M=832, N=1280, K=5120, seed=208)),
    ("conv_l0", lambda: case_conv3x3(n=26, h=64, w=64, c1=320, cout=320, seed=210)),
    ("conv_l0_two_src", lambda: case_conv3x3(n=26, h=64, w=64, c1=320, c2=320, cout=320, seed=211)),
    ("conv_l1_down", lambda: case_conv3x3(n=26, h=64, w=64, c1=320, cout=320, stride=2, seed=212)),
    ("conv_l1_two_src", lambda: case_conv3x3(n=26, h=32, w=32, c1=640, c2=640, cout=640, seed=213)),
    ("conv_l1_up", lambda: case_conv3x3(n=26, h=16, w=16, c1=1280, cout=1280, upsample=True, seed=214)),
    ("conv_l2", lambda: case_conv3x3(n=26, h=16, w=16, c1=1280, cout=1280, seed=215)),
    ("conv_l3", lambda: case_conv3x3(n=26, h=8, w=8, c1=1280, cout=1280, seed=216)),             # M 1664, K 11520: split-K
    ("conv_l3_half_two_src", lambda: case_conv3x3(n=13, h=8, w=8, c1=1280, c2=1280, cout=1280, seed=217)),  # M 832, K 23040
    ("tconv_l0", lambda: case_tconv3(b=2, t=13, hw=4096, c=320, seed=220)),
    ("tconv_l2", lambda: case_tconv3(b=2, t=13, hw=256, c=1280, seed=221)),
    ("tconv_l3", lambda: case_tconv3(b=2, t=13, hw=64, c=1280, seed=222)),                       # M 1664, K 3840: split-K
    ("tconv_l3_half", lambda: case_tconv3(b=1, t=13, hw=64, c=1280, seed=223)),
    ("attention_level0", case_attention_level0),
    # config 3's level-0 reference-only self-attention: own frame | condition frame | ReferenceNet tokens = 12 288 keys; frame 0 is the
    # condition frame itself (its duplicate segment is not walked, the ReferenceNet tokens keep their weight)
    ("attention_level0_refnet_tokens", lambda: case_attention_self_ref(d=40, b=1, t=2, lq=4096, lr=4096, cond_idx=0, seed=67)),
    ("attention_groups_l0_half", lambda: case_attention_groups(d=40, nb=13, t=13, lq=4096, seed=78)),   # level-0 cross attention, one CFG half
    ("attention_groups_l1_half", lambda: case_attention_groups(d=80, nb=13, t=13, lq=1024, seed=79)),
    ("tsa_block_l0_half", lambda: case_tsa_block(b=1, t=13, hw=4096, seed=750)),   # one CFG half of a level-0 temporal sub-block
    ("attention_resident_l0_half", lambda: case_attention_resident(d=40, nb=13, t=13, lq=4096, seed=96)),   # level-0 cross attention, one CFG half
    ("attention_resident_l0_text_ip_half", lambda: case_attention_resident(d=40, nb=13, t=13, lq=4096, face=False, seed=100)),
    ("attention_resident_l0_text_half", lambda: case_attention_resident(d=40, nb=13, t=13, lq=4096, groups=False, seed=97)),
    ("attention_resident_l1_half", lambda: case_attention_resident(d=80, nb=13, t=13, lq=1024, seed=98)),
    ("groupnorm_l0", lambda: case_groupnorm(n=26, rows=4096, c1=320, seed=230)),
    ("groupnorm_l0_tconv", lambda: case_groupnorm(n=2, rows=13 * 4096, c1=320, seed=231)),       # statistics over T*H*W
    ("layernorm_l0", lambda: case_layernorm(rows=106496, c=320, seed=232)),
    ("temporal_attention_l0", lambda: case_temporal_attention(b=2, t=13, hw=4096, d=40, seed=233)),
    # ---- config 5 (768 x 768 -> 96 x 96 latents, 12 + 1 frames): M = 119 808 / 29 952 / 7 488 / 1 872 rows per CFG half (the batch-1
    # launches of the two-stream step), 239 616 / 59 904 / 14 976 / 3 744 for both halves; the table is consulted by nearest M
    ("cfg5_gemm_l0_out_res_half", lambda: case_gemm(M=119808, N=320, K=320, seed=500)),
    ("cfg5_gemm_l0_out_res", lambda: case_gemm(M=239616, N=320, K=320, seed=501)),
    ("cfg5_gemm_l0_ln_qkv_half", lambda: case_gemm_ln(M=119808, N=960, K=320, seed=502)),
    ("cfg5_gemm_l0_geglu_half", lambda: case_gemm_geglu(M=119808, C=320)),
    ("cfg5_gemm_l0_ff2_half", lambda: case_gemm(M=119808, N=320, K=1280, seed=503)),
    ("cfg5_ffn_fused_l0_half", lambda: case_ffn_fused(M=119808, seed=731)),
    ("cfg5_gemm_l1_out_res_half", lambda: case_gemm(M=29952, N=640, K=640, seed=504)),
    ("cfg5_gemm_l1_out_res", lambda: case_gemm(M=59904, N=640, K=640, seed=505)),
    ("cfg5_gemm_l2_qkv_half", lambda: case_gemm(M=7488, N=3840, K=1280, epilogue=False, seed=506)),
    ("cfg5_gemm_l2_out_res", lambda: case_gemm(M=14976, N=1280, K=1280, seed=507)),
    ("cfg5_gemm_l3_out_res_half", lambda: case_gemm(M=1872, N=1280, K=1280, seed=508)),
    ("cfg5_gemm_l3_ff2", lambda: case_gemm(M=3744, N=1280, K=5120, seed=509)),
    ("cfg5_conv_l0_half", lambda: case_conv3x3(n=13, h=96, w=96, c1=320, cout=320, seed=510)),
    ("cfg5_conv_l1_two_src_half", lambda: case_conv3x3(n=13, h=48, w=48, c1=640, c2=640, cout=640, seed=511)),
    ("cfg5_conv_l1_down_half", lambda: case_conv3x3(n=13, h=96, w=96, c1=320, cout=320, stride=2, seed=512)),
    ("cfg5_conv_l3_half", lambda: case_conv3x3(n=13, h=12, w=12, c1=1280, cout=1280, seed=513)),
    ("cfg5_tconv_l0_half", lambda: case_tconv3(b=1, t=13, hw=9216, c=320, seed=514)),
    ("cfg5_tconv_l3", lambda: case_tconv3(b=2, t=13, hw=144, c=1280, seed=515)),
    ("cfg5_colstats_conv_l0_half", lambda: case_colstats_groupnorm(n=13, h=96, w=96, cin=320, c=320, seed=516)),
    ("cfg5_groupnorm_l0_tconv_half", lambda: case_groupnorm(n=1, rows=13 * 9216, c1=320, seed=517)),
    ("cfg5_layernorm_l0_half", lambda: case_layernorm(rows=119808, c=320, seed=518)),
    ("cfg5_attention_level0", lambda: case_attention_self(d=40, b=1, t=2, lq=9216, cond_idx=0, seed=519)),   # Lq 9216 x Lkv 18 432
    ("cfg5_temporal_attention_l0_half", lambda: case_temporal_attention(b=1, t=13, hw=9216, d=40, seed=520)),
]
