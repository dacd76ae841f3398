"""The temporal self-attention sub-block at level-0 size (one CFG half: 13 frames x 4096 pixels x 320 channels): the fused launch
(mv_temporal_attn_block_f16) against the three launches it replaces (LayerNorm-folded q / k / v projection, mv_temporal_attention_f16,
to_out + residual), inputs cycled through 4 buffers.   python tools/gpu_tsa_bench.py"""
import math
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from musev_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)


def rnd(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).half().to(dev)


def timed(fn, reps=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    c, heads, d = 320, 8, 40
    for b, t, hw in ((1, 13, 4096), (2, 13, 4096), (1, 13, 9216)):
        M = b * t * hw
        xs = [rnd((M, c), 10 + i, 1.5) for i in range(4)]
        gamma, beta = rnd((c,), 1, 0.2) + 1, rnd((c,), 2, 0.2)
        wq, wk, wv = (rnd((c, c), 3 + i, 1.6 / math.sqrt(c)) for i in range(3))
        wo, bo = rnd((c, c), 6, 1.0 / math.sqrt(c)), rnd((c,), 7, 0.3)
        wqkv = torch.cat([wq, wk, wv], 0).contiguous()
        wp, wop = ops.pack_tsa_qkv(wq, wk, wv, heads, d), ops.pack_tsa_out(wo, heads, d)
        lnf = ops.ln_fold_weights(wqkv, gamma, beta, None) if hasattr(ops, "ln_fold_weights") else None
        out = torch.empty_like(xs[0])
        k = [0]

        def fused():
            k[0] += 1
            ops.temporal_attn_block(xs[k[0] % 4], gamma, beta, 1e-5, wp, wop, bo, b, t, hw, heads, d, d ** -0.5, out=out)

        def three():
            k[0] += 1
            x = xs[k[0] % 4]
            xn = ops.layernorm(x, gamma, beta, 1e-5)
            qkv = ops.gemm(xn, wqkv)
            a = ops.temporal_attention(qkv[:, :c], qkv[:, c:2 * c], qkv[:, 2 * c:], b, t, hw, heads, d, d ** -0.5)
            ops.gemm(a, wo, bias=bo, residual=x, out=out)

        us_f, us_3 = timed(fused), timed(three)
        flops = 2.0 * M * c * 4 * c
        print(f"temporal sub-block b {b} t {t} hw {hw} (M = {M}): fused {us_f:7.1f} us ({flops / us_f / 1e6:6.0f} TFLOP/s)   "
              f"LayerNorm + QKV + attention + to_out {us_3:7.1f} us (the model's form folds the LayerNorm into the projection: ~20 us less)")


if __name__ == "__main__":
    main()
