"""Level-0 feed-forward: the fused launch (mv_ffn_geglu_f16) against the three launches it replaces, on the batch-1 / batch-2 shapes
of config 2 and the batch-1 shape of config 5 (run on the MI355X):   python tools/gpu_ffn_bench.py"""
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
    return (torch.randn(shape, generator=g) * scale).to(torch.float16).to(dev)


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


c, hd = 320, 1280
gamma, beta = rnd((c,), 1, 0.2) + 1, rnd((c,), 2, 0.2)
w1p, b1p = ops.pack_geglu(rnd((2 * hd, c), 3, 1 / math.sqrt(c)), rnd((2 * hd,), 4, 0.3))
w2, b2 = rnd((c, hd), 5, 1 / math.sqrt(hd)), rnd((c,), 6, 0.3)
for M in (53248, 106496, 119808):
    xs = [rnd((M, c), 10 + i, 1.5) for i in range(4)]   # cycled: 4 x 34 MB inputs
    k = [0]

    def fused():
        k[0] = (k[0] + 1) % 4
        return ops.ffn_geglu(xs[k[0]], gamma, beta, 1e-5, w1p, b1p, w2, b2, xs[k[0]])

    def three():
        k[0] = (k[0] + 1) % 4
        x = xs[k[0]]
        return ops.gemm(ops.gemm(ops.layernorm(x, gamma, beta, 1e-5), w1p, bias=b1p, geglu=True), w2, bias=b2, residual=x)

    ops.FFN_ROTATE = False
    tf = timed(fused)
    ops.FFN_ROTATE = True
    tr = timed(fused)
    t3 = timed(three)
    fl = 2.0 * M * c * 2 * hd + 2.0 * M * hd * c
    x0 = xs[0]
    a_ = ops.ffn_geglu(x0, gamma, beta, 1e-5, w1p, b1p, w2, b2, x0).float()
    b_ = ops.gemm(ops.gemm(ops.layernorm(x0, gamma, beta, 1e-5), w1p, bias=b1p, geglu=True), w2, bias=b2, residual=x0).float()
    print(f"M {M}: fused {tf:.1f} us ({fl / tf / 1e6:.0f} TFLOP/s), rotated chunk order {tr:.1f} us ({fl / tr / 1e6:.0f})   three launches {t3:.1f} us "
          f"({fl / t3 / 1e6:.0f} TFLOP/s)   diff max {(a_ - b_).abs().max().item():.2e}", flush=True)
