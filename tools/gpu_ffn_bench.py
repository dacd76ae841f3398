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


ABLATE = "--ablate" in sys.argv
if ABLATE:
    # ablation through the EXPERIMENT build (MV_EXTRA_FLAGS=-DMV_EXPERIMENT MV_LIB_NAME=libmusev_hip_exp.so): flags bits 8.. switch parts of
    # the kernel off (wrong results, timing only) -- what bounds the block?
    import ctypes as C
    from musev_amd import _lib
    exp = C.CDLL(os.path.join(ROOT, "musev_amd", "csrc", "libmusev_hip_exp.so"))
    exp.mv_ffn_geglu_f16.restype = C.c_int32
    exp.mv_ffn_geglu_f16.argtypes = [C.POINTER(_lib.FfnDesc), C.c_void_p]
    prod = _lib.load()
    abl = [0]

    class Shim:
        def __getattr__(self, name):
            if name == "mv_ffn_geglu_f16":
                def f(desc, stream):
                    desc._obj.flags |= abl[0] << 8
                    return exp.mv_ffn_geglu_f16(desc, stream)
                return f
            return getattr(prod, name)
    _lib._lib = Shim()
c, hd = 320, 1280
gamma, beta = rnd((c,), 1, 0.2) + 1, rnd((c,), 2, 0.2)
w1p, b1p = ops.pack_geglu(rnd((2 * hd, c), 3, 1 / math.sqrt(c)), rnd((2 * hd,), 4, 0.3))
w2, b2 = rnd((c, hd), 5, 1 / math.sqrt(hd)), rnd((c,), 6, 0.3)
for M in ((53248,) if ABLATE else (53248, 106496, 119808)):
    xs = [rnd((M, c), 10 + i, 1.5) for i in range(4)]   # cycled: 4 x 34 MB inputs
    k = [0]

    def fused():
        k[0] = (k[0] + 1) % 4
        return ops.ffn_geglu(xs[k[0]], gamma, beta, 1e-5, w1p, b1p, w2, b2, xs[k[0]])

    def three():
        k[0] = (k[0] + 1) % 4
        x = xs[k[0]]
        return ops.gemm(ops.gemm(ops.layernorm(x, gamma, beta, 1e-5), w1p, bias=b1p, geglu=True), w2, bias=b2, residual=x)

    if ABLATE:
        for bits, what in ((0, "full kernel"), (1, "no gelu (value * gate)"), (2, "no phase-1 MFMAs"), (4, "no phase-2 MFMAs"), (6, "no MFMAs at all"),
                           (7, "no MFMAs, no gelu"), (15, "no MFMAs, no gelu, no g write: the weight stream + barriers + fragment reads")):
            abl[0] = bits
            print(f"M {M} ablate {bits:2d} {what}: {timed(fused):.1f} us", flush=True)
        continue
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
