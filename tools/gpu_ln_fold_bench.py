"""Same-box micro A/B of LayerNorm folding: for the LayerNorm -> projection pairs of the UNet (norm1 -> QKV, norm2 -> to_q,
norm3 -> GEGLU) at the level-0 .. level-2 sizes, device time of  mv_layernorm_f16 + plain mv_gemm_f16  against the folded
mv_gemm_f16 (ln_colsum / ln_colbias), each as `reps` back-to-back launches between one HIP event pair.
Usage: python tools/gpu_ln_fold_bench.py [path/to/alternative/libmusev_hip.so ...]   (extra libraries = experiment builds of the
kernel, e.g. other MV_LN_VARIANT values; each is timed on the same operands)"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def timed(fn, reps=20):
    fn()
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3  # us


def main():
    from musev_amd import _lib, ops
    prod = _lib.load()
    libs = [("product", prod)]
    for path in sys.argv[1:]:
        lib = C.CDLL(path)
        for name, (res, args) in _lib.SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        libs.append((os.path.basename(path), lib))
    shapes = []
    for M in (53248, 106496):
        shapes += [(M, 960, 320, False), (M, 320, 320, False), (M, 2560, 320, True)]
    for M in (13312, 26624):
        shapes += [(M, 1920, 640, False), (M, 640, 640, False), (M, 5120, 640, True)]
    for M in (3328, 6656):
        shapes += [(M, 3840, 1280, False), (M, 1280, 1280, False), (M, 10240, 1280, True)]
    g = torch.Generator(device="cuda").manual_seed(0)
    tot = {name: [0.0, 0.0] for name, _ in libs}
    for M, N, K, geglu in shapes:
        x = torch.randn(M, K, device="cuda", generator=g).half()
        gamma = (1 + 0.1 * torch.randn(K, device="cuda", generator=g)).half()
        beta = (0.1 * torch.randn(K, device="cuda", generator=g)).half()
        w = (torch.randn(N, K, device="cuda", generator=g) / K ** 0.5).half()
        b = (0.1 * torch.randn(N, device="cuda", generator=g)).half()
        if geglu:
            w, b = ops.pack_geglu(w, b)
        wf, cs, cb = ops.fold_layernorm(w, b, gamma, beta)
        line = f"M{M:<7d} N{N:<6d} K{K:<5d} {'geglu' if geglu else '     '}"
        for name, lib in libs:
            _lib._lib = lib
            ln_us = timed(lambda: ops.layernorm(x, gamma, beta, 1e-5))
            y = ops.layernorm(x, gamma, beta, 1e-5)
            plain_us = timed(lambda: ops.gemm(y, w, bias=b, geglu=geglu))
            both_us = timed(lambda: ops.gemm(ops.layernorm(x, gamma, beta, 1e-5), w, bias=b, geglu=geglu))
            fold_us = timed(lambda: ops.gemm(x, wf, ln=(cs, cb, 1e-5), geglu=geglu)) if ops.ln_fold_applies(M, N, K, geglu) or True else float("nan")
            ref = ops.gemm(y, w, bias=b, geglu=geglu).float()
            got = ops.gemm(x, wf, ln=(cs, cb, 1e-5), geglu=geglu).float()
            err = (got - ref).abs().max().item()
            tot[name][0] += both_us
            tot[name][1] += fold_us
            line += f" | {name}: LN {ln_us:5.1f} + GEMM {plain_us:6.1f} = {both_us:6.1f} us, folded {fold_us:6.1f} us ({fold_us / both_us - 1:+.0%}) err {err:.1e}"
        _lib._lib = prod
        print(line, flush=True)
    for name, (a, b) in tot.items():
        print(f"{name}: sum LN + GEMM {a:.0f} us, sum folded {b:.0f} us ({b / a - 1:+.1%})")


if __name__ == "__main__":
    main()
