"""K sweep of the implicit-GEMM kernel at fixed (M, N): time = fixed part (launch, prologue, epilogue, tail) + slope * K tiles.
Separates what a faster main loop can buy from what only fewer / fatter launches can.  Usage: python tools/gpu_ksweep.py <tag>"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def timeit(fn, iters=20, warmup=3):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3  # us


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "ksweep"
    from musev_amd import ops
    dev = "cuda"
    rep = {}
    x = torch.randn(1024, device=dev).half()
    rep["tiny_kernel_us"] = timeit(lambda: ops.silu(x), iters=200)
    print(f"back-to-back tiny kernel: {rep['tiny_kernel_us']:.2f} us per launch")
    for (M, N) in ((26624, 640), (106496, 320), (6656, 1280), (106496, 960)):
        for cfg in (0, 6, 12, 16):
            for epi in ("none", "res"):
                row = []
                for K in (64, 128, 320, 640, 1280, 2560):
                    a = torch.randn(M, K, device=dev).half()
                    w = (torch.randn(N, K, device=dev) * K ** -0.5).half()
                    bias = torch.randn(N, device=dev).half()
                    res = torch.randn(M, N, device=dev).half() if epi == "res" else None
                    out = torch.empty(M, N, device=dev, dtype=torch.float16)
                    ops.GEMM_CFG, ops.GEMM_SPLITK = cfg, 1
                    us = timeit(lambda: ops.gemm(a, w, bias=bias if epi == "res" else None, residual=res, out=out))
                    row.append((K, us))
                ops.GEMM_CFG, ops.GEMM_SPLITK = -1, 0
                # least squares over K >= 320
                pts = [(k / 64.0, t) for k, t in row if k >= 320]
                n = len(pts)
                sx = sum(p[0] for p in pts); sy = sum(p[1] for p in pts)
                sxx = sum(p[0] * p[0] for p in pts); sxy = sum(p[0] * p[1] for p in pts)
                slope = (n * sxy - sx * sy) / (n * sxx - sx * sx)
                icpt = (sy - slope * sx) / n
                rep[f"M{M} N{N} cfg{cfg} {epi}"] = {"us": row, "slope_us_per_ktile": slope, "fixed_us": icpt}
                print(f"M{M:<7d} N{N:<5d} cfg {cfg:2d} {epi:4s}: " + " ".join(f"K{k}:{t:6.1f}" for k, t in row) +
                      f" | fixed {icpt:5.1f} us + {slope:5.2f} us / K tile ({2.0 * M * N * 64 / slope / 1e6:5.0f} TF/s in the loop)", flush=True)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", f"{tag}_ksweep.json"), "w") as f:
        json.dump(rep, f, indent=1)


if __name__ == "__main__":
    main()
