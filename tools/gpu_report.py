"""One-shot GPU report: runs every kernel parity case (no stop at first failure) and a set of micro-benchmarks at
the config-2 shapes; writes gpurun_out/kernel_report.json.  Usage: python tools/gpu_report.py [--bench]"""
import json
import os
import sys
import time
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

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
    return s.elapsed_time(e) / iters  # ms


def bench():
    from musev_amd import ops
    out = []
    dev = "cuda"

    def r(*shape, scale=1.0):
        return (torch.randn(*shape, device=dev) * scale).half()

    # GEMMs: (M, N, K) at config-2 levels
    for (M, N, K) in [(106496, 320, 320), (106496, 960, 320), (106496, 2560, 320), (106496, 320, 1280),
                      (26624, 640, 640), (26624, 1920, 640), (26624, 5120, 640), (26624, 640, 2560),
                      (6656, 1280, 1280), (6656, 3840, 1280), (6656, 10240, 1280), (6656, 1280, 5120),
                      (1664, 1280, 1280), (8192, 8192, 8192)]:
        a, w = r(M, K), r(N, K, scale=K ** -0.5)
        ms = timeit(lambda: ops.gemm(a, w), iters=10)
        out.append({"kind": "gemm", "M": M, "N": N, "K": K, "ms": ms, "tflops": 2.0 * M * N * K / ms / 1e9})
    # convs: (n, h, w, cin, cout)
    for (n, h, w, c1, c2, co) in [(26, 64, 64, 320, 0, 320), (26, 32, 32, 640, 0, 640), (26, 16, 16, 1280, 0, 1280),
                                  (26, 8, 8, 1280, 0, 1280), (26, 16, 16, 1280, 1280, 1280), (26, 32, 32, 640, 640, 640),
                                  (26, 64, 64, 320, 320, 320), (26, 64, 64, 640, 320, 320)]:
        x = r(n * h * w, c1)
        x2 = r(n * h * w, c2) if c2 else None
        wt = r(co, 9 * (c1 + c2), scale=(9 * (c1 + c2)) ** -0.5)
        ms = timeit(lambda: ops.conv3x3(x, wt, n, h, w, x2=x2), iters=10)
        out.append({"kind": "conv3x3", "n": n, "hw": h, "cin": c1 + c2, "cout": co, "ms": ms,
                    "tflops": 2.0 * n * h * w * 9 * (c1 + c2) * co / ms / 1e9})
    for (b, t, hw, c) in [(2, 13, 4096, 320), (2, 13, 1024, 640), (2, 13, 256, 1280), (2, 13, 64, 1280)]:
        x = r(b * t * hw, c)
        wt = r(c, 3 * c, scale=(3 * c) ** -0.5)
        ms = timeit(lambda: ops.tconv3(x, wt, b, t, hw), iters=10)
        out.append({"kind": "tconv3", "hw": hw, "c": c, "ms": ms, "tflops": 2.0 * b * t * hw * 3 * c * c / ms / 1e9})
    # attention: reference-only self attention at the three levels
    for (lq, d) in [(4096, 40), (1024, 80), (256, 160), (64, 160)]:
        nb, t, heads = 26, 13, 8
        c = heads * d
        qkv = r(nb * lq, 3 * c)
        q, k, v = qkv[:, :c], qkv[:, c:2 * c], qkv[:, 2 * c:]
        ms = timeit(lambda: ops.attention(q, [(k, v, lq, 1, 1, 0), (k, v, lq, t, t, 0)], nb, lq, heads, d, d ** -0.5), iters=10)
        out.append({"kind": "attn_self", "lq": lq, "d": d, "ms": ms, "tflops": 4.0 * nb * lq * 2 * lq * c / ms / 1e9})
        kt = r(2 * 77, 2 * c)
        ms = timeit(lambda: ops.attention(q, [(kt[:, :c], kt[:, c:], 77, t, 1, 0)], nb, lq, heads, d, d ** -0.5), iters=10)
        out.append({"kind": "attn_cross", "lq": lq, "d": d, "ms": ms, "tflops": 4.0 * nb * lq * 77 * c / ms / 1e9})
        b = 2
        qkv2 = r(b * t * lq, 3 * c)
        ms = timeit(lambda: ops.temporal_attention(qkv2[:, :c], qkv2[:, c:2 * c], qkv2[:, 2 * c:], b, t, lq, heads, d, d ** -0.5), iters=10)
        out.append({"kind": "attn_temporal", "hw": lq, "d": d, "ms": ms,
                    "GBps": (4.0 * b * t * lq * c * 2) / ms / 1e6})
    # norms (HBM-bound): report GB/s (read + write)
    for (n, rows, c) in [(26, 4096, 320), (26, 1024, 640), (26, 256, 1280), (2, 13 * 4096, 320), (26, 4096, 960)]:
        x = r(n * rows, c)
        g_, b_ = r(c), r(c)
        ms = timeit(lambda: ops.groupnorm(x, g_, b_, n, rows, eps=1e-5, silu=True), iters=10)
        out.append({"kind": "groupnorm", "n": n, "rows": rows, "c": c, "ms": ms, "GBps": 3.0 * n * rows * c * 2 / ms / 1e6})
    for (rows, c) in [(106496, 320), (26624, 640), (6656, 1280)]:
        x = r(rows, c)
        g_, b_ = r(c), r(c)
        ms = timeit(lambda: ops.layernorm(x, g_, b_), iters=10)
        out.append({"kind": "layernorm", "rows": rows, "c": c, "ms": ms, "GBps": 2.0 * rows * c * 2 / ms / 1e6})
    x = r(106496, 2560)
    ms = timeit(lambda: ops.geglu(x), iters=10)
    out.append({"kind": "geglu", "rows": 106496, "c": 2560, "ms": ms, "GBps": 1.5 * 106496 * 2560 * 2 / ms / 1e6})
    return out


def main():
    from kernel_cases import ALL_CASES
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    report = {"device": torch.cuda.get_device_name(0), "cases": [], "bench": []}
    t0 = time.time()
    for name, fn in ALL_CASES:
        try:
            res = fn()
            torch.cuda.synchronize()
        except Exception as ex:  # noqa: BLE001
            res = {"name": name, "ok": False, "error": repr(ex), "trace": traceback.format_exc()[-800:]}
        res["case"] = name
        report["cases"].append(res)
        print(("PASS " if res.get("ok") else "FAIL ") + name + "  " + json.dumps({k: v for k, v in res.items() if k not in ("trace", "case")}, default=str)[:400], flush=True)
    report["cases_seconds"] = time.time() - t0
    if "--bench" in sys.argv:
        try:
            report["bench"] = bench()
            for b in report["bench"]:
                print("BENCH " + json.dumps(b), flush=True)
        except Exception as ex:  # noqa: BLE001
            report["bench_error"] = repr(ex) + traceback.format_exc()[-800:]
            print("BENCH ERROR", report["bench_error"])
    with open(os.path.join(ROOT, "gpurun_out", "kernel_report.json"), "w") as f:
        json.dump(report, f, indent=1, default=str)
    n_fail = sum(1 for c in report["cases"] if not c.get("ok"))
    print(f"{len(report['cases']) - n_fail} passed, {n_fail} failed")


if __name__ == "__main__":
    main()
