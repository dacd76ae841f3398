"""A/B of tile configurations of the implicit-GEMM kernel on the GPU box (a "variant" is a catalogue id forced through
mv_gemm_desc.cfg; -1 = measured table + rules, -2 = rules only): parity of every implicit-GEMM case per variant, then micro-benchmarks
at the config-2 shapes WITH the epilogues the model uses.  Prints one table; writes gpurun_out/<tag>_gemm_ab.json.
Usage: python tools/gpu_gemm_ab.py <tag> [variants...]"""
import json
import os
import sys
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import torch  # noqa: E402


def timeit(fn, iters=10, warmup=2):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def shapes():
    from musev_amd import ops
    dev = "cuda"

    def r(*shape, scale=1.0):
        return (torch.randn(*shape, device=dev) * scale).half()

    out = []
    for (M, N, K, epi) in [(106496, 320, 320, "res"), (106496, 960, 320, ""), (106496, 2560, 320, "geglu"), (106496, 320, 1280, "res"),
                           (26624, 640, 640, "res"), (26624, 1920, 640, ""), (26624, 5120, 640, "geglu"), (26624, 640, 2560, "res"),
                           (6656, 1280, 1280, "res"), (6656, 3840, 1280, ""), (6656, 10240, 1280, "geglu"), (6656, 1280, 5120, "res"),
                           (1664, 1280, 1280, "res"), (8192, 8192, 8192, "")]:
        a, w = r(M, K), r(N, K, scale=K ** -0.5)
        bias = r(N)
        if epi == "geglu":
            wp, bp = ops.pack_geglu(w, bias)
            fn = lambda a=a, wp=wp, bp=bp: ops.gemm(a, wp, bias=bp, geglu=True)
        elif epi == "res":
            res = r(M, N)
            fn = lambda a=a, w=w, bias=bias, res=res: ops.gemm(a, w, bias=bias, residual=res)
        else:
            fn = lambda a=a, w=w: ops.gemm(a, w)
        out.append((f"gemm {M}x{N}x{K} {epi}", 2.0 * M * N * K, fn))
    for (n, h, w_, c1, c2, co) in [(26, 64, 64, 320, 0, 320), (26, 32, 32, 640, 0, 640), (26, 16, 16, 1280, 0, 1280),
                                   (26, 8, 8, 1280, 0, 1280), (26, 16, 16, 1280, 1280, 1280), (26, 32, 32, 640, 640, 640),
                                   (26, 64, 64, 320, 320, 320), (26, 64, 64, 640, 320, 320)]:
        x = r(n * h * w_, c1)
        x2 = r(n * h * w_, c2) if c2 else None
        wt = r(co, 9 * (c1 + c2), scale=(9 * (c1 + c2)) ** -0.5)
        bias, res = r(co), r(n * h * w_, co)
        fn = lambda x=x, wt=wt, n=n, h=h, w_=w_, x2=x2, bias=bias, res=res: ops.conv3x3(x, wt, n, h, w_, x2=x2, bias=bias, residual=res)
        out.append((f"conv3x3 n{n} {h}x{w_} {c1}+{c2}->{co}", 2.0 * n * h * w_ * 9 * (c1 + c2) * co, fn))
    for (b, t, hw, c) in [(2, 13, 4096, 320), (2, 13, 1024, 640), (2, 13, 256, 1280), (2, 13, 64, 1280)]:
        x = r(b * t * hw, c)
        wt = r(c, 3 * c, scale=(3 * c) ** -0.5)
        bias = r(c)
        fn = lambda x=x, wt=wt, b=b, t=t, hw=hw, bias=bias: ops.tconv3(x, wt, b, t, hw, bias=bias)
        out.append((f"tconv3 b{b} t{t} hw{hw} c{c}", 2.0 * b * t * hw * 3 * c * c, fn))
    return out


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "ab"
    variants = [int(v) for v in sys.argv[2:]] or [-1, -2]
    from musev_amd import _lib
    from kernel_cases import ALL_CASES
    lib = _lib.load()
    report = {"device": torch.cuda.get_device_name(0), "variants": {}}
    sh = shapes()
    from musev_amd import ops as _ops
    for v in variants:   # a "variant" is a forced catalogue id (-1 = measured table + rules, -2 = rules only)
        _ops.GEMM_CFG = v
        rep = {"cases": {}, "bench": {}}
        for name, fn in ALL_CASES:
            if not name.startswith(("gemm", "conv3x3", "tconv3")):
                continue
            try:
                res = fn()
                torch.cuda.synchronize()
            except Exception as ex:  # noqa: BLE001
                res = {"ok": False, "error": repr(ex), "trace": traceback.format_exc()[-600:]}
            rep["cases"][name] = {"ok": bool(res.get("ok")), "max_abs_err": res.get("max_abs_err"), "error": res.get("error")}
            print(f"variant {v} {'PASS' if res.get('ok') else 'FAIL'} {name} err={res.get('max_abs_err')} {res.get('error', '')}", flush=True)
        rep["all_ok"] = all(c["ok"] for c in rep["cases"].values())
        for name, flops, fn in sh:
            try:
                ms = timeit(fn)
                rep["bench"][name] = {"ms": ms, "tflops": flops / ms / 1e9}
            except Exception as ex:  # noqa: BLE001
                rep["bench"][name] = {"error": repr(ex)}
        report["variants"][str(v)] = rep
    _ops.GEMM_CFG = -1
    print(f"{'shape':44s}" + "".join(f"  v{v:>1d} TF/s   ms   " for v in variants))
    for name, _, _ in sh:
        line = f"{name:44s}"
        for v in variants:
            b = report["variants"][str(v)]["bench"][name]
            line += f"  {b.get('tflops', 0):7.1f} {b.get('ms', 0):7.3f}"
        print(line)
    for v in variants:
        print(f"variant {v}: all parity cases ok = {report['variants'][str(v)]['all_ok']}")
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", f"{tag}_gemm_ab.json"), "w") as f:
        json.dump(report, f, indent=1, default=str)


if __name__ == "__main__":
    main()
