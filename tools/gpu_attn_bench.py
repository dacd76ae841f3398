"""Timing of the spatial attention kernels at the config-2 shapes (reference-only self-attention with the vision-condition
segment, text cross-attention) + parity of every attention kernel case.  Usage: python tools/gpu_attn_bench.py [--variants]

--variants: same-box A/B of the attn3 kernel variants (Attn3Cfg VAR bits: 1 = 16-deep contraction tail, 2 = 48-half rows with the
ones column, 4 = four waves per SIMD) through the EXPERIMENT build of the library (musev_amd/csrc/libmusev_hip_exp.so, built with
MV_EXTRA_FLAGS=-DMV_EXPERIMENT MV_LIB_NAME=libmusev_hip_exp.so -- it exports mv_attention_f16_var; the product library does not):
every variant is checked against variant 0's output and against a torch fp32 reference on a small problem, then timed."""
import ctypes as C
import os
import sys

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


def variants():
    from musev_amd import _lib, ops
    exp = os.path.join(ROOT, "musev_amd", "csrc", "libmusev_hip_exp.so")
    lib = C.CDLL(exp)
    lib.mv_attention_f16_var.restype = C.c_int32
    lib.mv_attention_f16_var.argtypes = [C.POINTER(_lib.AttnDesc), C.c_int32, C.c_int32, C.c_void_p]
    lib.mv_last_error.restype = C.c_char_p
    var = [0, 0]
    prod = _lib.load()
    real = prod.mv_attention_f16

    class Shim:  # ops.attention -> the experiment library's variant entry (same descriptor)
        def __getattr__(self, name):
            if name == "mv_attention_f16":
                def f(desc, stream):
                    rc = lib.mv_attention_f16_var(desc, var[0], var[1], stream)
                    if rc:
                        print("ERROR", lib.mv_last_error().decode(), flush=True)
                    return rc
                return f
            return getattr(prod, name)
    _lib._lib = Shim()
    import kernel_cases as kc
    t, heads = 13, 8
    vs40 = tuple(int(x) for x in os.environ.get('ATTN_VARS40', '47,100,101,102,103,104').split(','))
    for d, vs in ((40, vs40), (80, (0, 8))):
        for lq, nb in (((4096, 26), (4096, 13)) if d == 40 else ((1024, 26), (1024, 13))):
            c = heads * d
            qkv = torch.randn(nb * lq, 3 * c, device="cuda").half()
            q, k, v = qkv[:, :c], qkv[:, c:2 * c], qkv[:, 2 * c:]
            kt = torch.randn(2 * 77, 2 * c, device="cuda").half()
            base = None
            for vv in vs:
                var[0] = var[1] = vv
                par = [kc.case_attention_self(d=d, b=1, t=2, lq=200, cond_idx=0), kc.case_attention_cross(d=d, nb=4, t=2, lq=130),
                       kc.case_attention_spike(d=d)] if nb == 26 else []
                out = ops.attention(q, [(k, v, lq, 1, 1, 0), (k, v, lq, t, t, 0)], nb, lq, heads, d, d ** -0.5)
                torch.cuda.synchronize()
                if base is None:
                    base = out.clone()
                dev = (out.float() - base.float()).abs().max().item()
                ms = timeit(lambda: ops.attention(q, [(k, v, lq, 1, 1, 0), (k, v, lq, t, t, 0)], nb, lq, heads, d, d ** -0.5), iters=int(os.environ.get('ATTN_ITERS', '10')), warmup=5)
                msx = timeit(lambda: ops.attention(q, [(kt[:, :c], kt[:, c:], 77, t, 1, 0)], nb, lq, heads, d, d ** -0.5))
                print(f"d{d} var {vv}: self nb{nb} lq{lq} {ms:.3f} ms {4.0 * nb * lq * 2 * lq * c / ms / 1e9:.0f} TF/s | cross {msx * 1e3:.0f} us | "
                      f"|out - first var|max {dev:.2e} | parity {' '.join('PASS' if r['ok'] else 'FAIL(%.2e)' % r['max_abs_err'] for r in par)}", flush=True)
    _lib._lib = prod


def main():
    if "--variants" in sys.argv:
        return variants()
    from musev_amd import ops
    from kernel_cases import ALL_CASES, AT_SIZE_CASES
    for name, fn in list(ALL_CASES) + list(AT_SIZE_CASES):
        if "attention" in name and "temporal" not in name:
            r = fn()
            torch.cuda.synchronize()
            print(f"{'PASS' if r['ok'] else 'FAIL'} {name} err={r.get('max_abs_err')}", flush=True)
    for (lq, d, nb) in ((4096, 40, 26), (4096, 40, 13), (1024, 80, 26), (256, 160, 26)):
        t, heads = 13, 8
        c = heads * d
        qkv = torch.randn(nb * lq, 3 * c, device="cuda").half()
        q, k, v = qkv[:, :c], qkv[:, c:2 * c], qkv[:, 2 * c:]
        ms = timeit(lambda: ops.attention(q, [(k, v, lq, 1, 1, 0), (k, v, lq, t, t, 0)], nb, lq, heads, d, d ** -0.5))
        print(f"attn_self nb{nb} lq{lq} d{d}: {ms:.3f} ms  {4.0 * nb * lq * 2 * lq * c / ms / 1e9:.0f} TF/s", flush=True)
        kt = torch.randn(2 * 77, 2 * c, device="cuda").half()
        ms = timeit(lambda: ops.attention(q, [(kt[:, :c], kt[:, c:], 77, t, 1, 0)], nb, lq, heads, d, d ** -0.5))
        print(f"attn_cross nb{nb} lq{lq} d{d}: {ms:.3f} ms  {4.0 * nb * lq * 77 * c / ms / 1e9:.0f} TF/s", flush=True)


if __name__ == "__main__":
    main()
