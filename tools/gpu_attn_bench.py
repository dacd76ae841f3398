"""Timing of the spatial attention kernels at the config-2 shapes (reference-only self-attention with the vision-condition
segment, text cross-attention) + parity of every attention kernel case.  Usage: python tools/gpu_attn_bench.py"""
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


def main():
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
