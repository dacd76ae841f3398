"""Timing of the temporal attention and GroupNorm / LayerNorm launches at the config-2 shapes (b = 2 CFG halves, 13 frames), with the
bytes each launch has to move and the HBM rate that implies.  Usage: python tools/gpu_norm_tattn_bench.py"""
import os
import sys

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
    return s.elapsed_time(e) / iters * 1e3  # us


def main():
    from musev_amd import ops
    from kernel_cases import ALL_CASES, AT_SIZE_CASES
    for name, fn in list(ALL_CASES) + list(AT_SIZE_CASES):
        if "temporal" in name or "groupnorm" in name or "layernorm" in name:
            r = fn()
            torch.cuda.synchronize()
            print(f"{'PASS' if r['ok'] else 'FAIL'} {name} err={r.get('max_abs_err')}", flush=True)
    b, t, heads = 2, 13, 8
    for hw, d in ((4096, 40), (1024, 80), (256, 160), (64, 160)):
        c = heads * d
        qkv = torch.randn(b * t * hw, 3 * c, device="cuda").half()
        us = timeit(lambda: ops.temporal_attention(qkv[:, :c], qkv[:, c:2 * c], qkv[:, 2 * c:], b, t, hw, heads, d, d ** -0.5))
        mb = b * t * hw * 4 * c * 2 / 1e6
        print(f"temporal_attention hw{hw} d{d}: {us:.1f} us  {mb:.0f} MB  {mb / us * 1e-3 * 1e3:.2f} TB/s".replace("TB/s", "GB/ms"), flush=True)
    for n, rows, c in ((26, 4096, 320), (2, 13 * 4096, 320), (26, 1024, 640), (2, 13 * 1024, 640), (26, 256, 1280), (2, 13 * 256, 1280),
                       (26, 64, 1280), (2, 13 * 64, 1280), (26, 64, 2560), (26, 256, 2560), (26, 256, 1920)):
        x = torch.randn(n * rows, c, device="cuda").half()
        gm, bt = torch.ones(c, device="cuda").half(), torch.zeros(c, device="cuda").half()
        us = timeit(lambda: ops.groupnorm(x, gm, bt, n, rows, eps=1e-5, silu=True))
        mb = n * rows * c * 2 * 3 / 1e6
        print(f"groupnorm n{n} rows{rows} c{c}: {us:.1f} us  {mb:.1f} MB (read twice + write)  {mb / us:.2f} TB/s", flush=True)
    for rows, c in ((26 * 4096, 320), (13 * 4096, 320), (26 * 1024, 640), (13 * 1024, 640), (26 * 256, 1280), (13 * 256, 1280), (26 * 64, 1280)):
        x = torch.randn(rows, c, device="cuda").half()
        gm, bt = torch.ones(c, device="cuda").half(), torch.zeros(c, device="cuda").half()
        us = timeit(lambda: ops.layernorm(x, gm, bt, 1e-5))
        mb = rows * c * 2 * 2 / 1e6
        print(f"layernorm rows{rows} c{c}: {us:.1f} us  {mb:.1f} MB  {mb / us:.2f} TB/s", flush=True)


if __name__ == "__main__":
    main()
