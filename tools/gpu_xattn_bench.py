"""Text (+ image-prompt) cross-attention: the resident-K/V kernel (mv_attn_desc.resident_kv, ops.XATTN_RESIDENT) against the tiled
kernel it would replace, on the level-0 / level-1 shapes of config 2 (one CFG half: 13 frames) and config 3 (IP-Adapter tokens as a
second softmax group), plus the GroupNorm fold inside the apply pass on the level-0 per-frame norm (run on the MI355X):

    python tools/gpu_xattn_bench.py

Prints microseconds per launch, the effective HBM rate over the algorithmic bytes (q read once + out written once) and the
max |difference| between the two kernels."""
import math
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from musev_amd import ops  # noqa: E402
ops.XATTN_RESIDENT_MAX_D = 80   # (the model's default is 40 since round 6: this tool measures the kernel at both head dims)

dev = torch.device("cuda", 0)


def rnd(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(torch.float16).to(dev)


def timed(fn, reps=30):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    heads, lk = 8, 77
    for d, lq, nb, ip in ((40, 4096, 13, False), (40, 4096, 13, True), (40, 4096, 26, False), (80, 1024, 13, False), (80, 1024, 13, True),
                          (40, 9216, 13, True)):
        c = heads * d
        qs = [rnd((nb * lq, c), 20 + i) for i in range(4)]   # cycled: the query tensor of a launch comes from HBM, not from the L2s
        kv = rnd((lk, 2 * c), 30)
        kvi = rnd((4, 2 * c), 31)
        segs = [(kv[:, :c], kv[:, c:], lk, nb, 1, 0)]
        gs = None
        if ip:
            segs.append((kvi[:, :c], kvi[:, c:], 4, nb, 1, 0))
            gs = [1.0, 0.8]
        outs = [torch.empty_like(qs[0]) for _ in range(2)]
        k = [0]
        res = {}
        sweep = {}
        for resident in (0, 1, 64, 128, 256, 512):   # 0: tiled; 1: the launcher's rows per block; else that many rows per block
            ops.XATTN_RESIDENT = resident
            hits = ops.XATTN_RESIDENT_HITS

            def run():
                k[0] += 1
                ops.attention(qs[k[0] % 4], segs, nb, lq, heads, d, d ** -0.5, out=outs[int(resident == 1)], group_scales=gs)
            us = timed(run)
            if resident and ops.XATTN_RESIDENT_HITS == hits:
                print(f"d {d} lq {lq} nb {nb}: the resident kernel did not take the launch")
                continue
            k[0] = 0
            run()
            if resident > 1:
                sweep[resident] = round(us, 1)
            else:
                res[bool(resident)] = us
        torch.cuda.synchronize()
        ops.XATTN_RESIDENT = 0
        mb = 2 * nb * lq * c * 2 / 1e6
        diff = (outs[0].float() - outs[1].float()).abs().max().item()
        print(f"cross-attention d {d} rows {nb} x {lq} keys {lk}{' + 4 (group)' if ip else ''}: tiled {res.get(False, float('nan')):7.1f} us "
              f"({mb / res.get(False, math.nan) :5.2f} TB/s)   resident {res.get(True, float('nan')):7.1f} us ({mb / res.get(True, math.nan):5.2f} TB/s)"
              f"   max |diff| {diff:.2e}   rows per block -> us: {sweep}")


if __name__ == "__main__":
    main()
