#!/usr/bin/env python3
"""The text cross-attention sub-block of level 0 as one launch against its three-launch form (run on the MI355X):

    python tools/gpu_xab_bench.py [frames ...]

times mv_xattn_block_f16 and [LayerNorm-folded to_q projection, resident-K/V cross-attention, to_out + residual] on the config-2 level-0
shapes (frames x 64 x 64 rows, C = 320, 77 keys), alone and as a concurrent pair on two streams, and checks the two forms against each other."""
from __future__ import annotations

import math
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from musev_amd import ops  # noqa: E402


def timed(fn, reps=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    dev = torch.device("cuda:0")
    side = torch.cuda.Stream()
    main_s = torch.cuda.current_stream()
    g = torch.Generator().manual_seed(0)
    c, heads, d, keys, hw = 320, 8, 40, 77, 4096
    for frames in [int(a) for a in sys.argv[1:]] or [13, 26]:
        M = frames * hw
        xs = [torch.randn(M, c, generator=g).half().to(dev) for _ in range(2)]
        gamma, beta = (torch.randn(c, generator=g) * 0.2 + 1).half().to(dev), (torch.randn(c, generator=g) * 0.2).half().to(dev)
        wq, wo = [(torch.randn(c, c, generator=g) / math.sqrt(c)).half().to(dev) for _ in range(2)]
        bo = (torch.randn(c, generator=g) * 0.3).half().to(dev)
        kv = torch.randn(keys, 2 * c, generator=g).half().to(dev)
        k, v = kv[:, :c], kv[:, c:]
        wq_p, wo_p = ops.pack_xab_q(wq, heads, d), ops.pack_tsa_out(wo, heads, d)
        scale = d ** -0.5
        outs = [torch.empty(M, c, dtype=torch.float16, device=dev) for _ in range(2)]

        def fused(i):
            ops.xattn_block(xs[i], gamma, beta, 1e-5, wq_p, k, v, keys, M, wo_p, bo, heads, d, scale, out=outs[i])

        def three(i):
            xh = ops.layernorm(xs[i], gamma, beta, 1e-5)
            q = ops.gemm(xh, wq)
            att = ops.attention(q, [(k, v, keys, frames, 1, 0)], frames, hw, heads, d, scale)
            ops.gemm(att, wo, bias=bo, residual=xs[i], out=outs[i])

        def pair(fn):
            def run():
                side.wait_stream(main_s)
                with torch.cuda.stream(side):
                    fn(1)
                fn(0)
                main_s.wait_stream(side)
            return run

        fused(0)
        a = outs[0].float().clone()
        three(0)
        b = outs[0].float().clone()
        torch.cuda.synchronize()
        print(f"{frames} frames: |fused - three launches|max {float((a - b).abs().max()):.2e}  | fused {timed(lambda: fused(0)):7.1f} us alone, {timed(pair(fused)) / 2:7.1f} us per launch of a pair | "
              f"LayerNorm + to_q + attention + to_out {timed(lambda: three(0)):7.1f} us alone, {timed(pair(three)) / 2:7.1f} us per launch of a pair", flush=True)


if __name__ == "__main__":
    main()
