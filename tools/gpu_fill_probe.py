#!/usr/bin/env python3
"""Grid-fill probe for the big implicit-GEMM tiles (run on the MI355X):

    python tools/gpu_fill_probe.py [cfg ...]

Every M of the step is 13 * 2^k rows (12 frames + 1 condition frame), so a 256-row tile covers a level in 208 / 52 / 13 row blocks:
the 256 x 320 tile launches 208 workgroups on 256 CUs at level 0 and 104 at level 1.  This probe times the same convolution /
projection at 13 and at 16 frames (16 frames = 256 / 64 row blocks: whole rounds of the CUs), alone and as a concurrent pair on two
streams: if a launch takes the same time at 13 and at 16 frames, the tile is paced by ROUNDS of workgroups (per-CU rate) and the
13-frame launches leave 19 % of the chip idle; if the time scales with the rows, the chip-wide operand stream paces it."""
from __future__ import annotations

import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from musev_amd import ops  # noqa: E402


def timed(fn, reps=50):
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
    frames_list = (13, 16)
    if "--13" in sys.argv:
        sys.argv.remove("--13")
        frames_list = (13,)
    cfgs = [int(a) for a in sys.argv[1:]] or [6]
    dev = torch.device("cuda:0")
    side = torch.cuda.Stream()
    main_s = torch.cuda.current_stream()
    g = torch.Generator(device="cpu").manual_seed(0)
    cases = [("conv3x3", 64, 320, 320), ("conv3x3", 32, 640, 640), ("conv3x3", 32, 1280, 640), ("tconv3", 64, 320, 320), ("tconv3", 32, 640, 640),
             ("geglu", 32, 640, 5120), ("geglu", 16, 1280, 10240), ("linear", 32, 640, 1920),
             ("conv3x3", 16, 1280, 1280), ("tconv3", 16, 1280, 1280), ("linear", 16, 1280, 3840), ("linear", 16, 5120, 1280), ("linear", 32, 2560, 640)]
    for cfg in cfgs:
        ops.GEMM_CFG = cfg
        for kind, hw, cin, cout in cases:
            row = []
            for frames in frames_list:
                M = frames * hw * hw
                taps = 9 if kind == "conv3x3" else 3 if kind == "tconv3" else 1
                xs = [(torch.randn(M, cin, generator=g) * 0.5).half().to(dev) for _ in range(2)]
                w = (torch.randn(cout, taps * cin, generator=g) * 0.02).half().to(dev)
                outs = [torch.empty(M, cout // 2 if kind == "geglu" else cout, dtype=torch.float16, device=dev) for _ in range(2)]

                def one(i):
                    if kind == "conv3x3":
                        ops.conv3x3(xs[i], w, frames, hw, hw, out=outs[i])
                    elif kind == "tconv3":
                        ops.tconv3(xs[i], w, 1, frames, hw * hw, out=outs[i])
                    else:
                        ops.gemm(xs[i], w, geglu=(kind == "geglu"), out=outs[i])

                def pair():
                    side.wait_stream(main_s)
                    with torch.cuda.stream(side):
                        one(1)
                    one(0)
                    main_s.wait_stream(side)

                t1 = timed(lambda: one(0))
                t2 = timed(pair) / 2
                fl = 2.0 * M * cout * taps * cin
                row.append((frames, t1, fl / t1 * 1e-6, t2, fl / t2 * 1e-6))
            if len(row) == 1:
                f0, a1, r1, a2, r2 = row[0]
                print(f"cfg {cfg:2d} {kind:8s} hw {hw:2d} {cin:4d}->{cout:5d}: 13 frames alone {a1:7.1f} us {r1:6.0f} TF/s, in a pair {a2:7.1f} us {r2:6.0f} TF/s", flush=True)
                continue
            (f0, a1, r1, a2, r2), (f1, b1, s1, b2, s2) = row
            print(f"cfg {cfg:2d} {kind:8s} hw {hw:2d} {cin:4d}->{cout:5d}: 13 frames alone {a1:7.1f} us {r1:6.0f} TF/s, in a pair {a2:7.1f} us {r2:6.0f} TF/s | "
                  f"16 frames alone {b1:7.1f} us {s1:6.0f} TF/s, in a pair {b2:7.1f} us {s2:6.0f} TF/s | time ratio 16/13: alone {b1 / a1:.3f}, pair {b2 / a2:.3f} (rows 1.231)",
                  flush=True)


if __name__ == "__main__":
    main()
