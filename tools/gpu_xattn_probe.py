"""Structured-operand probe of the resident-K/V cross-attention on hardware (V = 1, K = 0, V = column / key index, one dominant key,
row r attends key r, random vs the tiled kernel): the printout says WHICH stage or index mapping is off.  Written for round 5's
first GPU run of the kernel (green on the host simulator, wrong on the MI355X: profiles/r05b_debug_unproven.log ... r05e): it
located a v_mfma_f32_16x16x16_f16 that accumulated onto the result of the v_mfma_f32_16x16x32_f16 issued right in front of it and
read a half-written accumulator.   python tools/gpu_xattn_probe.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from musev_amd import ops  # noqa: E402
ops.XATTN_RESIDENT_MAX_D = 80   # (the model's default is 40 since round 6: this tool measures the kernel at both head dims)

torch.set_printoptions(precision=3, linewidth=220, sci_mode=False)
dev = torch.device("cuda", 0)


def run(q, k, v, nb, lq, lk, heads, d, resident):
    flag = ops.XATTN_RESIDENT
    try:
        ops.XATTN_RESIDENT = resident
        return ops.attention(q, [(k, v, lk, 1, 1, 0)], nb, lq, heads, d, d ** -0.5).float()
    finally:
        ops.XATTN_RESIDENT = flag


def probe(d=40, heads=8, lq=16, lk=16):
    c = heads * d
    g = torch.Generator().manual_seed(1)
    qr = (torch.randn(lq, c, generator=g)).half().to(dev)
    kr = (torch.randn(lk, c, generator=g)).half().to(dev)
    vr = (torch.randn(lk, c, generator=g)).half().to(dev)
    zero_k = torch.zeros(lk, c, dtype=torch.float16, device=dev)
    print(f"==== d {d} heads {heads} lq {lq} lk {lk}")
    # A: V = 1 -> out = 1
    o = run(qr, kr, torch.ones_like(vr), 1, lq, lk, heads, d, True)
    print("A  V = 1 (out must be 1): min %.3f max %.3f" % (o.min().item(), o.max().item()))
    # B: K = 0 -> out = column mean of V
    o = run(qr, zero_k, vr, 1, lq, lk, heads, d, True)
    want = vr.float().mean(0, keepdim=True).expand(lq, c)
    print("B  K = 0, V random (out = mean over keys): max err %.4f" % (o - want).abs().max().item())
    # C: V[key, col] = col (K = 0) -> out[:, col] = col: reveals a column permutation
    vc = torch.arange(c, dtype=torch.float32).repeat(lk, 1).half().to(dev)
    o = run(qr, zero_k, vc, 1, lq, lk, heads, d, True)
    bad = (o[0] - vc[0].float()).abs() > 0.5
    print("C  V = column index: wrong columns %d of %d; out row 0, head 0: %s" % (int(bad.sum()), c, o[0, :d].tolist()))
    if heads > 1:
        print("   head 1:", o[0, d:2 * d].tolist())
    # D: V[key, col] = key (K = 0) -> mean = (lk - 1) / 2
    vk = torch.arange(lk, dtype=torch.float32)[:, None].repeat(1, c).half().to(dev)
    o = run(qr, zero_k, vk, 1, lq, lk, heads, d, True)
    print("D  V = key index: out must be %.2f: min %.3f max %.3f" % ((lk - 1) / 2, o.min().item(), o.max().item()))
    # E: a dominant key j per head: K[j, head h columns] = 20 * sign pattern, q = same pattern -> out = V[j_h]
    qe = torch.ones(lq, c).half().to(dev)
    ke = torch.zeros(lk, c)
    for h in range(heads):
        ke[(3 * h + 1) % lk, h * d:(h + 1) * d] = 8.0
    o = run(qe, ke.half().to(dev), vk, 1, lq, lk, heads, d, True)
    print("E  dominant key (3h+1)%%lk per head, V = key index: out row 0 per head:", [round(o[0, h * d].item(), 2) for h in range(heads)],
          "want", [(3 * h + 1) % lk for h in range(heads)])
    # F: scores depend on the query row: q[r] = r-th unit pattern selecting key r
    #    K[j, :] = e_j scaled in the first 16 columns of each head; q[r, :] = 30 * e_r -> row r attends key r -> out[r] = V[r] = r
    kf = torch.zeros(lk, c)
    qf = torch.zeros(lq, c)
    for h in range(heads):
        for j in range(min(lk, 16)):
            kf[j, h * d + j] = 6.0
        for r in range(min(lq, 16)):
            qf[r, h * d + r] = 30.0
    o = run(qf.half().to(dev), kf.half().to(dev), vk, 1, lq, lk, heads, d, True)
    print("F  row r attends key r: out[:, 0] =", [round(x, 2) for x in o[:, 0].tolist()])
    # G: random vs tiled
    o1 = run(qr, kr, vr, 1, lq, lk, heads, d, True)
    o0 = run(qr, kr, vr, 1, lq, lk, heads, d, False)
    err = (o1 - o0).abs()
    print("G  random: max |resident - tiled| %.4f; per head max:" % err.max().item(), [round(err[:, h * d:(h + 1) * d].max().item(), 3) for h in range(heads)])
    print("   per row max:", [round(x, 3) for x in err.max(1).values.tolist()])
    print("   per column (head 0) max:", [round(x, 3) for x in err[:, :d].max(0).values.tolist()])


if __name__ == "__main__":
    probe(40, 8, 16, 16)
    probe(40, 8, 32, 77)
    probe(80, 8, 16, 16)
    probe(40, 5, 16, 16)
