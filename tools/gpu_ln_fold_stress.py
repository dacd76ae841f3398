"""Determinism / parity stress of the LayerNorm-folded GEMM: 30 launches per shape, every run compared bit for bit with run 0 and with a
torch fp32 reference (the tool that located the packed-fp32 op_sel failure of round 3, profiles/r03d, r03e).
Usage: python tools/gpu_ln_fold_stress.py [extra libmusev_hip builds ...]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.nn.functional as F
from musev_amd import _lib, ops
prod = _lib.load()
libs = [("product", prod)]
for path in sys.argv[1:]:
    lib = C.CDLL(path)
    for name, (res, args) in _lib.SIGNATURES.items():
        fn = getattr(lib, name); fn.restype, fn.argtypes = res, args
    libs.append((os.path.basename(path), lib))
g = torch.Generator(device="cuda").manual_seed(0)
for (M, N, K, geglu) in [(106496, 320, 320, False), (53248, 320, 320, False), (26624, 5120, 640, True), (3328, 10240, 1280, True), (53248, 960, 320, False)]:
    x = torch.randn(M, K, device="cuda", generator=g).half()
    gamma = (1 + 0.1 * torch.randn(K, device="cuda", generator=g)).half()
    beta = (0.1 * torch.randn(K, device="cuda", generator=g)).half()
    w = (torch.randn(N, K, device="cuda", generator=g) / K ** 0.5).half()
    b = (0.1 * torch.randn(N, device="cuda", generator=g)).half()
    yref = F.layer_norm(x.float(), (K,), gamma.float(), beta.float(), 1e-5)
    ref = yref @ w.float().t() + b.float()
    if geglu:
        ref = ref[:, :N // 2] * F.gelu(ref[:, N // 2:])
        wp, bp = ops.pack_geglu(w, b)
    else:
        wp, bp = w, b
    wf, cs, cb = ops.fold_layernorm(wp, bp, gamma, beta)
    for name, lib in libs:
        _lib._lib = lib
        for kind in ("plain", "fold"):
            outs = []
            for it in range(30):
                if kind == "plain":
                    o = ops.gemm(ops.layernorm(x, gamma, beta, 1e-5), wp, bias=bp, geglu=geglu)
                else:
                    o = ops.gemm(x, wf, ln=(cs, cb, 1e-5), geglu=geglu)
                outs.append(o)
            torch.cuda.synchronize()
            bad = [i for i, o in enumerate(outs) if not torch.equal(o, outs[0])]
            errs = [(o.float() - ref).abs().max().item() for o in outs]
            worst = max(range(30), key=lambda i: errs[i])
            msg = f"M{M} N{N} K{K} geglu={geglu} {name} {kind}: runs differing from run 0: {len(bad)}; max err vs fp32 {max(errs):.2e} (min {min(errs):.2e})"
            if max(errs) > 0.05:
                d = (outs[worst].float() - ref).abs()
                rows = (d.max(dim=1).values > 0.05).nonzero().flatten()
                cols = (d.max(dim=0).values > 0.05).nonzero().flatten()
                msg += f" | bad rows {rows.numel()} [{rows[:6].tolist()}..{rows[-3:].tolist()}] bad cols {cols.numel()} [{cols[:4].tolist()}..{cols[-2:].tolist()}]"
            print(msg, flush=True)
    _lib._lib = prod
