"""Ablation of the attention kernel (cdna_hip_programming.md 5.4: ablate before optimising): text-level variants of attention.hip are
compiled ON THE GPU BOX into private shared libraries (the product library and its source are not touched) and timed on the level-0
self-attention.  Variants remove one phase each (results are wrong by construction; timing only)."""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

CSRC = os.path.join(ROOT, "musev_amd", "csrc")
SRC = open(os.path.join(CSRC, "attention.hip")).read()

VARIANTS = {
    "base": [],
    "no_exp": [("acc_s[qt][st][r] = __builtin_amdgcn_exp2f(acc_s[qt][st][r]);", "asm volatile(\"\" : \"+v\"(acc_s[qt][st][r]));")],
    "no_barrier": [("        __builtin_amdgcn_s_barrier();  // every wave's pieces of tile t are visible", "        // (ablated barrier)  // every wave's pieces of tile t are visible")],
    "no_max": [("if (first || __any(fmaxf(lmx[0], lmx[1]) > kThr)) {", "if (first) {")],
    "no_rowsum": [("acc_l[qt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ones_f, pfrag[qt][cc], acc_l[qt], 0, 0, 0);", "acc_l[qt][0] += (float)pfrag[qt][cc][0];")],
    "no_dma": [("        if (t + 2 < total) issue(t + 2);", "        if (t + 2 < total && t < 1) issue(t + 2);"),
               ("            if (npw == 2) asm volatile(\"s_waitcnt vmcnt(2)\" ::: \"memory\");\n            else if (npw == 3) asm volatile(\"s_waitcnt vmcnt(3)\" ::: \"memory\");\n            else asm volatile(\"s_waitcnt vmcnt(5)\" ::: \"memory\");", "            asm volatile(\"s_waitcnt vmcnt(0)\" ::: \"memory\");")],
    "waves2": [("__launch_bounds__(256, D == 40 ? 3 : 2) void attn3_kernel", "__launch_bounds__(256, 2) void attn3_kernel")],
}


def build(name, edits):
    text = SRC.replace('#include "common.h"', '#include "%s"' % os.path.join(CSRC, "common.h"))
    for old, new in edits:
        assert old in text, (name, old[:60])
        text = text.replace(old, new)
    src = f"/tmp/attn_abl_{name}.hip"
    so = f"/tmp/libattn_abl_{name}.so"
    open(src, "w").write(text)
    r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=fast", "-shared",
                        "-o", so, src, os.path.join(CSRC, "lib.hip")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    return so


def main():
    from musev_amd import _lib
    nb, lq, d, t, heads = 13, 4096, 40, 13, 8
    c = heads * d
    qkv = torch.randn(nb * lq, 3 * c, device="cuda").half()
    out = torch.empty(nb * lq, c, device="cuda", dtype=torch.float16)
    ds = _lib.AttnDesc()
    ds.q, ds.out, ds.ldq, ds.ldo = qkv.data_ptr(), out.data_ptr(), 3 * c, c
    ds.nb, ds.lq, ds.heads, ds.d = nb, lq, heads, d
    ds.scale, ds.nseg, ds.accumulate, ds.out_scale = d ** -0.5, 2, 0, 1.0
    for i, (div, mul, add) in enumerate(((1, 1, 0), (t, t, 0))):
        s = ds.seg[i]
        s.k, s.v, s.ldk, s.ldv, s.len, s.div, s.mul, s.add = qkv.data_ptr() + 2 * c, qkv.data_ptr() + 4 * c, 3 * c, 3 * c, lq, div, mul, add
    for name, edits in VARIANTS.items():
        lib = C.CDLL(build(name, edits))
        lib.mv_attention_f16.restype = C.c_int32
        lib.mv_attention_f16.argtypes = [C.POINTER(_lib.AttnDesc), C.c_void_p]
        st = torch.cuda.current_stream().cuda_stream
        for _ in range(2):
            assert lib.mv_attention_f16(C.byref(ds), st) == 0
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            lib.mv_attention_f16(C.byref(ds), st)
        e1.record()
        torch.cuda.synchronize()
        print(f"{name:12s} {e0.elapsed_time(e1) / 5:.3f} ms", flush=True)


if __name__ == "__main__":
    main()
