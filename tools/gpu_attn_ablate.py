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
    "no_dma": [("        if (t + 2 < total) issue(t + 2);", "        if (t + 2 < total && t < 1) issue(t + 2);"),
               ("            if (npw == 1) asm volatile(\"s_waitcnt vmcnt(1)\" ::: \"memory\");\n            else if (npw == 2) asm volatile(\"s_waitcnt vmcnt(2)\" ::: \"memory\");\n            else if (npw == 3) asm volatile(\"s_waitcnt vmcnt(3)\" ::: \"memory\");\n            else asm volatile(\"s_waitcnt vmcnt(5)\" ::: \"memory\");", "            asm volatile(\"s_waitcnt vmcnt(0)\" ::: \"memory\");")],
    # the P.V product replaced by one VALU add per accumulator (keeps the dependence on P and on the V fragment read)
    "no_pv": [("                    acc_o[qt][dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf, pfrag[qt][cc], acc_o[qt][dt], 0, 0, 0);",
               "                    acc_o[qt][dt][0] += (float)pfrag[qt][cc][0] + (float)vf[0];")],
    # the 32-deep score products replaced by one VALU add each (the 16-deep tail stays)
    "no_qk32": [("                    acc_s[qt][st] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf, qf[qt][c], acc_s[qt][st], 0, 0, 0);",
                 "                    acc_s[qt][st][0] += (float)kf[0] * (float)qf[qt][c][0];")],
    "no_exp_no_max": [("acc_s[qt][st][r] = __builtin_amdgcn_exp2f(acc_s[qt][st][r]);", "asm volatile(\"\" : \"+v\"(acc_s[qt][st][r]));"),
                      ("if (first || __any(fmaxf(lmx[0], lmx[1]) > kThr)) {", "if (first) {")],
    # round 6: the MFMA order around the 16-deep tail is pinned by sched_barriers; this variant removes them
    "no_pin": [("__builtin_amdgcn_sched_barrier(0x7F6);", ";")],
    "pv_pin": [("                    acc_o[qt][dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf, pfrag[qt][cc], acc_o[qt][dt], 0, 0, 0);\n            }\n        }\n        if constexpr ((VAR & 32) != 0) __builtin_amdgcn_s_setprio(0);\n        first = false;",
                "                    acc_o[qt][dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf, pfrag[qt][cc], acc_o[qt][dt], 0, 0, 0);\n                __builtin_amdgcn_sched_barrier(0x7F6);\n            }\n        }\n        if constexpr ((VAR & 32) != 0) __builtin_amdgcn_s_setprio(0);\n        first = false;")],
    "var43": [("constexpr int kAttnVar40 = 47;", "constexpr int kAttnVar40 = 43;")],
    "var46": [("constexpr int kAttnVar40 = 47;", "constexpr int kAttnVar40 = 46;")],
    "var39": [("constexpr int kAttnVar40 = 47;", "constexpr int kAttnVar40 = 39;")],
    "prio3": [("if constexpr ((VAR & 32) != 0) __builtin_amdgcn_s_setprio(1);", "if constexpr ((VAR & 32) != 0) __builtin_amdgcn_s_setprio(3);")],
    "no_setprio": [("if constexpr ((VAR & 32) != 0) __builtin_amdgcn_s_setprio(1);", ""), ("if constexpr ((VAR & 32) != 0) __builtin_amdgcn_s_setprio(0);", "")],
}


VARIANTS4 = {   # ABL_SET=attn4: the software-pipelined kernel (mv_attention_f16_var, ABL_VAR40 = 100 ...)
    "base": [],
    "no_sched": [("#ifndef MV_ATTN4_NO_SCHED", "#if 0")],
    "no_exp": [("cur[qt][st][r] = __builtin_amdgcn_exp2f(cur[qt][st][r]);", "asm volatile(\"\" : \"+v\"(cur[qt][st][r]));")],
    "no_barrier": [("            __builtin_amdgcn_s_barrier();   // tile t+1 visible; every wave has left iteration t-1: the stage of tile t-2 is free", "")],
    "no_dma": [("            if (t + C::AHEAD < total) issue(t + C::AHEAD);", "")],
    "no_pv": [("            pv((t + NST - 1) % NST);", "")],
    "no_scores": [("            scores((t + 1) % NST, nxt);", "#pragma unroll\n            for (int qt = 0; qt < C::QT; ++qt)\n#pragma unroll\n                for (int st = 0; st < 4; ++st) nxt[qt][st] = cur[qt][st] * 0.5f;")],
    "no_max": [("            if (first || __any(fmaxf(lmx[0], lmx[1]) > kThr)) {\n#pragma unroll\n                for (int qt = 0; qt < C::QT; ++qt) {\n                    float mx = lmx[qt];\n                    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));\n                    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));\n                    const float delta = first ? mx : fmaxf(mx, 0.f);\n                    m_ref[qt] += delta;\n                    const float alpha = __builtin_amdgcn_exp2f(-delta);\n#pragma unroll\n                    for (int dt = 0; dt < C::NDT; ++dt) acc_o[qt][dt] *= alpha;\n#pragma unroll\n                    for (int st = 0; st < 4; ++st) cur[qt][st] -= delta;",
                "            if (first) {\n#pragma unroll\n                for (int qt = 0; qt < C::QT; ++qt) {\n                    float mx = lmx[qt];\n                    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));\n                    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));\n                    const float delta = first ? mx : fmaxf(mx, 0.f);\n                    m_ref[qt] += delta;\n                    const float alpha = __builtin_amdgcn_exp2f(-delta);\n#pragma unroll\n                    for (int dt = 0; dt < C::NDT; ++dt) acc_o[qt][dt] *= alpha;\n#pragma unroll\n                    for (int st = 0; st < 4; ++st) cur[qt][st] -= delta;")],
    "no_wait": [("            if (t + C::AHEAD <= total) wait_steady();   // tiles 0 .. t + AHEAD - 1 are issued\n            else asm volatile(\"s_waitcnt vmcnt(0)\" ::: \"memory\");", "")],
}
SET = os.environ.get("ABL_SET", "attn3")
VAR40 = int(os.environ.get("ABL_VAR40", "-1"))
ITERS = int(os.environ.get("ABL_ITERS", "100"))
ONLY = [x for x in os.environ.get("ABL_ONLY", "").split(",") if x]
if SET == "attn4":
    VARIANTS = VARIANTS4


def build(name, edits):
    text = SRC.replace('#include "common.h"', '#include "../../../musev_amd/csrc/common.h"')   # (relative to tools/scratch/abl/: the same text on every machine)
    for old, new in edits:
        assert old in text, (name, old[:60])
        text = text.replace(old, new)   # (every occurrence)
    # built HERE (hipcc cross-compiles; `--build-only`) into tools/scratch/abl/, which travels to the GPU box with the snapshot -- a
    # variant compiles for ~80 s, which is better spent on the build container than on GPU minutes
    out_dir = os.path.join(ROOT, "tools", "scratch", "abl")
    os.makedirs(out_dir, exist_ok=True)
    src = os.path.join(out_dir, f"attn_abl_{SET}_{name}.hip")
    so = os.path.join(out_dir, f"libattn_abl_{SET}_{name}.so")
    if os.path.exists(so) and os.path.exists(src) and open(src).read() == text:
        return so
    open(src, "w").write(text)
    r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=fast", "-shared", "-DMV_EXPERIMENT",
                        "-o", so, src, os.path.join(CSRC, "lib.hip")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    return so


def main():
    if "--build-only" in sys.argv:
        import concurrent.futures
        with concurrent.futures.ThreadPoolExecutor(max_workers=int(os.environ.get("ABL_JOBS", "3"))) as ex:
            for name, so in zip(VARIANTS, ex.map(lambda kv: build(*kv), VARIANTS.items())):
                print("built", name, so, flush=True)
        return
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
    for name, edits in [kv for kv in VARIANTS.items() if not ONLY or kv[0] in ONLY] * int(os.environ.get('ABL_ROUNDS', '1')):
        lib = C.CDLL(build(name, edits))
        lib.mv_attention_f16_var.restype = C.c_int32
        lib.mv_attention_f16_var.argtypes = [C.POINTER(_lib.AttnDesc), C.c_int32, C.c_int32, C.c_void_p]
        lib.mv_attention_f16 = lambda d_, st_, _l=lib: _l.mv_attention_f16_var(d_, VAR40, -1, st_)
        st = torch.cuda.current_stream().cuda_stream
        for _ in range(10):
            assert lib.mv_attention_f16(C.byref(ds), st) == 0
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(ITERS):
            lib.mv_attention_f16(C.byref(ds), st)
        e1.record()
        torch.cuda.synchronize()
        print(f"{SET} var {VAR40} {name:12s} {e0.elapsed_time(e1) / ITERS:.3f} ms", flush=True)


if __name__ == "__main__":
    main()
