"""EXPERIMENT: does breaking the lockstep of co-resident blocks (every other first-generation block sleeps before its prologue)
let the epilogue HBM writes of one block overlap the K loop of its neighbour?  MUSEV_EXP_STAGGER=mode,count is read once per
process by the library, so the driver runs one process per setting.  Usage: python tools/gpu_stagger.py"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def worker():
    import torch
    from musev_amd import ops
    dev = "cuda"

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
        return s.elapsed_time(e) / iters * 1e3
    out_line = []
    for (M, N, K, epi) in ((106496, 960, 320, "none"), (106496, 320, 320, "res"), (106496, 2560, 320, "geglu"), (26624, 640, 640, "res"),
                           (26624, 5120, 640, "geglu")):
        a = torch.randn(M, K, device=dev).half()
        w = (torch.randn(N, K, device=dev) * K ** -0.5).half()
        bias = torch.randn(N, device=dev).half()
        res = torch.randn(M, N, device=dev).half() if epi == "res" else None
        for cfg in (0, 15, 16, 6):
            ops.GEMM_CFG, ops.GEMM_SPLITK = cfg, 1
            if epi == "geglu":
                us = timeit(lambda: ops.gemm(a, w, bias=bias, geglu=True))
            else:
                us = timeit(lambda: ops.gemm(a, w, bias=bias if epi == "res" else None, residual=res))
            out_line.append(f"{M}x{N}x{K}/{epi}/cfg{cfg}:{us:6.1f}")
    print(" ".join(out_line), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "worker":
        worker()
    else:
        for setting in ("0,0", "1,2", "1,4", "1,8", "2,2", "2,4", "2,8", "3,4"):
            env = dict(os.environ, MUSEV_EXP_STAGGER=setting)
            r = subprocess.run([sys.executable, __file__, "worker"], env=env, capture_output=True, text=True, timeout=300)
            print(f"stagger {setting}: {r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-400:]}", flush=True)
