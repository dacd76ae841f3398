"""Where does a GEMM block spend its life?  Needs the library built with -DMV_TIMELINE (tools/gpu_r02l.sh does that on the GPU box):
thread 0 of every block records the 100 MHz wall clock at entry, prologue issued, first K tile landed, K loop done, epilogue issued,
stores drained.  Prints per (shape, configuration) the median phase lengths, the launch span and how densely a CU's block slots
are filled.  Usage: python tools/gpu_gemm_timeline.py"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    from musev_amd import ops, _lib
    lib = _lib.load()
    fn = lib.mv_debug_timeline
    fn.argtypes = [ctypes.c_void_p, ctypes.c_int]
    fn.restype = ctypes.c_int
    dev = "cuda"
    cases = [(106496, 960, 320, "none"), (106496, 320, 320, "res"), (106496, 2560, 320, "geglu"), (26624, 640, 640, "res"),
             (26624, 5120, 640, "geglu"), (6656, 1280, 1280, "res"), (106496, 320, 2880, "none")]
    for (M, N, K, epi) in cases:
        a = torch.randn(M, K, device=dev).half()
        w = (torch.randn(N, K, device=dev) * K ** -0.5).half()
        bias = torch.randn(N, device=dev).half()
        res = torch.randn(M, N, device=dev).half() if epi == "res" else None
        for cfg in ((7, 8, 16, 2) if epi == "geglu" else (6, 0, 16, 4)):
            ops.GEMM_CFG, ops.GEMM_SPLITK = cfg, 1

            def run():
                if epi == "geglu":
                    return ops.gemm(a, w, bias=bias, geglu=True)
                return ops.gemm(a, w, bias=bias if epi == "res" else None, residual=res)
            for _ in range(3):
                run()
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            run()
            e.record()
            torch.cuda.synchronize()
            us = s.elapsed_time(e) * 1e3
            buf = np.zeros((32768, 8), dtype=np.uint64)
            n = fn(buf.ctypes.data, 32768)
            assert n > 0
            t = buf[:, :6].astype(np.int64)
            # blocks of THIS launch: entry stamps within the last launch window (the buffer keeps older launches' rows beyond the grid)
            t0max = t[:, 0].max()
            live = (t[:, 0] > t0max - 200000) & (t[:, 5] >= t[:, 0])   # within 2 ms of the newest entry
            tt = t[live] * 0.01  # us
            hw = buf[live, 6]
            xcc = buf[live, 7] & 0xF
            cu = (hw >> 8) & 0xF
            se = (hw >> 13) & 0x7
            sh = (hw >> 12) & 0x1
            key = (xcc * 8 + se) * 32 + sh * 16 + cu
            nb = len(tt)
            ph = np.stack([tt[:, 1] - tt[:, 0], tt[:, 2] - tt[:, 1], tt[:, 3] - tt[:, 2], tt[:, 4] - tt[:, 3], tt[:, 5] - tt[:, 4], tt[:, 5] - tt[:, 0]], 1)
            med = np.median(ph, 0)
            span = tt[:, 5].max() - tt[:, 0].min()
            ncu = len(np.unique(key))
            busy = ph[:, 5].sum()
            # co-resident blocks per CU: maximum overlap count on the busiest CU
            k0 = key[0]
            sel = key == k0
            ev = sorted([(x, 1) for x in tt[sel, 0]] + [(x, -1) for x in tt[sel, 5]])
            c = mx = 0
            for _, d in ev:
                c += d
                mx = max(mx, c)
            gaps = []
            for kk in np.unique(key)[:64]:
                sel = key == kk
                st = np.sort(tt[sel, 0])
                en = np.sort(tt[sel, 5])
                if len(st) > mx:
                    gaps.extend((st[mx:] - en[:-mx]).tolist())
            gap = float(np.median(gaps)) if gaps else float("nan")
            print(f"M{M} N{N} K{K} {epi:5s} cfg{cfg:2d}: launch {us:6.1f} us, span {span:6.1f}; {nb} blocks on {ncu} CUs, {mx} co-resident; "
                  f"median us: setup {med[0]:.2f} first-tile {med[1]:.2f} k-loop {med[2]:.2f} epilogue {med[3]:.2f} drain {med[4]:.2f} "
                  f"block {med[5]:.2f}; slot refill gap {gap:.2f}; slot fill {busy / (span * ncu * mx) * 100:.0f}%", flush=True)
    ops.GEMM_CFG, ops.GEMM_SPLITK = -1, 0


if __name__ == "__main__":
    main()
