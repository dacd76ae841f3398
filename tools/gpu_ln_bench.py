"""LayerNorm launch variants on the config-2 shapes, in the two regimes the step has: `cold` = inputs cycled through more distinct
buffers than the 256 MB infinity cache holds, `warm` = each input rewritten by a copy kernel right before it is normalised (what a
producer GEMM leaves behind).  Needs an experiment build of the library:
    MV_LIB_NAME=libmusev_hip_exp.so MV_EXTRA_FLAGS=-DMV_EXPERIMENT bash musev_amd/csrc/build.sh
    MUSEV_HIP_LIBRARY=musev_amd/csrc/libmusev_hip_exp.so python tools/gpu_ln_bench.py"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def main():
    from musev_amd import _lib
    lib = _lib.load()
    fn = lib.mv_layernorm_f16_var  # (only an experiment build exports it)
    fn.restype = C.c_int32
    fn.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_float, C.c_int32, C.c_void_p]
    st = torch.cuda.current_stream().cuda_stream
    for rows, c in ((13 * 4096, 320), (13 * 1024, 640), (13 * 256, 1280), (26 * 4096, 320)):
        nbuf = max(2, int(1.2e9 // (rows * c * 2)))
        nbuf = min(nbuf, 24)
        xs = [torch.randn(rows, c, device="cuda").half() for _ in range(nbuf)]
        ys = [torch.empty_like(xs[0]) for _ in range(4)]
        src = torch.randn(rows, c, device="cuda").half()
        gm, bt = (torch.ones(c, device="cuda") + 0.1 * torch.randn(c, device="cuda")).half(), (0.1 * torch.randn(c, device="cuda")).half()
        ref = torch.nn.functional.layer_norm(xs[0].float(), (c,), gm.float(), bt.float(), 1e-5)
        for var in (0, 1, 2, 4, 8, 16):   # rows per wave (0 = the product's choice)
            if var in (8, 16) and c > 512:
                continue

            def run(i, var=var):
                rc = fn(xs[i % nbuf].data_ptr(), c, ys[i % 4].data_ptr(), c, rows, c, gm.data_ptr(), bt.data_ptr(), 1e-5, var, st)
                assert rc == 0, rc
            xs[0].copy_(src)
            run(0)
            err = float((ys[0].float() - torch.nn.functional.layer_norm(src.float(), (c,), gm.float(), bt.float(), 1e-5)).abs().max())
            res = []
            for warm in (False, True):
                for i in range(3):
                    run(i)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                n = 48
                tot = 0.0
                if not warm:
                    e0.record()
                    for i in range(n):
                        run(i)
                    e1.record()
                    torch.cuda.synchronize()
                    tot = e0.elapsed_time(e1) / n * 1e3
                else:
                    # copy + layernorm pairs minus copies alone
                    e0.record()
                    for i in range(n):
                        xs[i % nbuf].copy_(src)
                        run(i)
                    e1.record()
                    torch.cuda.synchronize()
                    both = e0.elapsed_time(e1)
                    e0.record()
                    for i in range(n):
                        xs[i % nbuf].copy_(src)
                    e1.record()
                    torch.cuda.synchronize()
                    tot = (both - e0.elapsed_time(e1)) / n * 1e3
                res.append(tot)
            mb = rows * c * 4 / 1e6
            print(f"layernorm rows{rows} c{c} var{var}: cold {res[0]:6.1f} us ({mb / res[0]:.2f} TB/s)  after-producer {res[1]:6.1f} us ({mb / res[1]:.2f} TB/s)  err {err:.1e}", flush=True)


if __name__ == "__main__":
    main()
