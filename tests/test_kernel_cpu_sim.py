"""CPU (-m "not gpu"): simple HIP kernels that have not run on a GPU yet are ALSO compiled for the host and executed
thread-per-thread (tests/cpu_sim/hip_cpu_sim.h: a block = 256 real threads, __syncthreads = barrier) on small shapes, against
the same torch references the GPU cases use.  The kernel text is extracted from the .hip source, so an indexing / predicate /
staging mistake in the code that ships shows up here; what this cannot show is anything GPU-specific (alignment faults, LDS
size, occupancy) -- that stays with `-m gpu`."""
import math
import os
import re
import shutil
import subprocess

import numpy as np
import pytest
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SIM = os.path.join(ROOT, "tests", "cpu_sim")
CLANG = "/opt/rocm/lib/llvm/bin/clang++"
# The simulator runs hundreds of OS threads per block: the default CPU suite runs a representative subset of the cases below
# (about two minutes on 8 idle cores); MUSEV_SIM_FULL=1 runs all of them (every one was run when it was written).
_FULL = bool(os.environ.get("MUSEV_SIM_FULL"))


def _subset(keep: bool):
    if not _FULL and not keep:
        pytest.skip("covered by MUSEV_SIM_FULL=1")


def _extract(src_path: str, tag: str) -> str:
    src = open(src_path).read()
    m = re.search(r"// \[cpu-sim:begin %s\][^\n]*\n(.*?)// \[cpu-sim:end %s\]" % (tag, tag), src, re.S)
    assert m, f"markers for {tag} not found in {src_path}"
    return m.group(1)


@pytest.fixture(scope="module")
def conv_direct_bin(tmp_path_factory):
    if not os.path.exists(CLANG):
        pytest.skip("ROCm host clang not available")
    work = tmp_path_factory.mktemp("cpu_sim")
    body = _extract(os.path.join(ROOT, "musev_amd", "csrc", "elementwise.hip"), "conv3x3_direct")
    # dynamic LDS ("extern __shared__ ... dsw[]") becomes the per-block heap buffer of the simulator
    body, n = re.subn(r"extern __shared__ __attribute__\(\(aligned\(16\)\)\) half_t dsw\[\];[^\n]*", "dsw_DECL;", body)
    assert n == 1
    for fn in ("hip_cpu_sim.h", "conv3x3_direct_main.cpp"):
        shutil.copy(os.path.join(SIM, fn), work / fn)
    (work / "conv3x3_direct_extract.inc").write_text(body)
    exe = work / "conv3x3_direct"
    subprocess.run([CLANG, "-O1", "-std=c++17", "-pthread", "-o", str(exe), str(work / "conv3x3_direct_main.cpp")], check=True,
                   cwd=work, capture_output=True)
    return work, exe


@pytest.mark.parametrize("cin,cout,h,w,stride,act", [(3, 16, 10, 13, 1, 1), (16, 32, 10, 13, 2, 1), (16, 16, 9, 11, 2, 0),
                                                      (24, 40, 6, 5, 1, 1), (96, 8, 5, 4, 2, 1)])
def test_conv3x3_direct_kernel_on_the_host(conv_direct_bin, cin, cout, h, w, stride, act):
    work, exe = conv_direct_bin
    n = 2
    g = torch.Generator().manual_seed(cin * 100 + cout)
    x = torch.randn(n, cin, h, w, generator=g).half()
    wt = (torch.randn(cout, cin, 3, 3, generator=g) / math.sqrt(9 * cin)).half()
    b = torch.randn(cout, generator=g).half()
    ref = F.conv2d(x.float(), wt.float(), b.float(), stride=stride, padding=1)
    if act:
        ref = F.silu(ref)
    ref = ref.permute(0, 2, 3, 1).reshape(-1, cout)
    xl = x.permute(0, 2, 3, 1).reshape(-1, cin).contiguous()
    wp = wt.reshape(cout, cin, 9).permute(0, 2, 1).reshape(cout, 9 * cin).contiguous()      # tap-major, channel-minor
    with open(work / "in.bin", "wb") as f:
        f.write(xl.numpy().tobytes() + wp.numpy().tobytes() + b.numpy().tobytes())
    subprocess.run([str(exe), str(work / "in.bin"), str(work / "out.bin"), str(cin), str(cout), str(n), str(h), str(w), str(stride), str(act)],
                   check=True, timeout=300)
    got = torch.from_numpy(np.fromfile(work / "out.bin", dtype=np.float16).reshape(ref.shape)).float()
    err = (got - ref).abs()
    assert torch.isfinite(got).all()
    assert (err <= 3e-3 + 2e-3 * ref.abs()).all(), f"max err {err.max().item()}"


# ---- the implicit-GEMM kernels (MFMA + LDS-DMA) through the real mv_gemm_f16 dispatch, on the host simulator -------------
HIP_SIM = os.path.join(SIM, "hip")


def _transform_gemm_source(text: str) -> str:
    import sim_lib
    return sim_lib.transform(text)


def _build_gemm_sim(work):
    src = open(os.path.join(ROOT, "musev_amd", "csrc", "gemm.hip")).read()
    (work / "gemm_sim.inc").write_text(_transform_gemm_source(src))
    shutil.copy(os.path.join(SIM, "gemm_main.cpp"), work / "gemm_main.cpp")
    exe = work / "gemm_sim"
    r = subprocess.run([CLANG, "-O1", "-std=c++17", "-pthread", "-w", "-I", SIM, "-I", str(work), "-o", str(exe), str(work / "gemm_main.cpp")],
                       cwd=work, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    return exe


@pytest.fixture(scope="module")
def gemm_sim(tmp_path_factory):
    if not os.path.exists(CLANG):
        pytest.skip("ROCm host clang not available")
    work = tmp_path_factory.mktemp("gemm_sim")
    return work, _build_gemm_sim(work)


def _run_gemm_job(work, exe, name, tensors: dict, ints: dict, defer: int, timeout=600, env_extra=None, trace=None):
    d = work / name
    d.mkdir(exist_ok=True)
    for fn in ("a", "a2", "w", "bias", "rowbias", "residual", "alpha", "c"):
        p = d / f"{fn}.bin"
        if p.exists():
            p.unlink()
    for k, t in tensors.items():
        if t is not None:
            t.contiguous().numpy().tofile(d / f"{k}.bin")
    (d / "job.txt").write_text("".join(f"{k} {int(v)}\n" for k, v in ints.items()))
    env = dict(os.environ, SIM_DEFER=str(defer), SIM_TRACE="1", **(env_extra or {}))
    r = subprocess.run([str(exe), str(d)], capture_output=True, text=True, timeout=timeout, env=env)
    assert r.returncode == 0, (r.returncode, r.stderr[-2000:])
    if trace is not None:
        trace.append(r.stderr)
    M, ldc = ints["M"], ints["ldc"]
    return torch.from_numpy(np.fromfile(d / "c.bin", dtype=np.float16).reshape(M, ldc)).float()


def _close(got, ref, atol=4e-3, rtol=2e-3):
    err = (got - ref).abs()
    assert torch.isfinite(got).all()
    assert (err <= atol + rtol * ref.abs()).all(), f"max err {err.max().item()} (ref absmax {ref.abs().max().item()})"


def _rnd(shape, seed, scale=1.0):
    return (torch.randn(shape, generator=torch.Generator().manual_seed(seed)) * scale).half()


_BASE = dict(lda2=0, ldr=0, ldrb=0, c2=0, mode=0, stride=1, upsample=0, hin=0, win=0, hout=0, wout=0, t=0, hw=0, rows_per_group=0,
             act=0, geglu=0)


@pytest.mark.parametrize("defer", [0, 1])
def test_gemm_linear_epilogue_on_the_host(gemm_sim, defer):
    """ragged M (200 = 128 + 72), two n-tiles, K = 128 (two K steps), bias + per-group row bias + |alpha| + residual"""
    work, exe = gemm_sim
    M, N, K = 200, 320, 128
    a, w, bias, res = _rnd((M, K), 1), _rnd((N, K), 2, 1 / math.sqrt(K)), _rnd((N,), 3), _rnd((M, N), 4)
    rpg = 70
    rowbias = _rnd((3, N), 5)
    alpha = torch.tensor([-0.37])
    ref = a.float() @ w.float().t() + bias.float() + rowbias.float()[torch.arange(M) // rpg]
    ref = ref * 0.37 + res.float()
    got = _run_gemm_job(work, exe, f"lin{defer}", dict(a=a, w=w, bias=bias, rowbias=rowbias, residual=res, alpha=alpha),
                        dict(_BASE, M=M, N=N, K=K, lda=K, ldc=N, ldr=N, ldrb=N, c1=K, rows_per_group=rpg), defer)
    _close(got, ref)


def _conv_ref(x, wt, bias, stride=1, upsample=False):
    xr = x.float()
    if upsample:
        xr = F.interpolate(xr, scale_factor=2.0, mode="nearest")
    y = F.conv2d(xr, wt.float(), bias.float(), stride=stride, padding=1)
    return y.permute(0, 2, 3, 1).reshape(-1, wt.shape[0]), y.shape[2], y.shape[3]


def _pack(wt):  # [O, I, *k] -> [O, taps * I], tap-major / channel-minor (mv_pack_conv_weight_f16)
    o, i = wt.shape[0], wt.shape[1]
    return wt.reshape(o, i, -1).permute(0, 2, 1).reshape(o, -1).contiguous()


@pytest.mark.parametrize("defer", [0, 1])
@pytest.mark.parametrize("kind", ["plain", "stride2", "upsample", "two_src"])
def test_gemm_conv3x3_on_the_host(gemm_sim, kind, defer):
    """implicit-GEMM 3x3 convolution: halo predicates, tap-major K walk (9 taps x 64 channels = 9 K steps), stride 2,
    fused nearest x2 upsample, two-source channel concat, bias + per-image row bias + residual"""
    _subset((kind, defer) in (("stride2", 1), ("two_src", 1), ("upsample", 0)))
    work, exe = gemm_sim
    n, h, w, c1, c2, cout = 2, 6, 10, 64, 0, 160
    stride, up = (2, False) if kind == "stride2" else (1, kind == "upsample")
    if kind == "two_src":
        c2 = 64
    cin = c1 + c2
    x = _rnd((n, cin, h, w), 20)
    wt = _rnd((cout, cin, 3, 3), 21, 1 / math.sqrt(9 * cin))
    bias = _rnd((cout,), 22)
    ref, ho, wo = _conv_ref(x, wt, bias, stride, up)
    temb = _rnd((n, cout), 23)
    res = _rnd((n * ho * wo, cout), 24)
    ref = ref + temb.float().repeat_interleave(ho * wo, dim=0) + res.float()
    xl = x.permute(0, 2, 3, 1).reshape(n * h * w, cin).contiguous()
    tensors = dict(a=xl[:, :c1].contiguous(), w=_pack(wt), bias=bias, rowbias=temb, residual=res)
    if c2:
        tensors["a2"] = xl[:, c1:].contiguous()
    ints = dict(_BASE, M=n * ho * wo, N=cout, K=9 * cin, lda=c1, lda2=c2, ldc=cout, ldr=cout, ldrb=cout, c1=c1, c2=c2, mode=1,
                stride=stride, upsample=int(up), hin=h, win=w, hout=ho, wout=wo, rows_per_group=ho * wo)
    _close(_run_gemm_job(work, exe, f"conv_{kind}{defer}", tensors, ints, defer), ref, atol=6e-3)


@pytest.mark.parametrize("defer", [0, 1])
def test_gemm_tconv3_geglu_and_narrow_on_the_host(gemm_sim, defer):
    _subset(defer == 1)
    work, exe = gemm_sim
    # temporal conv (3,1,1): rows (b, t, p), taps walk t -+ 1 with zero padding at the clip ends; |alpha| * conv + residual
    b, t, hw, c = 2, 5, 12, 64
    x = _rnd((b, c, t, hw, 1), 30)
    wt = _rnd((c, c, 3, 1, 1), 31, 1 / math.sqrt(3 * c))
    bias = _rnd((c,), 32)
    ref = x.float() + 0.6 * F.conv3d(x.float(), wt.float(), bias.float(), padding=(1, 0, 0))
    ref = ref.permute(0, 2, 3, 4, 1).reshape(b * t * hw, c)
    xl = x.permute(0, 2, 3, 4, 1).reshape(b * t * hw, c).contiguous()
    got = _run_gemm_job(work, exe, f"tconv{defer}", dict(a=xl, w=_pack(wt), bias=bias, residual=xl, alpha=torch.tensor([0.6])),
                        dict(_BASE, M=b * t * hw, N=c, K=3 * c, lda=c, ldc=c, ldr=c, c1=c, mode=2, t=t, hw=hw), defer)
    _close(got, ref)
    # GEGLU epilogue: packed [16 value | 16 gate] weight rows, output N/2 columns
    M, C = 77, 64
    a, wf, bf = _rnd((M, C), 33), _rnd((8 * C, C), 34, 1 / math.sqrt(C)), _rnd((8 * C,), 35, 0.1)
    hfull = a.float() @ wf.float().t() + bf.float()
    ref = hfull[:, :4 * C] * F.gelu(hfull[:, 4 * C:])
    idx = torch.arange(4 * C).view(-1, 16)
    perm = torch.cat([idx, idx + 4 * C], dim=1).reshape(-1)
    got = _run_gemm_job(work, exe, f"geglu{defer}", dict(a=a, w=wf[perm].contiguous(), bias=bf[perm].contiguous()),
                        dict(_BASE, M=M, N=8 * C, K=C, lda=C, ldc=4 * C, c1=C, geglu=1), defer)
    _close(got, ref)
    # narrow epilogue: N = 36 (not a multiple of 8), ragged K = 72, SiLU
    M, N, K = 50, 36, 72
    a, w2, b2 = _rnd((M, K), 36), _rnd((N, K), 37, 1 / math.sqrt(K)), _rnd((N,), 38)
    got = _run_gemm_job(work, exe, f"narrow{defer}", dict(a=a, w=w2, bias=b2), dict(_BASE, M=M, N=N, K=K, lda=K, ldc=N, c1=K, act=1), defer)
    _close(got, F.silu(a.float() @ w2.float().t() + b2.float()))


def test_gemm_tile_order_on_the_host(gemm_sim):
    """a grid 2 x 10 tiles (wider than the group of 8, so the grouped order is active): every output element is produced
    exactly once"""
    work, exe = gemm_sim
    M, N, K = 200, 1600, 64
    a, w = _rnd((M, K), 50), _rnd((N, K), 51, 1 / math.sqrt(K))
    got = _run_gemm_job(work, exe, "order", dict(a=a, w=w), dict(_BASE, M=M, N=N, K=K, lda=K, ldc=N, c1=K), 0)
    _close(got, a.float() @ w.float().t())


@pytest.mark.parametrize("kind", ["conv_slices3_ring", "conv_two_src_slices4", "tconv_slices2", "linear_ragged_slices3", "rule", "conv_slices3_eight_waves"])
def test_gemm_split_k_on_the_host(gemm_sim, kind):
    """split-K: blockIdx.y owns a contiguous range of K tiles starting in the MIDDLE of the (tap, channel) walk; raw fp32 slabs
    + fixed-order reduce with the full epilogue.  Slices that start inside a tap, at a tap boundary and at the source switch of
    a two-source convolution; the three-stage counted ring (cfg 17) and the two-stage ring; LATEST legal LDS-DMA landing."""
    work, exe = gemm_sim
    trace = []
    if kind.startswith("conv"):
        two = "two_src" in kind
        n, h, w_, c1, c2, cout = 2, 5, 6, (64 if two else 128), (64 if two else 0), 160
        cin = c1 + c2
        x, wt, bias = _rnd((n, cin, h, w_), 130), _rnd((cout, cin, 3, 3), 131, 1 / math.sqrt(9 * cin)), _rnd((cout,), 132)
        ref, ho, wo = _conv_ref(x, wt, bias)
        temb, res = _rnd((n, cout), 133), _rnd((n * ho * wo, cout), 134)
        ref = ref + temb.float().repeat_interleave(ho * wo, dim=0) + res.float()
        xl = x.permute(0, 2, 3, 1).reshape(n * h * w_, cin).contiguous()
        tensors = dict(a=xl[:, :c1].contiguous(), w=_pack(wt), bias=bias, rowbias=temb, residual=res)
        if two:
            tensors["a2"] = xl[:, c1:].contiguous()
        # K = 9 * 128 = 18 K tiles: 3 slices of 6 (tap boundaries), 4 slices of 5 / 5 / 5 / 3 (inside a tap, at the source switch)
        ints = dict(_BASE, M=n * ho * wo, N=cout, K=9 * cin, lda=c1, lda2=c2, ldc=cout, ldr=cout, ldrb=cout, c1=c1, c2=c2, mode=1,
                    hin=h, win=w_, hout=ho, wout=wo, rows_per_group=ho * wo, splitk=(4 if two else 3), force=(1 if two else (15 if "eight_waves" in kind else 17)))
        got = _run_gemm_job(work, exe, kind, tensors, ints, 1, trace=trace)
        assert f"nsplit {4 if two else 3}" in trace[0], trace[0]
        _close(got, ref, atol=6e-3)
    elif kind.startswith("tconv"):
        b, t, hw, c = 2, 5, 12, 256
        x = _rnd((b, c, t, hw, 1), 140)
        wt, bias = _rnd((c, c, 3, 1, 1), 141, 1 / math.sqrt(3 * c)), _rnd((c,), 142)
        ref = x.float() + 0.6 * F.conv3d(x.float(), wt.float(), bias.float(), padding=(1, 0, 0))
        ref = ref.permute(0, 2, 3, 4, 1).reshape(b * t * hw, c)
        xl = x.permute(0, 2, 3, 4, 1).reshape(b * t * hw, c).contiguous()
        got = _run_gemm_job(work, exe, kind, dict(a=xl, w=_pack(wt), bias=bias, residual=xl, alpha=torch.tensor([-0.6])),
                            dict(_BASE, M=b * t * hw, N=c, K=3 * c, lda=c, ldc=c, ldr=c, c1=c, mode=2, t=t, hw=hw, splitk=2, force=3), 1,
                            trace=trace)
        assert "nsplit 2" in trace[0], trace[0]   # 12 K tiles -> 6 + 6: the second slice starts at tap 1, channel 128
        _close(got, ref)
    elif kind.startswith("linear"):
        M, N, K = 150, 36, 1048   # narrow N (8-byte epilogue contract), ragged K (17 K tiles, the last one 24 deep), SiLU
        a, w2, b2 = _rnd((M, K), 150), _rnd((N, K), 151, 1 / math.sqrt(K)), _rnd((N,), 152)
        got = _run_gemm_job(work, exe, kind, dict(a=a, w=w2, bias=b2),
                            dict(_BASE, M=M, N=N, K=K, lda=K, ldc=N, c1=K, act=1, splitk=3), 1, trace=trace)
        assert "nsplit 3" in trace[0], trace[0]
        _close(got, F.silu(a.float() @ w2.float().t() + b2.float()))
    else:
        # the library's own choice on the problem class the split exists for (8x8-latent level: small M, long K), chip of 256 CUs:
        # 64x160 tiles on the three-stage ring, 2 x 8 = 16 blocks -> 8 slices would fill the chip, K = 18 tiles caps it at 2
        n, h, w_, cin, cout = 2, 8, 8, 128, 1280
        x, wt, bias = _rnd((n, cin, h, w_), 160), _rnd((cout, cin, 3, 3), 161, 1 / math.sqrt(9 * cin)), _rnd((cout,), 162)
        ref, ho, wo = _conv_ref(x, wt, bias)
        xl = x.permute(0, 2, 3, 1).reshape(n * h * w_, cin).contiguous()
        got = _run_gemm_job(work, exe, kind, dict(a=xl, w=_pack(wt), bias=bias),
                            dict(_BASE, M=n * ho * wo, N=cout, K=9 * cin, lda=cin, ldc=cout, c1=cin, mode=1, hin=h, win=w_, hout=ho, wout=wo),
                            1, trace=trace)
        assert "choice cfg 17 nsplit 2" in trace[0], trace[0]
        _close(got, ref, atol=6e-3)


@pytest.mark.parametrize("defer", [0, 1])
def test_gemm_eight_wave_three_stage_ring_on_the_host(gemm_sim, defer):
    """the DEFAULT rule's one-round-grid kernel (8 waves, 256x160, three LDS stages behind counted s_waitcnt vmcnt(N) + raw
    s_barrier): reached here by telling the dispatch the chip has 4 CUs; K = 320 = 5 K steps through a 3-deep ring"""
    _subset(defer == 1)
    work, exe = gemm_sim
    M, N, K = 500, 320, 320
    a, w, bias, res = _rnd((M, K), 60), _rnd((N, K), 61, 1 / math.sqrt(K)), _rnd((N,), 62), _rnd((M, N), 63)
    trace = []
    got = _run_gemm_job(work, exe, f"ring{defer}", dict(a=a, w=w, bias=bias, residual=res),
                        dict(_BASE, M=M, N=N, K=K, lda=K, ldc=N, ldr=N, c1=K), defer, env_extra=dict(SIM_CUS="4"), trace=trace)
    assert "block 512" in trace[0], trace[0]
    _close(got, a.float() @ w.float().t() + bias.float() + res.float())


@pytest.mark.parametrize("kind", ["linear320", "conv320", "geglu256"])
def test_gemm_big_tiles_on_the_host(gemm_sim, kind):
    """the 256x320 (8 waves as 2x4, wave tile 128x80) and 256x256-GEGLU tiles (catalogue ids 6 / 7, what the measured table
    selects for the wide level-0 / level-1 problems)"""
    _subset(kind == "geglu256")
    work, exe = gemm_sim
    trace = []
    if kind == "linear320":
        M, N, K = 300, 640, 640
        a, w, bias, res = _rnd((M, K), 70), _rnd((N, K), 71, 1 / math.sqrt(K)), _rnd((N,), 72), _rnd((M, N), 73)
        got = _run_gemm_job(work, exe, kind, dict(a=a, w=w, bias=bias, residual=res),
                            dict(_BASE, M=M, N=N, K=K, lda=K, ldc=N, ldr=N, c1=K, force=6), 1, trace=trace)
        ref = a.float() @ w.float().t() + bias.float() + res.float()
    elif kind == "conv320":
        n, h, w_, cin, cout = 2, 12, 12, 128, 320
        x, wt, bias = _rnd((n, cin, h, w_), 74), _rnd((cout, cin, 3, 3), 75, 1 / math.sqrt(9 * cin)), _rnd((cout,), 76)
        ref, ho, wo = _conv_ref(x, wt, bias)
        xl = x.permute(0, 2, 3, 1).reshape(n * h * w_, cin).contiguous()
        got = _run_gemm_job(work, exe, kind, dict(a=xl, w=_pack(wt), bias=bias),
                            dict(_BASE, M=n * ho * wo, N=cout, K=9 * cin, lda=cin, ldc=cout, c1=cin, mode=1, hin=h, win=w_, hout=ho, wout=wo,
                                 force=6), 1, trace=trace)
    else:
        M, C = 300, 64
        a, wf, bf = _rnd((M, C), 77), _rnd((8 * C, C), 78, 1 / math.sqrt(C)), _rnd((8 * C,), 79, 0.1)
        hfull = a.float() @ wf.float().t() + bf.float()
        ref = hfull[:, :4 * C] * F.gelu(hfull[:, 4 * C:])
        idx = torch.arange(4 * C).view(-1, 16)
        perm = torch.cat([idx, idx + 4 * C], dim=1).reshape(-1)
        got = _run_gemm_job(work, exe, kind, dict(a=a, w=wf[perm].contiguous(), bias=bf[perm].contiguous()),
                            dict(_BASE, M=M, N=8 * C, K=C, lda=C, ldc=4 * C, c1=C, geglu=1, force=7), 1, trace=trace)
    assert "block 512" in trace[0], trace[0]
    _close(got, ref, atol=6e-3)


# ---- the attention kernels (MFMA, transpose LDS reads, cross-lane softmax reductions) on the host simulator ---------------
@pytest.fixture(scope="module")
def attn_sim(tmp_path_factory):
    if not os.path.exists(CLANG):
        pytest.skip("ROCm host clang not available")
    work = tmp_path_factory.mktemp("attn_sim")
    import sim_lib
    text = sim_lib.transform(open(os.path.join(ROOT, "musev_amd", "csrc", "attention.hip")).read())
    (work / "attention_sim.inc").write_text(text)
    shutil.copy(os.path.join(SIM, "attention_main.cpp"), work / "attention_main.cpp")
    exe = work / "attention_sim"
    r = subprocess.run([CLANG, "-O1", "-std=c++17", "-pthread", "-w", "-I", SIM, "-I", str(work), "-o", str(exe), str(work / "attention_main.cpp")],
                       cwd=work, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    return work, exe


def _run_attn_job(work, exe, name, tensors: dict, kv: dict, timeout=900):
    d = work / name
    d.mkdir(exist_ok=True)
    for p in d.glob("*.bin"):
        p.unlink()
    for k, t in tensors.items():
        t.contiguous().numpy().tofile(d / f"{k}.bin")
    (d / "job.txt").write_text("".join(f"{k} {float(v)!r}\n" for k, v in kv.items()))
    r = subprocess.run([str(exe), str(d)], capture_output=True, text=True, timeout=timeout, env=dict(os.environ, SIM_TRACE="1"))
    assert r.returncode == 0, (r.returncode, r.stderr[-2000:])
    return torch.from_numpy(np.fromfile(d / "out.bin", dtype=np.float16).reshape(int(kv["rows"]), int(kv["ldo"]))).float()


def _attn_ref(q, ks, vs, heads, d, scale):
    nb, lq, c = q.shape
    qh = q.float().view(nb, lq, heads, d).transpose(1, 2)
    kh = ks.float().view(nb, -1, heads, d).transpose(1, 2)
    vh = vs.float().view(nb, -1, heads, d).transpose(1, 2)
    o = torch.softmax((qh @ kh.transpose(-1, -2)) * scale, dim=-1) @ vh
    return o.transpose(1, 2).reshape(nb * lq, c)


@pytest.mark.parametrize("d,variant", [(40, 3), (80, 3), (160, 3)])
def test_attention_self_plus_condition_frame_on_the_host(attn_sim, d, variant):
    """reference-only self-attention: two segments (own frame | vision-condition frame of the batch item), ragged lengths
    (lq = 70: a partial query tile; 70 keys per segment: a partial key tile), fused QKV storage (ld = 3C)"""
    work, exe = attn_sim
    heads, b, t, lq = 2, 1, 2, 70
    c, nb = heads * d, 1 * 2
    qkv = _rnd((nb * lq, 3 * c), 80 + d)
    q, k, v = qkv[:, :c], qkv[:, c:2 * c], qkv[:, 2 * c:]
    scale = d ** -0.5
    k3, v3 = k.reshape(nb, lq, c), v.reshape(nb, lq, c)
    cond = [(n // t) * t + 1 for n in range(nb)]
    ref = _attn_ref(q.reshape(nb, lq, c), torch.cat([k3, k3[cond]], 1), torch.cat([v3, v3[cond]], 1), heads, d, scale)
    # the kernels read q / k / v as column slices of ONE fused buffer: hand the simulator the same strided layout
    kvj = dict(rows=nb * lq, ldo=c, variant=variant, temporal=0, ldq=3 * c, nb=nb, lq=lq, heads=heads, d=d, scale=scale, nseg=2,
               accumulate=0, out_scale=1.0, ldk0=3 * c, ldv0=3 * c, len0=lq, div0=1, mul0=1, add0=0, ldk1=3 * c, ldv1=3 * c, len1=lq,
               div1=t, mul1=t, add1=1)
    flat = qkv.reshape(-1)
    tensors = dict(q=flat, k0=flat[c:], v0=flat[2 * c:], k1=flat[c:], v1=flat[2 * c:])
    got = _run_attn_job(work, exe, f"self_d{d}_v{variant}", tensors, kvj)
    _close(got, ref, atol=3e-3)


def test_attention_cross_accumulate_and_temporal_on_the_host(attn_sim):
    work, exe = attn_sim
    # text cross-attention (77 keys shared by the t frames of a batch item) + IP-Adapter second attention accumulated with a scale
    heads, d, nb, t, lq, lk = 2, 80, 4, 2, 40, 77
    c, b = heads * d, 2
    q = _rnd((nb * lq, c), 90)
    kvt = _rnd((b * lk, 2 * c), 91)
    bidx = [n // t for n in range(nb)]
    ref = _attn_ref(q.reshape(nb, lq, c), kvt[:, :c].reshape(b, lk, c)[bidx], kvt[:, c:].reshape(b, lk, c)[bidx], heads, d, d ** -0.5)
    base = dict(rows=nb * lq, ldo=c, variant=3, temporal=0, ldq=c, nb=nb, lq=lq, heads=heads, d=d, scale=d ** -0.5, nseg=1, accumulate=0,
                out_scale=1.0, ldk0=2 * c, ldv0=2 * c, len0=lk, div0=t, mul0=1, add0=0)
    flat = kvt.reshape(-1)
    got = _run_attn_job(work, exe, "cross", dict(q=q, k0=flat, v0=flat[c:]), base)
    _close(got, ref, atol=3e-3)
    kvi = _rnd((b * 4, 2 * c), 92)
    ref_ip = _attn_ref(q.reshape(nb, lq, c), kvi[:, :c].reshape(b, 4, c)[bidx], kvi[:, c:].reshape(b, 4, c)[bidx], heads, d, d ** -0.5)
    fl2 = kvi.reshape(-1)
    got2 = _run_attn_job(work, exe, "cross_ip", dict(q=q, k0=fl2, v0=fl2[c:], out0=got.half()),
                         dict(base, len0=4, accumulate=1, out_scale=0.7))
    _close(got2, got.half().float() + 0.7 * ref_ip, atol=3e-3)
    # temporal attention over t = 13 frames per pixel (rows stay in (b, t, p) order), both kernels
    bb, tt, hw, heads, d = 1, 13, 9, 2, 40
    c = heads * d
    qkv = _rnd((bb * tt * hw, 3 * c), 93)

    def seq(x):
        return x.reshape(bb, tt, hw, c).permute(0, 2, 1, 3).reshape(bb * hw, tt, c)
    ref = _attn_ref(seq(qkv[:, :c]), seq(qkv[:, c:2 * c]), seq(qkv[:, 2 * c:]), heads, d, d ** -0.5)
    ref = ref.reshape(bb, hw, tt, c).permute(0, 2, 1, 3).reshape(bb * tt * hw, c)
    flat = qkv.reshape(-1)
    for variant in (3,):
        got = _run_attn_job(work, exe, f"temporal{variant}", dict(q=flat, k0=flat[c:], v0=flat[2 * c:]),
                            dict(rows=bb * tt * hw, ldo=c, variant=variant, temporal=1, ldq=3 * c, ldk0=3 * c, ldv0=3 * c, b=bb, t=tt, hw=hw,
                                 heads=heads, d=d, scale=d ** -0.5))
        _close(got, ref, atol=3e-3)


# ---- the tile-configuration catalogue (mv_set_gemm_force): what the per-shape tuner may pick must be right everywhere ------
def _catalogue():
    import ctypes as C
    from musev_amd import _lib
    lib = _lib.load()
    out = []
    for i in range(lib.mv_gemm_num_configs()):
        d = (C.c_int32 * 5)()
        assert lib.mv_gemm_config_desc(i, d) == 0
        out.append(tuple(d))
    return out


def test_gemm_catalogue_is_consistent():
    cat = _catalogue()
    assert len(cat) == 19 and len(set(cat)) == 19, "configurations must be distinct"  # (round 3 removed the ping-pong ids 19-22)
    for rows, cols, waves, bk, stages in cat:
        assert rows in (32, 64, 128, 256) and cols in (80, 128, 160, 256, 320) and waves in (2, 4, 8) and (bk, stages) in ((64, 2), (64, 3))
        assert stages * (rows + cols) * bk * 2 <= 160 * 1024, "operand stages must fit the 160 KB LDS"


@pytest.mark.parametrize("cfg", list(range(19)))
def test_gemm_every_configuration_on_the_host(gemm_sim, cfg):
    """linear GEMM with the full epilogue, forced onto each catalogue entry: ragged M (300) and N = 320 (ragged for the 128- and
    256-wide tiles), K = 192 = three 64-deep or six 32-deep K steps (every ring wraps), LATEST legal LDS-DMA landing; the GEGLU
    epilogue on the even-TN configurations"""
    if not _FULL and cfg not in (0, 6, 12, 13, 15, 17, 18):
        pytest.skip("covered by MUSEV_SIM_FULL=1 (every configuration was run when it was added)")
    work, exe = gemm_sim
    rows, cols, waves, bk, stages = _catalogue()[cfg]
    M, N, K = 300, 320, 192
    a, w, bias, res = _rnd((M, K), 100), _rnd((N, K), 101, 1 / math.sqrt(K)), _rnd((N,), 102), _rnd((M, N), 103)
    rowbias = _rnd((2, N), 104)
    trace = []
    got = _run_gemm_job(work, exe, f"cfg{cfg}", dict(a=a, w=w, bias=bias, rowbias=rowbias, residual=res),
                        dict(_BASE, M=M, N=N, K=K, lda=K, ldc=N, ldr=N, ldrb=N, c1=K, rows_per_group=150, force=cfg), 1, trace=trace)
    assert f"block {64 * waves} " in trace[0] and f"grid {((M + rows - 1) // rows) * ((N + cols - 1) // cols)} " in trace[0], trace[0]
    _close(got, a.float() @ w.float().t() + bias.float() + rowbias.float()[torch.arange(M) // 150] + res.float())
    if cols in (128, 256):  # even TN: the GEGLU epilogue applies
        Mg, C = 140, 32
        ag, wf, bf = _rnd((Mg, 2 * C), 105), _rnd((16 * C, 2 * C), 106, 1 / math.sqrt(2 * C)), _rnd((16 * C,), 107, 0.1)
        hfull = ag.float() @ wf.float().t() + bf.float()
        idx = torch.arange(8 * C).view(-1, 16)
        perm = torch.cat([idx, idx + 8 * C], dim=1).reshape(-1)
        gotg = _run_gemm_job(work, exe, f"cfg{cfg}g", dict(a=ag, w=wf[perm].contiguous(), bias=bf[perm].contiguous()),
                             dict(_BASE, M=Mg, N=16 * C, K=2 * C, lda=2 * C, ldc=8 * C, c1=2 * C, geglu=1, force=cfg), 1)
        _close(gotg, hfull[:, :8 * C] * F.gelu(hfull[:, 8 * C:]))


@pytest.mark.parametrize("cfg", [10, 11, 12, 13, 14, 15, 16, 17, 18])
def test_gemm_new_configurations_conv_on_the_host(gemm_sim, cfg):
    """the configurations added for the tuner, on the two-source 3x3 convolution with stride 2 (halo + tap walk + concat)"""
    _subset(cfg in (12, 15, 18))
    work, exe = gemm_sim
    n, h, w, c1, c2, cout = 2, 9, 12, 64, 64, 320
    cin = c1 + c2
    x, wt, bias = _rnd((n, cin, h, w), 110), _rnd((cout, cin, 3, 3), 111, 1 / math.sqrt(9 * cin)), _rnd((cout,), 112)
    ref, ho, wo = _conv_ref(x, wt, bias, stride=2)
    xl = x.permute(0, 2, 3, 1).reshape(n * h * w, cin).contiguous()
    got = _run_gemm_job(work, exe, f"cfgconv{cfg}", dict(a=xl[:, :c1].contiguous(), a2=xl[:, c1:].contiguous(), w=_pack(wt), bias=bias),
                        dict(_BASE, M=n * ho * wo, N=cout, K=9 * cin, lda=c1, lda2=c2, ldc=cout, c1=c1, c2=c2, mode=1, stride=2, hin=h, win=w,
                             hout=ho, wout=wo, force=cfg), 1)
    _close(got, ref, atol=6e-3)


def test_gemm_tuned_table_lookup_on_the_host(tmp_path_factory):
    """the measured table of gemm_tuned.h (written by tools/gpu_gemm_tune.py; {mode, M, N, K, geglu, ln, cfg, nsplit}, looked up by
    exact (mode, N, K, geglu) and the nearest M within a factor of 3, then the keyed table {mode, geglu, ln, K bucket, fill bucket,
    cfg}): a build with a two-entry table and one keyed entry must send the matching problems -- and the same layer at a nearby M in
    the same key bucket -- to the listed configurations, a problem more than 3x away in M to the rules, a problem of ANOTHER layer
    that falls on the keyed entry to that entry's tile, with unchanged results; cfg = -2 (rules only) and a forced id through the
    descriptor"""
    if not os.path.exists(CLANG):
        pytest.skip("ROCm host clang not available")
    import sim_lib
    work = tmp_path_factory.mktemp("gemm_sim_tuned")
    table = work / "gemm_tuned_test.h"
    table.write_text("static const GemmTuned kGemmTuned[] = {\n    {0, 300, 320, 192, 0, 0, 6, 1},\n    {0, 140, 512, 64, 1, 0, 7, 1},\n"
                     "    {-1, 0, 0, 0, 0, 0, -1, 0},\n};\nstatic const int kNumGemmTuned = 2;\n"
                     # K = 512 -> 8 K tiles = bucket 1; 90 x 320 under 256 CUs = fill bucket 0 -> catalogue id 13 (64 x 80 tiles, 2 waves)
                     "static const GemmKeyed kGemmKeyed[] = {\n    {0, 0, 0, 1, 0, 13},\n    {-1, 0, 0, 0, 0, -1},\n};\nstatic const int kNumGemmKeyed = 1;\n")
    src = open(os.path.join(ROOT, "musev_amd", "csrc", "gemm.hip")).read().replace('#include "gemm_tuned.h"', f'#include "{table}"')
    (work / "gemm_sim.inc").write_text(sim_lib.transform(src))
    shutil.copy(os.path.join(SIM, "gemm_main.cpp"), work / "gemm_main.cpp")
    exe = work / "gemm_sim"
    r = subprocess.run([CLANG, "-O1", "-std=c++17", "-pthread", "-w", "-I", SIM, "-I", str(work), "-o", str(exe), str(work / "gemm_main.cpp")],
                       cwd=work, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    M, N, K = 300, 320, 192
    a, w = _rnd((M, K), 120), _rnd((N, K), 121, 1 / math.sqrt(K))
    ref = a.float() @ w.float().t()
    base = dict(_BASE, M=M, N=N, K=K, lda=K, ldc=N, c1=K)
    for extra, block in ((dict(), 512), (dict(force=-2), 256), (dict(force=0), 256)):   # table -> cfg 6 (8 waves); rules; forced cfg 0
        trace = []
        got = _run_gemm_job(work, exe, "t" + str(block) + str(len(extra)), dict(a=a, w=w), dict(base, **extra), 1, trace=trace)
        assert f"block {block} " in trace[0], (extra, trace[0])
        _close(got, ref)
    trace = []   # the same layer at a nearby M inherits the entry ...
    got = _run_gemm_job(work, exe, "tnear", dict(a=a[:200], w=w), dict(base, M=200), 1, trace=trace)
    assert "block 512 " in trace[0], trace[0]
    _close(got, ref[:200])
    trace = []   # ... more than 3x away it follows the rules
    got = _run_gemm_job(work, exe, "tmiss", dict(a=a[:90], w=w), dict(base, M=90), 1, trace=trace)
    assert "block 256 " in trace[0], trace[0]
    _close(got, ref[:90])
    # a layer the table does not hold at all (K = 512), in the bucket of the keyed entry: that entry's tile (2 waves)
    a2, w2 = _rnd((90, 512), 122), _rnd((N, 512), 123, 1 / math.sqrt(512))
    trace = []
    got = _run_gemm_job(work, exe, "tkeyed", dict(a=a2, w=w2), dict(base, M=90, K=512, lda=512, c1=512), 1, trace=trace)
    assert "block 128 " in trace[0], trace[0]
    _close(got, a2.float() @ w2.float().t())
