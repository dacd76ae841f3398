"""CPU dry run of tools/gpu_gemm_tune.py (the first GPU call of the next round depends on it): the model is the small `musev`
architecture on the emulated kernels, the CUDA bits are stand-ins, and every GEMM-family launch reports a synthetic duration
that depends on the configuration in the (re-issued) descriptor -- so the aggregation, the choice rule and the generated gemm_tuned.h can be checked
without a GPU.  The header it writes is then compiled into the host-simulator build of gemm.hip."""
import importlib.util
import os
import re
import sys
import types

import pytest
import torch

import emu_ops

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def fake_desc(mode, a, w, out, geglu=False, c2=0, conv=None, tconv=None):
    """a GemmDesc with the real problem geometry and made-up (non-null, 16-byte aligned) addresses: enough for the library's
    host-side entry points (mv_gemm_workspace_bytes / mv_gemm_choice validate and choose, they launch nothing)"""
    from musev_amd import _lib
    d = _lib.GemmDesc()
    d.a, d.w, d.c = 0x10000, 0x20000, 0x30000
    d.M, d.N, d.K = out.shape[0], w.shape[0], w.shape[1]
    d.lda, d.ldc, d.c1 = a.shape[1], out.shape[1], a.shape[1]
    if c2:
        d.a2, d.lda2, d.c2 = 0x40000, c2, c2
    d.mode, d.geglu, d.cfg = mode, int(geglu), -1
    if conv is not None:
        d.hin, d.win, d.hout, d.wout, d.stride, d.upsample = conv
    if tconv is not None:
        d.t, d.hw = tconv
    return d


def test_gemm_tuner_dry_run(monkeypatch, tmp_path):
    from musev_amd import _lib, ops
    from oracle import unet3d
    from musev_amd.models.unet_loader import load_unet_by_name
    emu_ops.install(monkeypatch)
    monkeypatch.setattr(emu_ops, "STRICT_WIDTHS", True)
    lib = _lib.load()

    # synthetic timing model: rules = 1.0 ms; configuration 6 is 20 % faster on conv3x3 problems (30 % with 4 K slices on
    # configuration 1), configuration 13 is 10 % faster on linear problems with K <= 320, configuration 3 is 1 % faster everywhere
    # (below the 3 % threshold: must NOT be picked)
    def fake_ms(mode, K, cfg, split):
        if cfg == 1 and split == 4 and mode == 1:
            return 0.7
        if cfg == 6 and mode == 1:
            return 0.8
        if cfg == 13 and mode == 0 and K <= 320:
            return 0.9
        if cfg == 3 and split <= 1:
            return 0.99
        return 1.0 if cfg == -2 else 1.05

    # the emulated kernels bypass ops._launch_gemm: feed the launch recorder here, and time "replays" with the synthetic model
    def wrap(name, mode_of):
        inner = getattr(emu_ops, name)

        def f(*a, **k):
            out = inner(*a, **k)
            if ops.GEMM_RECORD is not None:
                x2 = k.get("a2") if name == "gemm" else k.get("x2")
                conv = tconv = None
                if name == "conv3x3":
                    h, w_ = a[3], a[4]
                    st, up = k.get("stride", 1), int(bool(k.get("upsample")))
                    ho, wo = (2 * h, 2 * w_) if up else ((h - 1) // st + 1, (w_ - 1) // st + 1)
                    conv = (h, w_, ho, wo, st, up)
                if name == "tconv3":
                    tconv = (a[3], a[4])
                d = fake_desc(mode_of, a[0], a[1], out if not k.get("geglu") else out.new_empty(out.shape[0], 2 * out.shape[1]),
                              geglu=bool(k.get("geglu")), c2=0 if x2 is None else x2.shape[1], conv=conv, tconv=tconv)
                if k.get("geglu"):
                    d.ldc = out.shape[1]
                ops.GEMM_RECORD.append((d, (), 0))
            return out
        monkeypatch.setattr(ops, name, f)

    wrap("gemm", 0)
    wrap("conv3x3", 1)
    wrap("tconv3", 2)
    monkeypatch.setattr(ops, "replay_gemms", lambda rec, reps=1: sum(fake_ms(int(d.mode), int(d.K), int(d.cfg), int(d.splitk)) for d, _k, _b in rec) * reps)

    arch = dict(block_out_channels=(320, 640), layers_per_block=1, down_block_types=("CrossAttnDownBlock3D", "DownBlock3D"),
                up_block_types=("UpBlock3D", "CrossAttnUpBlock3D"))
    cfg = unet3d.flavour_config("musev", **arch)
    sd = unet3d.init_state_dict(cfg, 3)

    def build_unet(flavour, dev):
        m = load_unet_by_name(flavour, sd_unet_model=sd, dtype=torch.float16, **arch)
        m._device_check = False
        return m

    spec = importlib.util.spec_from_file_location("gpu_gemm_tune", os.path.join(ROOT, "tools", "gpu_gemm_tune.py"))
    tune = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(tune)
    import bench
    monkeypatch.setattr(bench, "build_unet", build_unet)
    monkeypatch.setattr(torch.cuda, "set_device", lambda d: None)
    monkeypatch.setattr(torch.cuda, "synchronize", lambda *a: None)
    monkeypatch.setattr(torch.cuda, "get_device_name", lambda *a: "dry run")
    real_device = torch.device
    monkeypatch.setattr(tune.torch, "device", lambda *a, **k: real_device("cpu"))
    monkeypatch.setattr(tune, "ROOT", str(tmp_path))
    monkeypatch.setattr(sys, "argv", ["gpu_gemm_tune.py", "dry", "--size", "64", "--reps", "1"])
    tune.main()
    assert (ops.GEMM_CFG, ops.GEMM_SPLITK) == (-1, 0) and ops.GEMM_RECORD is None, "the tuner must leave the process on `table + rules`"

    hdr = open(tmp_path / "gpurun_out" / "dry_gemm_tuned.h").read()
    # {mode, M, N, K, geglu, ln, cfg, nsplit}; every problem has an entry, cfg -2 = "the rules measured best"
    entries = [tuple(int(v) for v in m.groups()) for m in re.finditer(r"\{(\d+), (\d+), (\d+), (\d+), (\d+), (\d+), (-?\d+), (\d+)\},", hdr)]
    entries = [e for e in entries if e[0] >= 0]
    n_decl = int(re.search(r"kNumGemmTuned = (\d+);", hdr).group(1))
    assert n_decl == len(entries) > 0
    assert all((c, sp) == (1, 4) for mode, M, N, K, g, ln, c, sp in entries if mode == 1) and any(mode == 1 for mode, *_ in entries)
    lin = [(K, c, sp) for mode, M, N, K, g, ln, c, sp in entries if mode == 0]
    assert all((c, sp) == ((13, 1) if K <= 320 else (-2, 0)) for K, c, sp in lin) and any(K <= 320 for K, _, _ in lin) and any(K > 320 for K, _, _ in lin)
    assert all(c == -2 for mode, M, N, K, g, ln, c, sp in entries if mode == 2) and any(mode == 2 for mode, *_ in entries), "a 1 % gain is below the threshold"
    import json
    rep = json.load(open(tmp_path / "gpurun_out" / "dry_gemm_tune.json"))
    assert rep["tuned_ms_per_forward"] < rep["rules_ms_per_forward"] and len(rep["problems"]) > 10
    # the generated header must compile into gemm.hip (host-simulator build)
    import shutil
    import subprocess
    import sim_lib
    if not os.path.exists(sim_lib.CLANG):
        return
    src = open(os.path.join(ROOT, "musev_amd", "csrc", "gemm.hip")).read().replace('#include "gemm_tuned.h"', '#include "%s"' % (tmp_path / "gpurun_out" / "dry_gemm_tuned.h"))
    (tmp_path / "gemm_sim.inc").write_text(sim_lib.transform(src))
    shutil.copy(os.path.join(sim_lib.SIM, "gemm_main.cpp"), tmp_path / "gemm_main.cpp")
    r = subprocess.run([sim_lib.CLANG, "-O0", "-std=c++17", "-pthread", "-w", "-fsyntax-only", "-I", sim_lib.SIM, "-I", str(tmp_path), str(tmp_path / "gemm_main.cpp")],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]


def test_trace_overlap_on_a_synthetic_two_stream_trace(tmp_path):
    """tools/trace_overlap.py (round 6: the in-step roofline from a rocprofv3 kernel trace with timestamps): two streams, known
    overlaps -- the families' attributed times add up to the busy time, a family's time never exceeds the step, the overlap share and
    the per-step cut at cfg_ddim_step are exact."""
    import json
    import subprocess
    rows = ["\"Kind\",\"Agent_Id\",\"Queue_Id\",\"Stream_Id\",\"Kernel_Name\",\"Start_Timestamp\",\"End_Timestamp\""]
    t = 1000

    def k(q, name, s, e):
        rows.append(f"\"KERNEL_DISPATCH\",1,{q},{q},\"{name}\",{s},{e}")
    for step in range(3):   # 3 steps of 10 000 ns: the first is warm-up
        b = t + step * 10000
        k(1, "void gemm2_kernel<0, 2, 5>(GemmArgs2)", b + 0, b + 4000)        # alone 0-2000, beside the other stream's gemm 2000-4000
        k(2, "void gemm2_kernel<1, 8, 5>(GemmArgs2)", b + 2000, b + 6000)     # beside attention 4000-6000
        k(1, "void attn3_kernel<40, 47>(AttnArgs)", b + 4000, b + 7000)       # alone 6000-7000
        k(2, "void gn_apply_kernel<false>(GnArgs)", b + 7000, b + 8000)       # alone
        k(1, "void cfg_ddim_step_kernel(float*)", b + 9000, b + 10000)        # closes the step (idle 8000-9000)
    p = tmp_path / "x_kernel_trace.csv"
    p.write_text("\n".join(rows) + "\n")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "trace_overlap.py"), str(p), "--steps", "2", "--family-tflop", "0.01"],
                         check=True, capture_output=True, text=True).stdout
    d = json.loads(out)
    assert d["steps"] == 2 and abs(d["wall_ms_per_step"] - 0.010) < 1e-9
    assert abs(d["busy_ms_per_step"] - 0.009) < 1e-9                   # 0-8000 + 9000-10000
    assert abs(d["overlap_frac_of_busy"] - 4000 / 9000) < 1e-9         # 2000-6000
    fam = d["families"]
    assert abs(fam["matrix"]["serial_ms"] - 0.008) < 1e-9 and abs(fam["matrix"]["union_ms"] - 0.006) < 1e-9
    assert abs(fam["matrix"]["attributed_ms"] - (2000 + 2000 + 1000) / 1e6) < 1e-9   # alone, two gemms, half of gemm + attention
    assert abs(fam["attention"]["attributed_ms"] - (1000 + 1000) / 1e6) < 1e-9
    assert abs(sum(f["attributed_ms"] for f in fam.values()) - d["busy_ms_per_step"]) < 1e-9
    assert all(f["attributed_ms"] <= d["wall_ms_per_step"] for f in fam.values())
    assert abs(d["matrix_roofline"]["in_step"]["tflops"] - 0.01 / 5e-6) < 1e-3
