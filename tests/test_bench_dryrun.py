"""CPU dry run of bench.py's main() (the script the driver times at round end): small `musev` architecture on the emulated
kernels, CUDA entry points replaced by stand-ins -- checks that the whole flow (timed loop with the warm-up callback,
recorded-launch roofline pass, JSON assembly) runs and that the line carries every field of the contract.  Numbers are
meaningless here; on the GPU the same code runs on libmusev_hip.so."""
import json
import os
import sys

import pytest
import torch

import emu_ops


class _FakeEvent:
    _clock = [0.0]

    def __init__(self, enable_timing=True):
        self.t = 0.0

    def record(self):
        _FakeEvent._clock[0] += 0.01
        self.t = _FakeEvent._clock[0]

    def elapsed_time(self, other):
        return other.t - self.t


def test_bench_main_dry_run(monkeypatch, capsys):
    import bench
    from musev_amd import ops
    from oracle import unet3d
    from musev_amd.models.unet_loader import load_unet_by_name
    from musev_amd.pipelines.parallel_denoise import ParallelDenoiser
    emu_ops.install(monkeypatch)
    arch = dict(block_out_channels=(320, 640), layers_per_block=1, down_block_types=("CrossAttnDownBlock3D", "DownBlock3D"),
                up_block_types=("UpBlock3D", "CrossAttnUpBlock3D"))
    cfg = unet3d.flavour_config("musev", **arch)
    sd = unet3d.init_state_dict(cfg, 3)

    def build_unet(flavour, dev):
        m = load_unet_by_name(flavour, sd_unet_model=sd, dtype=torch.float16, **arch)
        m._device_check = False
        return m

    # every mv_gemm_f16 launch goes through ops._launch_gemm on the GPU; the emulation bypasses it, so feed the launch recorder
    # here (problem geometry only) and time the "replays" with a constant per launch
    from types import SimpleNamespace
    for name, mode in (("gemm", 0), ("conv3x3", 1), ("tconv3", 2)):
        inner = getattr(ops, name)

        def f(*a, _inner=inner, _mode=mode, **k):
            out = _inner(*a, **k)
            if ops.GEMM_RECORD is not None:
                w = a[1]
                ops.GEMM_RECORD.append((SimpleNamespace(mode=_mode, M=out.shape[0], N=w.shape[0], K=w.shape[1], geglu=int(bool(k.get("geglu")))), (), 1000))
            return out
        monkeypatch.setattr(ops, name, f)
    monkeypatch.setattr(ops, "replay_gemms", lambda rec, reps=1: 0.01 * len(rec) * reps)
    monkeypatch.setattr(ops, "replay_gemms_two_streams", lambda a, b, reps=1: 0.008 * (len(a) + len(b)) * reps)

    monkeypatch.setattr(bench, "build_unet", build_unet)
    monkeypatch.setattr(ParallelDenoiser, "_device_check", False)
    monkeypatch.setattr(torch.cuda, "set_device", lambda d: None)
    monkeypatch.setattr(torch.cuda, "synchronize", lambda *a: None)
    real_device = torch.device
    monkeypatch.setattr(bench.torch, "device", lambda *a, **k: real_device("cpu"))
    monkeypatch.setenv("WORLD_SIZE", "1")
    monkeypatch.setattr(sys, "argv", ["bench.py", "--size", "64", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"])
    bench.main()
    line = [ln for ln in capsys.readouterr().out.splitlines() if ln.startswith("{")][-1]
    rec = json.loads(line)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
                "data", "config", "roofline", "cpu_baseline"):
        assert key in rec, key
    assert rec["n_gpus"] == 1 and rec["steps"] == 2 and rec["warmup"] == 1 and rec["higher_is_better"] is True
    assert rec["unit"] == "frames/s" and rec["value"] > 0 and rec["ms_per_step"] > 0 and rec["dtype"] == "f16" and rec["vs_baseline"] is None
    assert rec["config"]["workload"].startswith("config2") and rec["config"]["frames"] == 12 and rec["config"]["output_finite"] is True
    rf = rec["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in rf, key
    assert rf["bound"] == "mfma" and rf["peak"] == 2500.0 and rf["launches_per_step"] > 50 and set(rf["by_mode"]) == {"linear", "conv3x3", "tconv3"}
    assert abs(rf["avg_launch_ms"] - 0.01) < 1e-9 and rec["config"]["graphs"] is False   # (no device: eager emulation)
    assert rec["cpu_baseline"] is None   # --no-cpu-baseline


class _MP:
    """monkeypatch stand-in for spawned workers (their patches die with the process)"""
    def setattr(self, o, n, v):
        setattr(o, n, v)

    def setenv(self, k, v):
        os.environ[k] = v


def _bench_rank(rank, world, port, ret):
    import io
    import contextlib
    from types import SimpleNamespace
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(1)
    import bench
    from musev_amd import ops
    from oracle import unet3d
    from musev_amd.models.unet_loader import load_unet_by_name
    from musev_amd.pipelines.parallel_denoise import ParallelDenoiser
    mp_ = _MP()
    emu_ops.install(mp_)
    arch = dict(block_out_channels=(320, 640), layers_per_block=1, down_block_types=("CrossAttnDownBlock3D", "DownBlock3D"),
                up_block_types=("UpBlock3D", "CrossAttnUpBlock3D"))
    sd = unet3d.init_state_dict(unet3d.flavour_config("musev", **arch), 3)

    def build_unet(flavour, dev):
        m = load_unet_by_name(flavour, sd_unet_model=sd, dtype=torch.float16, **arch)
        m._device_check = False
        return m

    bench.build_unet = build_unet
    ParallelDenoiser._device_check = False
    torch.cuda.set_device = lambda d: None
    torch.cuda.synchronize = lambda *a: None
    torch.cuda.get_device_properties = lambda d: SimpleNamespace(uuid=None)
    torch.cuda.current_device = lambda: 0
    real_device = torch.device
    bench.torch.device = lambda *a, **k: real_device("cpu")
    sys.argv = ["bench.py", "--gpus", str(world), "--rehearse-shared-gpu", "--workload", "weak", "--size", "64", "--steps", "1", "--warmup", "1", "--no-cpu-baseline",
                "--no-roofline"]
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        bench.main()
    ret[rank] = buf.getvalue()
    import torch.distributed as dist
    if dist.is_initialized():
        dist.destroy_process_group()


def test_bench_two_ranks_over_gloo_emits_the_multi_gpu_block():
    """`bench.py --gpus 2` as the driver's SCALE run launches it (one process per rank, RANK / WORLD_SIZE / MASTER_* from the
    environment), on the CPU over gloo with the kernels emulated: rank 0 alone prints the line, and the line carries the `multi_gpu`
    diagnostics -- backend, rank count, distinct devices, every rank's own ms_per_step, exposed exchange time and unit count (VERDICT
    r4 item 8).  The `weak` workload keeps the CPU run short (16 frames -> 2 windows x 2 CFG halves = 4 units, 2 per rank); the
    driver's default for N > 1 is config 4, whose unit list is covered by tests/test_parallel_sharding.py (world 8)."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_bench_rank, args=(2, port, ret), nprocs=2, join=True)
    out = dict(ret)
    assert not [ln for ln in out[1].splitlines() if ln.startswith("{")], "only rank 0 prints the line"
    rec = json.loads([ln for ln in out[0].splitlines() if ln.startswith("{")][-1])
    assert rec["n_gpus"] == 2 and rec["scaling"] == "weak" and rec["config"]["workload"].startswith("weak") and rec["rehearsal"]
    m = rec["multi_gpu"]
    assert m["backend"] == "gloo" and m["ranks"] == 2 and m["distinct_devices"] == 1
    assert len(m["per_rank_ms_per_step"]) == 2 and all(v > 0 for v in m["per_rank_ms_per_step"])
    assert len(m["per_rank_exposed_exchange_ms_per_step"]) == 2 and m["per_rank_units"] == [2, 2]
    assert rec["config"]["windows"] == 2 and rec["config"]["units_per_gpu_max"] == 2 and rec["config"]["ideal_speedup_vs_1gpu_same_workload"] == 2.0
    assert rec["value"] > 0 and rec["config"]["output_finite"] is True


def test_committed_pmc_traffic_is_a_measurement_of_the_shipping_kernel_sources():
    """roofline.traffic comes from profiles/hbm_traffic.json, which is bound to the hash of the kernel sources it was measured on
    (bench.measured_traffic): a tree whose kernels changed after the last PMC pass -- or a file bench.py cannot read -- would report
    `traffic: null` on the driver's box.  Re-run tools/gpu_final_profile.sh and commit the summary when this fails."""
    import bench
    per_step, source = bench.measured_traffic("config2")
    assert per_step is not None, source
    assert 40e9 < per_step < 150e9 and "pmc_summary.json" in source
