#!/usr/bin/env python3
"""bench.py -- MuseV parallel-denoise hot path on MI355X: denoised frames/sec @512x512, 12-frame window, 20 DDIM steps.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one denoise step of the hot path: for every window owned by this rank, window gather -> UNet3D forward
(HIP kernels) -> prediction exchange (RCCL all-gather, N > 1 only) -> scatter-add / average / CFG / DDIM update.
The metric is quoted for a 20-step denoise, so  value = generated_frames / (20 * seconds_per_step), aggregated over
all ranks (the time is the MAX over ranks of K timed steps bracketed by barrier + device synchronize).

Workloads (BASELINE.json configs; synthetic N(0,1) latents / prompt embeddings, seeded random fp16 weights of the
real SD-1.5 MuseV architecture, 1.42 B parameters -- no checkpoints or datasets exist in this environment):
  N = 1  : config 2 -- text2video `musev`, 512x512 (latent 64x64), 12 frames + 1 vision-condition frame, CFG batch 2 (the
           configuration the metric is quoted on).  The same line carries `config4_n1`: a short 1-GPU run of config 4, the
           denominator of the strong-scaling curve the N > 1 runs belong to.
  N > 1  : config 4 -- 96 frames = 12 windows (window 12, overlap 4, `uniform`, incl. the wrap-around window) x 2 CFG halves =
           24 units sharded over the ranks ("scaling": "strong"; 3 units per rank at N = 8 -> ideal 8x the 1-GPU config-4 rate).
  --workload weak    : one window x 2 halves per rank (8*N frames): fixed per-GPU work.
  --workload config3 : `musev_referencenet` + IP-Adapter (+13 ReferEmbFuse attentions), single window.
  --workload config5 : `musev_referencenet_pose` (== `musev_referencenet` architecture, unet_loader.py:243-268) at 768x768, 48
                       frames = 6 windows, synthetic ControlNet residuals on the 12 skips + mid block.

Extra JSON objects (tier contract):
  roofline      the dominant kernel (the implicit-GEMM family, MFMA-bound): algorithmic FLOPs per launch / launch
                duration, measured live on the device: one step's launches recorded, then re-issued back to back on the
                launch stream between one HIP event pair; plus the whole-step figure (SURVEY.md 8d analytic FLOPs / step time).
  cpu_baseline  the oracle (plain torch fp32) timed on this box's host cores on a bounded sample (rank 0, N = 1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")  # before the first HIP call: see musev_amd/__init__.py (streams sharing a hardware queue serialise)

import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_MFMA_TFLOPS = 2500.0  # dense fp16/bf16 MFMA, MI355X (MI355X_MICROARCH.md)
DENOISE_STEPS = 20


def unet_flops(H, W, T, B, model="musev", n_vis=1, n_ip_tokens=4, text_tokens=77, xdim=768):
    """Algorithmic FLOPs (2 x MACs of GEMM / conv / attention-matmul terms) of one UNet3D forward -- the analytic counter
    of SURVEY.md Appendix C, kept identical so that roofline numerators agree with BASELINE.md."""
    F_ = B * T
    tot = 0

    def res(cin, cout, hw):
        return F_ * (hw * (9 * cin * cout + 9 * cout * cout + (cin * cout if cin != cout else 0)) + 1280 * cout)

    def tconv(c, hw):
        return F_ * hw * 12 * c * c

    def t2d(c, hw):
        kv = hw * (1 + n_vis)
        lin = hw * c * c * 2
        a1_lin = hw * c * c * 2 + kv * c * c * 2
        a1_att = 2 * hw * kv * c
        a2_lin = hw * c * c * 2 + text_tokens * xdim * c * 2
        a2_att = 2 * hw * text_tokens * c
        if model == "musev_referencenet":
            a2_lin += n_ip_tokens * xdim * c * 2
            a2_att += 2 * hw * n_ip_tokens * c
        return F_ * (lin + a1_lin + a2_lin + hw * 12 * c * c + a1_att + a2_att)

    def tt(c, hw):
        tok = B * hw * T
        return tok * c * c * 22 + tok * (2 * T * c) * 2

    def refer(c, hw, ref_hw):
        kv = ref_hw + hw
        return F_ * (hw * c * c * 2 + kv * c * c * 2 + 2 * hw * kv * c)

    hw0 = (H // 8) * (W // 8)
    hws = [hw0, hw0 // 4, hw0 // 16, hw0 // 64]
    ch = [320, 640, 1280, 1280]
    rn = model == "musev_referencenet"
    tot += F_ * hw0 * 9 * 4 * 320
    if model == "musev":
        tot += tt(320, hw0)
    if rn:
        tot += refer(320, hw0, hw0)
    cin = 320
    for i in range(4):
        cout, hw = ch[i], hws[i]
        for l in range(2):
            tot += res(cin if l == 0 else cout, cout, hw) + tconv(cout, hw)
            if i < 3:
                tot += t2d(cout, hw) + tt(cout, hw)
            if rn:
                tot += refer(cout, hw, hw)
        if i < 3:
            tot += F_ * hws[i + 1] * 9 * cout * cout
            if rn:
                tot += refer(cout, hws[i + 1], hws[i + 1])
        cin = cout
    hw = hws[3]
    tot += res(1280, 1280, hw) + tconv(1280, hw) + t2d(1280, hw) + tt(1280, hw) + res(1280, 1280, hw) + tconv(1280, hw)
    if rn:
        tot += refer(1280, hw, hw)
    rev = [1280, 1280, 640, 320]
    skips = [[1280, 1280, 1280], [1280, 1280, 640], [640, 640, 320], [320, 320, 320]]
    prev = 1280
    for i in range(4):
        cout, hw = rev[i], hws[3 - i]
        for l in range(3):
            rin = (prev if l == 0 else cout) + skips[i][l]
            tot += res(rin, cout, hw) + tconv(cout, hw)
            if i > 0:
                tot += t2d(cout, hw) + tt(cout, hw)
        if i < 3:
            tot += F_ * hws[3 - i - 1] * 9 * cout * cout
        prev = cout
    tot += F_ * hw0 * 9 * 320 * 4
    return 2 * tot


def kernel_source_hash() -> str:
    """sha256 over the kernel sources the library is built from (musev_amd/csrc/*.hip, *.h, include/musev_hip.h): ties a PMC
    measurement to the code it was taken on"""
    import glob
    import hashlib
    hsh = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(ROOT, "musev_amd", "csrc", "*.hip")) + glob.glob(os.path.join(ROOT, "musev_amd", "csrc", "*.h")) +
                   [os.path.join(ROOT, "include", "musev_hip.h")])
    for f in files:
        hsh.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            hsh.update(fh.read())
    return hsh.hexdigest()[:16]


def measured_traffic(workload: str):
    """HBM bytes of the GEMM family (gemm2_kernel + split-K reduce) PER DENOISE STEP from the PMC counters: collected offline by tools/gpu_profile.sh (rocprofv3 --pmc
    FETCH_SIZE / WRITE_SIZE in separate passes over this same bench command, gfx950 read correction x2 applied by
    tools/pmc_summary.py) and committed as profiles/hbm_traffic.json -- a counter pass cannot run inside the timed
    process.  The file records the hash of the kernel sources it was measured on: a measurement of OTHER code is not
    reported (None), so `traffic` is reproducible from profiles/ or absent."""
    path = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    try:
        with open(path) as f:
            rec = json.load(f)
        ent = rec.get(workload, {})
        fam = ent.get("gemm")
        here = kernel_source_hash()
        if fam is None or ent.get("kernel_source_hash") != here:
            return None, (f"profiles/hbm_traffic.json holds no PMC measurement of this build (kernel sources {here}, "
                          f"file: {ent.get('kernel_source_hash')})")
        per_step = ent.get("gemm_family", {}).get("hbm_bytes_per_step")
        if per_step is None:  # (a round-3 summary: split-K reduces counted as gemm dispatches over 2 profiled steps)
            per_step = fam["hbm_bytes_per_launch"] * fam["launches"] / 2.0
        return per_step, ent.get("source")
    except (OSError, ValueError, KeyError):
        return None, None


def refer_shapes(h, w):
    ch, out = (320, 640, 1280, 1280), [(320, h, w)]
    hh, ww = h, w
    for i, c in enumerate(ch):
        out += [(c, hh, ww)] * 2
        if i != 3:
            hh, ww = hh // 2, ww // 2
            out.append((c, hh, ww))
    return out, (1280, hh, ww)


def cpu_baseline(flavour: str, full_flops: float, frames: int, threads: int, state_dict, win_frames: int = 13):
    """Oracle (kind "port": plain-torch fp32 restatement of the reference) timed on the host cores on a bounded
    sample of the same workload: ONE UNet3D forward of the full window (CFG batch 2, 1 condition + 12 generated
    frames, same weights as the GPU run) at 256x256 px (32x32 latents) instead of 512x512 -- ~8 TFLOP, 10-30 s of CPU
    work -- extrapolated to the 512x512 forward by algorithmic FLOPs and to the 20-step denoise by x20."""
    from oracle import unet3d
    torch.set_num_threads(threads)
    cfg = unet3d.flavour_config(flavour)
    sd = {k: v.detach().float().cpu() for k, v in state_dict.items()}  # same weights as the GPU run, fp32 on the host
    g = torch.Generator().manual_seed(0)
    size = 256
    x = torch.randn(2, 4, win_frames, size // 8, size // 8, generator=g)
    ehs = torch.randn(2, 77, 768, generator=g)
    kw = dict(sample_index=torch.arange(1, win_frames), vision_conditon_frames_sample_index=torch.tensor([0]), sample_frame_rate=8)
    if flavour == "musev_referencenet":
        shapes, mid = refer_shapes(size // 8, size // 8)
        kw["down_block_refer_embs"] = [torch.randn(1, c, 1, a, b_, generator=g).repeat(2, 1, 1, 1, 1) for c, a, b_ in shapes]
        kw["mid_block_refer_emb"] = torch.randn(1, mid[0], 1, mid[1], mid[2], generator=g).repeat(2, 1, 1, 1, 1)
        kw["vision_clip_emb"] = torch.randn(2, 4, 768, generator=g)
        kw["ip_adapter_scale"] = 1.0
    # torch's intra-op pool scales badly past a few dozen threads on these small fp32 ops (measured on the 256-core host of
    # the GPU box: 16 threads 14.7 s, 64 and 256 threads several times slower and wildly variable -- one run spent minutes
    # there), so the sample runs ONCE at min(cores, 16) threads; `cores` in the JSON is that thread count
    threads = min(threads, 16)
    torch.set_num_threads(threads)
    t0 = time.time()
    with torch.no_grad():
        unet3d.unet3d_forward(sd, cfg, x, torch.tensor(951), ehs, **kw)
    dt = time.time() - t0
    sample_flops = unet_flops(size, size, win_frames, 2, "musev" if flavour == "musev" else "musev_referencenet")
    t_full = dt * full_flops / sample_flops
    return {
        "value": frames / (DENOISE_STEPS * t_full), "unit": "frames/s", "cores": threads, "kind": "port",
        "sample": f"oracle UNet3D forward, CFG batch 2, {win_frames} frames (1 cond + {win_frames - 1} generated) @{size}x{size} px "
                  f"({size // 8}x{size // 8} latents): {dt:.1f} s for {sample_flops / 1e12:.2f} TFLOP ({sample_flops / dt / 1e12:.3f} TFLOP/s); "
                  f"extrapolated by algorithmic FLOPs to the {full_flops / 1e12:.1f} TFLOP 512x512 forward x {DENOISE_STEPS} steps",
        "seconds_sample": dt,
    }


def build_unet(flavour: str, dev):
    """real architecture, seeded random fp16 weights (the same on every rank): built on the meta device and materialised +
    randomised directly on the GPU (a CPU init of 1.42 B parameters would burn minutes of GPU-box time); the philox generator
    gives every rank identical weights"""
    from musev_amd.models.layers import bump_pack_epoch
    from musev_amd.models.unet_loader import load_unet_by_name
    with torch.device("meta"):
        unet = load_unet_by_name(flavour, dtype=torch.float16)
    unet = unet.to_empty(device=dev)
    gg = torch.Generator(device=dev).manual_seed(3)
    res_out = ("conv2.weight", "to_out.0.weight", "ff.net.2.weight", "proj_out.weight", "conv4.3.weight")
    with torch.no_grad():
        for name, p in unet.named_parameters():
            if name.endswith("temporal_weight"):   # the reference zero-initialises these branches: re-randomise
                p.copy_(0.1 + 0.9 * torch.rand(p.shape, generator=gg, device=dev))
            elif p.ndim >= 2:
                gain = 0.3 if name.endswith(res_out) else 1.0
                p.copy_(torch.randn(p.shape, generator=gg, device=dev) * (gain / p[0].numel() ** 0.5))
            elif name.endswith(".weight"):
                p.copy_(1.0 + 0.1 * torch.randn(p.shape, generator=gg, device=dev))
            else:
                p.copy_(0.05 * torch.randn(p.shape, generator=gg, device=dev))
    unet.eval()
    bump_pack_epoch()
    return unet


def device_state(local_rank: int) -> dict:
    """clocks / power cap / temperature of the GPU under this rank (`rocm-smi --json`), so that a slow box explains itself in the
    line the driver keeps; best effort -- any failure is recorded, never raised"""
    import subprocess
    try:
        p = subprocess.run(["rocm-smi", "-d", str(local_rank), "--showclocks", "--showpower", "--showmaxpower", "--showtemp", "--showperflevel", "--json"],
                           capture_output=True, text=True, timeout=20)
        card = next(iter(json.loads(p.stdout).values()))
        keep = {}
        for k, v in card.items():
            kl = k.lower()
            if any(t in kl for t in ("sclk", "mclk", "fclk", "power", "temperature (sensor junction)", "temperature (sensor memory)", "performance level")):
                keep[k] = v
        return keep
    except Exception as ex:  # noqa: BLE001
        return {"error": repr(ex)[:200]}


FAMILY_ENTRIES = ("mv_gemm_f16", "mv_ffn_geglu_f16", "mv_temporal_attn_block_f16", "mv_xattn_block_f16")


class issue_family_twice:
    """context manager: every launch of the matrix family (FAMILY_ENTRIES of the C ABI) is issued TWICE with the same arguments (the
    entries are idempotent: same inputs, same outputs) -- the step's extra time is what the family's launches cost WHERE THEY RUN
    (two HIP streams, hipGraph replay), the quantity tools/gpu_insitu_cost.py tabulates for every entry"""

    def __enter__(self):
        from musev_amd import _lib
        self.lib = _lib.load()
        self.saved = {}
        for name in FAMILY_ENTRIES:
            orig = getattr(self.lib, name)
            self.saved[name] = orig

            def twice(*a, _orig=orig):
                rc = _orig(*a)
                return _orig(*a) if rc == 0 else rc
            setattr(self.lib, name, twice)   # an instance attribute of the CDLL: musev_amd.ops resolves the entry through it
        return self

    def __exit__(self, *exc):
        for name, orig in self.saved.items():
            setattr(self.lib, name, orig)
        return False


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)   # ~2.5 s of timed GPU work at config 2 (activity samplers see it)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="auto", choices=["auto", "config2", "config3", "config4", "config5", "weak"])
    ap.add_argument("--no-config4", action="store_true", help="N = 1 default run: skip the short config-4 measurement")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--dump-gemm-launches", default=None, metavar="PATH",
                    help="write the implicit-GEMM launches of one eager step, in launch order, as JSON (mode, M, N, K, geglu, ln, residual, "
                         "tile configuration, K slices, algorithmic bytes): tools/pmc_by_problem.py joins them with per-dispatch PMC rows")
    ap.add_argument("--gemm-by-problem", default=None, metavar="PATH",
                    help="roofline pass: also time every distinct recorded implicit-GEMM problem on its own (+ torch.matmul of the same "
                         "M, N, K as a library yardstick for the linear ones) and write the table as JSON")
    ap.add_argument("--size", type=int, default=None)
    ap.add_argument("--rehearse-shared-gpu", action="store_true",
                    help="REHEARSAL of the N > 1 code path on a 1-GPU box: every rank uses cuda:0, collectives over gloo "
                         "(RCCL refuses two ranks on one device).  The line is marked; its timings are not multi-GPU numbers.")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus N>1 must be launched with torch.distributed.run --nproc-per-node N")
        args.gpus = world
    if args.rehearse_shared_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    group = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.rehearse_shared_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)  # nccl == RCCL on ROCm
        group = dist.group.WORLD

    from musev_amd import ops
    from musev_amd.models.unet_loader import load_unet_by_name
    from musev_amd.pipelines.parallel_denoise import ParallelDenoiser, shard_units, group_units

    workload = args.workload
    if workload == "auto":
        workload = "config2" if world == 1 else "config4"
    flavour = "musev_referencenet" if workload in ("config3", "config5") else "musev"
    n_cond, win = 1, 12
    if args.size is None:
        args.size = 768 if workload == "config5" else 512
    T = {"config2": 12, "config3": 12, "config4": 96, "config5": 48}.get(workload, 12 if world == 1 else 8 * world)
    h = w = args.size // 8

    unet = build_unet(flavour, dev)

    g = torch.Generator().manual_seed(0)
    prompt = torch.randn(2, 77, 768, generator=torch.Generator().manual_seed(1)).to(dev)
    cond = (0.18215 * torch.randn(1, 4, n_cond, h, w, generator=torch.Generator().manual_seed(2))).to(dev)

    def make_latents(frames):
        return torch.randn(1, 4, frames, h, w, generator=torch.Generator().manual_seed(0)).to(dev)

    latents = make_latents(T)
    unet_kwargs = {}
    if flavour == "musev_referencenet":
        shapes, mid = refer_shapes(h, w)
        g4 = torch.Generator().manual_seed(4)
        unet_kwargs["down_block_refer_embs"] = [torch.randn(1, c, 1, a, b_, generator=g4).repeat(2, 1, 1, 1, 1).to(dev) for c, a, b_ in shapes]
        unet_kwargs["mid_block_refer_emb"] = torch.randn(1, mid[0], 1, mid[1], mid[2], generator=g4).repeat(2, 1, 1, 1, 1).to(dev)
        unet_kwargs["vision_clip_emb"] = torch.randn(2, 4, 768, generator=torch.Generator().manual_seed(5)).to(dev)
        unet_kwargs["ip_adapter_scale"] = 1.0
    if workload == "config5":
        # ControlNet residuals of one window ([(b t), C, h, w] on every skip + the mid block, N(0, 0.1), seed 6: SURVEY 8d row 5);
        # the ControlNet itself is a side model (section 8f), its outputs are inputs of the path.  The same tensors serve every
        # window (timing does not depend on their values).
        shapes, mid = refer_shapes(h, w)
        g6 = torch.Generator().manual_seed(6)
        n_rows = 2 * (win + n_cond)
        unet_kwargs["down_block_additional_residuals"] = [(0.1 * torch.randn(n_rows, c, a, b_, generator=g6)).to(dev, torch.float16) for c, a, b_ in shapes]
        unet_kwargs["mid_block_additional_residual"] = (0.1 * torch.randn(n_rows, mid[0], mid[1], mid[2], generator=g6)).to(dev, torch.float16)

    den = ParallelDenoiser(unet, context_frames=win, context_overlap=4, context_stride=1, context_schedule="uniform")
    n_windows = len(den.windows(T, DENOISE_STEPS))
    total = args.warmup + args.steps

    # The loop object runs `num_inference_steps` steps; time the last K of (W + K) through the per-step callback.
    marks = {}

    def sync_all():
        torch.cuda.synchronize(dev)
        if world > 1:
            torch.distributed.barrier(group)
            torch.cuda.synchronize(dev)

    def make_cb(first_timed):
        def cb(step, t, lat):
            if step + 1 == first_timed:  # end of warmup
                sync_all()
                marks["t0"] = time.perf_counter()
        return cb

    def run_steps(n_steps, cb=None, lat=None):
        # a DDIM schedule with n_steps entries: every step does identical work, which is all the timing needs
        return den(latents if lat is None else lat, prompt, num_inference_steps=n_steps, guidance_scale=3.5, condition_latents=cond,
                   motion_speed=8.0, unet_kwargs=unet_kwargs, group=group, callback=cb)

    if args.warmup == 0:
        sync_all()
        marks["t0"] = time.perf_counter()
    den.time_exchange = world > 1
    out = run_steps(total, make_cb(args.warmup))
    torch.cuda.synchronize(dev)
    local_elapsed = time.perf_counter() - marks["t0"]   # this rank's own finish (before the closing barrier): load balance / skew
    sync_all()
    elapsed = time.perf_counter() - marks["t0"]
    multi = None
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX, group=group)
        elapsed = float(tt.item())
        # diagnostics of the one line the driver keeps: per-rank step time, the exposed part of the exchange
        # ([wait for the slots' all-gathers + table reduce] between two HIP events per step, timed steps only), the ranks and the
        # distinct devices the process group really spans
        evs = den.exchange_events[-args.steps:]
        ex_ms = sum(a.elapsed_time(b) for a, b in evs) / max(len(evs), 1) if evs else 0.0
        props = torch.cuda.get_device_properties(dev)
        import zlib
        # a stable id of the DEVICE under this rank (hostname + the device's uuid, else its index): python's hash() is salted per process
        ident = zlib.crc32(f"{os.uname().nodename}:{getattr(props, 'uuid', None) or torch.cuda.current_device()}".encode()) % (1 << 31)
        mine = torch.tensor([local_elapsed * 1e3 / args.steps, ex_ms, float(ident), float(len(shard_units(n_windows, 2, world)[rank]))], dtype=torch.float64, device=dev)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        torch.distributed.all_gather(allr, mine, group=group)
        multi = {"backend": torch.distributed.get_backend(group), "ranks": world,
                 "distinct_devices": len({int(v[2].item()) for v in allr}),
                 "per_rank_ms_per_step": [round(float(v[0].item()), 3) for v in allr],
                 "per_rank_exposed_exchange_ms_per_step": [round(float(v[1].item()), 3) for v in allr],
                 "per_rank_units": [int(v[3].item()) for v in allr],
                 "exchange_payload": "fp32 predictions, one async all_gather_into_tensor per unit slot (<= 786 KB per unit and rank)"}
    den.time_exchange = False
    dev_state = device_state(local_rank) if rank == 0 else None   # right after the timed region: the clocks it ran at
    ms_per_step = elapsed * 1e3 / args.steps
    value = T / (DENOISE_STEPS * ms_per_step / 1e3)
    finite = bool(torch.isfinite(out).all())

    # ---- N = 1 default run: the 1-GPU point of the config-4 strong-scaling curve (12 windows per step), a few steps ----
    config4_n1 = None
    if world == 1 and args.workload == "auto" and not args.no_config4:
        lat96 = make_latents(96)
        k4, w4 = 3, 1
        marks4 = {}

        def cb4(step, t, lat):
            if step + 1 == w4:
                sync_all()
                marks4["t0"] = time.perf_counter()
        run_steps(w4 + k4, cb4, lat96)
        sync_all()
        ms4 = (time.perf_counter() - marks4["t0"]) * 1e3 / k4
        config4_n1 = {"workload": "config4: musev, 512x512, 96 frames, window 12 overlap 4 -> 12 windows x 2 CFG halves on 1 GPU",
                      "value": 96 / (DENOISE_STEPS * ms4 / 1e3), "unit": "frames/s", "ms_per_step": ms4, "steps": k4, "warmup": w4}

    # ---- the matrix family's cost IN THE STEP (roofline.achieved): the same timed loop once more with every launch of the family issued
    # twice, on fresh captures; extra ms per step = the family's marginal cost under the step's two streams + graph replay ----
    in_step = None
    if world == 1 and not args.no_roofline and dev.type == "cuda":   # (a CPU dry run has no launches to double: its difference is timer noise)
        den2 = ParallelDenoiser(unet, context_frames=win, context_overlap=4, context_stride=1, context_schedule="uniform")
        k2, w2 = args.steps, max(args.warmup, 2)
        marks2 = {}

        def cb2(step, t, lat):
            if step + 1 == w2:
                sync_all()
                marks2["t0"] = time.perf_counter()
        with issue_family_twice():
            den2(latents, prompt, num_inference_steps=w2 + k2, guidance_scale=3.5, condition_latents=cond, motion_speed=8.0,
                 unet_kwargs=unet_kwargs, callback=cb2)
            sync_all()
        ms_twice = (time.perf_counter() - marks2["t0"]) * 1e3 / k2
        if den2.use_graphs and dev.type == "cuda" and den2.graph_replays() == 0:
            raise SystemExit("bench.py: the doubled-family steps did not replay a hipGraph")
        in_step = {"ms_per_step_family_issued_twice": ms_twice, "family_marginal_ms_per_step": ms_twice - ms_per_step, "steps": k2, "warmup": w2}
        del den2

    # ---- whole-step algorithmic FLOPs (what each rank executes per step, summed over ranks) ----
    halves = 2
    shards = shard_units(n_windows, halves, world)
    step_flops = 0.0
    for s in shards:
        for _, hs in group_units(s):
            step_flops += unet_flops(args.size, args.size, win + n_cond, len(hs), flavour, n_vis=n_cond)
    per_rank_flops = max(sum(unet_flops(args.size, args.size, win + n_cond, len(hs), flavour, n_vis=n_cond) for _, hs in group_units(s)) for s in shards)

    # every timed step must have been a hipGraph replay (a failed capture raises in ParallelDenoiser; MUSEV_NO_GRAPH=1 is the only
    # way to get here eager, and then the line says so)
    graphs = den.use_graphs and den.graph_replays() > 0
    if den.use_graphs and dev.type == "cuda" and not graphs:
        raise SystemExit("bench.py: the timed steps did not replay a hipGraph")

    if args.dump_gemm_launches and rank == 0:
        import ctypes as C
        from musev_amd import _lib
        ops.GEMM_RECORD = []
        den.use_graphs = False
        sync_all()
        run_steps(1)
        sync_all()
        rec, ops.GEMM_RECORD = ops.GEMM_RECORD, None
        den.use_graphs = True
        rows = []
        for d, _keep, nb, *_ in rec:
            if isinstance(d, _lib.FfnDesc):   # the fused level-0 feed-forward (mv_ffn_geglu_f16): both projections in one launch
                rows.append({"mode": 3, "M": int(d.M), "N": int(d.C), "K": int(d.H), "geglu": 1, "ln": 1, "residual": 1, "colstats": 0, "cfg": -1,
                             "nsplit": 1, "algorithmic_bytes": nb})
                continue
            if isinstance(d, _lib.TsaDesc):   # the fused temporal self-attention sub-block (mv_temporal_attn_block_f16): q / k / v + to_out
                rows.append({"mode": 4, "M": int(d.B) * int(d.T) * int(d.HW), "N": int(d.C), "K": 4 * int(d.C), "geglu": 0, "ln": 1, "residual": 1,
                             "colstats": 0, "cfg": -1, "nsplit": 1, "algorithmic_bytes": nb})
                continue
            if isinstance(d, _lib.XabDesc):   # the fused text cross-attention sub-block (mv_xattn_block_f16): to_q + to_out
                rows.append({"mode": 5, "M": int(d.M), "N": int(d.C), "K": 2 * int(d.C), "geglu": 0, "ln": 1, "residual": 1,
                             "colstats": 0, "cfg": -1, "nsplit": 1, "algorithmic_bytes": nb})
                continue
            cfg, ns = C.c_int32(), C.c_int32()
            _lib.load().mv_gemm_choice(C.byref(d), C.byref(cfg), C.byref(ns))
            rows.append({"mode": int(d.mode), "M": int(d.M), "N": int(d.N), "K": int(d.K), "geglu": int(d.geglu), "ln": int(bool(d.ln_colsum)),
                         "residual": int(bool(d.residual)), "colstats": int(bool(d.colstats)), "cfg": cfg.value, "nsplit": ns.value,
                         "algorithmic_bytes": nb})
        with open(args.dump_gemm_launches, "w") as f:
            json.dump(rows, f)
        del rec

    roofline = None
    if not args.no_roofline:
        # Dominant kernel family (every mv_gemm_f16 launch: linear / conv3x3 / tconv3 + split-K reduce), measured on the device:
        # ONE step is recorded eagerly (descriptor copies, tensors kept alive), then the recorded launches are re-issued back to
        # back on one stream between ONE pair of HIP events (ops.replay_gemms) -- the host enqueues a launch in microseconds and
        # the kernels take tens to hundreds, so the elapsed time is the kernels' own plus the dispatcher's back-to-back gaps, the
        # same thing a hipGraph replay has.  (Round 2 bracketed every eager launch with its own event pair, which put the host's
        # launch latency inside each "duration": 104 us per launch on the driver's box against 88 us in the rocprofv3 trace.)
        # The record is taken with the loop's own stream setting, so the problem sizes are the ones the timed region ran (two
        # batch-1 forwards per window when the CFG halves run on two streams).
        # The recorder keeps every operand of the recorded launches alive: a whole step of a many-window workload (config 4: 12
        # windows, config 5: 768 x 768) would pin hundreds of GB.  Every window issues the same launches, so ONE window is recorded
        # (12 frames) and the per-step figures are that window's times the step's window count (`recorded_windows` says so).
        # (one GPU only: in a sharded run a rank records its own few units, and a one-window step would leave most ranks without work)
        one_window = n_windows > 1 and world == 1
        rec_lat = make_latents(win) if one_window else None
        rec_scale = n_windows if one_window else 1
        ops.GEMM_RECORD = []
        den.use_graphs = False
        sync_all()
        run_steps(1, None, rec_lat)
        sync_all()
        rec_all, ops.GEMM_RECORD = ops.GEMM_RECORD, None
        den.use_graphs = True
        from musev_amd import _lib as _mvlib
        names = {0: "linear", 1: "conv3x3", 2: "tconv3", 3: "ffn_fused", 4: "tsa_fused", 5: "xab_fused"}

        def rmode(d):
            return 3 if isinstance(d, _mvlib.FfnDesc) else 4 if isinstance(d, _mvlib.TsaDesc) else 5 if isinstance(d, _mvlib.XabDesc) else int(d.mode)
        reps = 3
        ops.replay_gemms(rec_all, 1)  # warm (clocks, code objects)
        fam_ms = ops.replay_gemms(rec_all, reps) / reps
        fam_n = len(rec_all)
        fam_flops = sum(ops.record_flops(d) for d, _k, _b, *_ in rec_all)
        fam_bytes = float(sum(nb for _d, _k, nb, *_ in rec_all))
        by_mode = {}
        for mode, nm in names.items():
            sub = [r for r in rec_all if rmode(r[0]) == mode]
            if not sub:
                continue
            ms = ops.replay_gemms(sub, reps) / reps
            fl = sum(ops.record_flops(d) for d, _k, _b, *_ in sub)
            by_mode[nm] = {"tflops": fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0, "ms_per_step": ms * rec_scale, "launches_per_step": len(sub) * rec_scale}
        # the same launches as the loop runs them: the two CFG halves' lists concurrently on two streams (info; `achieved` stays the
        # one-stream figure, which is what a rocprofv3 kernel trace -- it serialises the streams -- reproduces)
        two_stream = None
        if den.half_streams and fam_n >= 2 and dev.type == "cuda" and hasattr(ops, "replay_gemms_two_streams"):
            # the two lists = the launches each of the loop's two streams issued (every recorded launch carries its stream; with more
            # than one window per step a plain split of the list in half would mix the CFG halves of different windows)
            main_s = torch.cuda.current_stream().cuda_stream
            ha = [r for r in rec_all if r[3] == main_s]
            hb = [r for r in rec_all if r[3] != main_s]
            if not ha or not hb:
                ha, hb = rec_all[:fam_n // 2], rec_all[fam_n // 2:]
            ops.replay_gemms_two_streams(ha, hb, 1)
            ms2 = ops.replay_gemms_two_streams(ha, hb, reps) / reps
            two_stream = {"family_ms_per_step": ms2 * rec_scale, "tflops": fam_flops / (ms2 * 1e-3) / 1e12, "frac": fam_flops / (ms2 * 1e-3) / 1e12 / PEAK_MFMA_TFLOPS}
        # and the same step as ONE batch-2 forward per window on one stream (MUSEV_HALF_STREAMS=0): the launches round 2's roofline
        # timed (its timed path was the two-stream one as well, but the per-launch pass ran the batch-2 forward) -- info, for
        # round-over-round comparison
        batch2 = None
        if den.half_streams and world == 1:
            ops.GEMM_RECORD = []
            den.use_graphs, den.half_streams = False, False
            sync_all()
            run_steps(1, None, rec_lat)
            sync_all()
            rec_b2, ops.GEMM_RECORD = ops.GEMM_RECORD, None
            den.use_graphs, den.half_streams = True, True
            ops.replay_gemms(rec_b2, 1)
            ms_b2 = ops.replay_gemms(rec_b2, reps) / reps
            fl_b2 = sum(ops.record_flops(d) for d, _k, _b, *_ in rec_b2)
            batch2 = {"family_ms_per_step": ms_b2 * rec_scale, "tflops": fl_b2 / (ms_b2 * 1e-3) / 1e12, "frac": fl_b2 / (ms_b2 * 1e-3) / 1e12 / PEAK_MFMA_TFLOPS,
                      "launches_per_step": len(rec_b2) * rec_scale}
            del rec_b2
        if args.gemm_by_problem and rank == 0:
            probs = {}
            for r in rec_all:
                d = r[0]
                if rmode(d) == 3:
                    key = ("ffn_fused", int(d.M), int(d.C), int(d.H), "ln+geglu+res", 0, 0)
                elif rmode(d) == 4:
                    key = ("tsa_fused", int(d.B) * int(d.T) * int(d.HW), int(d.C), 4 * int(d.C), "ln+attn+res", 0, 0)
                elif rmode(d) == 5:
                    key = ("xab_fused", int(d.M), int(d.C), 2 * int(d.C), "ln+xattn+res", 0, 0)
                else:
                    key = (names[int(d.mode)], int(d.M), int(d.N), int(d.K), "geglu" if d.geglu else "ln" if d.ln_colsum else "res" if d.residual else "-",
                           int(bool(d.colstats)), int(bool(d.a2)))
                probs.setdefault(key, []).append(r)
            rows = []
            for key, lst in probs.items():
                one = lst[:1]
                ops.replay_gemms(one, 2)
                us = ops.replay_gemms(one, 10) / 10 * 1e3
                d = one[0][0]
                fl = ops.record_flops(d)
                row = {"mode": key[0], "M": key[1], "N": key[2], "K": key[3], "epilogue": key[4], "colstats": key[5], "two_source": key[6],
                       "launches_per_step": len(lst), "us": us, "tflops": fl / us / 1e6, "algorithmic_GBps": one[0][2] / us / 1e3,
                       "ms_per_step": us * len(lst) / 1e3}
                if key[0] == "linear":
                    a_ = torch.randn(key[1], key[3], device=dev, dtype=torch.float16)
                    w_ = torch.randn(key[2], key[3], device=dev, dtype=torch.float16)
                    for _ in range(2):
                        torch.matmul(a_, w_.t())
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(10):
                        torch.matmul(a_, w_.t())
                    e1.record()
                    e1.synchronize()
                    row["torch_matmul_us"] = e0.elapsed_time(e1) / 10 * 1e3
                    del a_, w_
                rows.append(row)
            rows.sort(key=lambda r_: -r_["ms_per_step"])
            with open(args.gemm_by_problem, "w") as f:
                json.dump(rows, f, indent=0)
        del rec_all
        iso = fam_flops / (fam_ms * 1e-3) / 1e12 if fam_ms > 0 else 0.0
        step_fam_flops = fam_flops * rec_scale
        if in_step is not None and in_step["family_marginal_ms_per_step"] > 0:
            fam_step_ms = in_step["family_marginal_ms_per_step"]
            ach = step_fam_flops / (fam_step_ms * 1e-3) / 1e12
        else:   # (N > 1 / --workload runs without the doubled leg: the isolated figure)
            fam_step_ms, ach = fam_ms * rec_scale, iso
        roofline = {
            "bound": "mfma", "kernel": "gemm2_kernel<MODE,TM,TN,WGM,WGN,SCHED> (implicit-GEMM family: linear / conv3x3 / tconv3) + ffn_geglu_kernel (the fused level-0 feed-forward: two projections per launch) + tsa_kernel (the fused level-0 temporal self-attention sub-block: q / k / v projection + to_out per launch) + xab_kernel (the fused level-0 text cross-attention sub-block: to_q + to_out per launch)",
            "achieved": ach, "peak": PEAK_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": ach / PEAK_MFMA_TFLOPS,
            # PMC bytes of the family per step / the step's API launches: the same denominator as algorithmic_bytes_per_launch
            "traffic": (measured_traffic(workload)[0] / max(fam_n * rec_scale, 1)) if measured_traffic(workload)[0] is not None else None,
            "traffic_unit": "HBM bytes per mv_gemm_f16 launch (PMC bytes of gemm2_kernel + splitk_reduce per step / launches per step)",
            "traffic_ratio": (measured_traffic(workload)[0] / (fam_bytes * rec_scale)) if (measured_traffic(workload)[0] is not None and fam_bytes > 0) else None,
            "traffic_source": measured_traffic(workload)[1],
            # (round 6) `achieved` describes the TIMED REGION: the family's algorithmic FLOPs per step / the family's
            # marginal time in the step (the same graph-replayed two-stream loop with every launch of the family issued twice, minus
            # the timed step) -- by construction below the step time.  rocprofv3 serialises the two HIP streams under kernel tracing
            # (profiles/r06b_trace_overlap.json: 3 % of the busy time overlapped, 63.9 ms per traced step against ~49 untraced), so
            # a trace cannot give the in-step figure; tools/gpu_insitu_cost.py's table (profiles/r06*_insitu_cost.json) reproduces it:
            # frac = family TFLOP per step / (sum of the marginal ms of mv_gemm_f16:* + mv_ffn_geglu_f16 + mv_temporal_attn_block_f16) / 2500.
            # `isolated` = the same launches alone on one stream (what rocprofv3's per-kernel durations reproduce; round 1-5's `achieved`).
            "method": (f"in-step: {in_step['steps']} graph-replayed steps with every launch of the family issued twice ({in_step['ms_per_step_family_issued_twice']:.3f} ms) "
                       f"minus the timed step ({ms_per_step:.3f} ms) = the family's marginal time per step; " if in_step is not None and ach != iso else "") +
                      f"isolated: one recorded step's {fam_n} mv_gemm_f16 / mv_ffn_geglu_f16 / mv_temporal_attn_block_f16 / mv_xattn_block_f16 launches re-issued back to back on one stream, "
                      f"{reps} repetitions between one HIP event pair (device time; no per-launch host gap)",
            "in_step": in_step,
            "isolated": {"achieved": iso, "frac": iso / PEAK_MFMA_TFLOPS, "family_ms_per_step": fam_ms * rec_scale, "avg_launch_ms": fam_ms / max(fam_n, 1)},
            "algorithmic_bytes_per_launch": fam_bytes / max(fam_n, 1),
            "launches_per_step": fam_n * rec_scale,
            "recorded_windows": (f"1 of {n_windows} (every window issues the same launches; per-step figures = the recorded window x {n_windows})"
                                 if one_window else f"all of this rank's units of the {n_windows}-window step"),
            "avg_launch_ms": fam_step_ms / max(fam_n * rec_scale, 1),
            "algorithmic_flops_per_launch": fam_flops / max(fam_n, 1),
            "family_tflop_per_step": step_fam_flops / 1e12,
            "family_ms_per_step": fam_step_ms,
            "by_mode": by_mode,
            "two_streams": two_stream,
            "batch2_one_stream": batch2,
            "whole_step": {"algorithmic_tflop_per_rank_step": per_rank_flops / 1e12,
                           "achieved_tflops_per_gpu": per_rank_flops / (ms_per_step * 1e-3) / 1e12,
                           "frac_of_mfma_peak": per_rank_flops / (ms_per_step * 1e-3) / 1e12 / PEAK_MFMA_TFLOPS,
                           # algorithmic = the reference's work (both CFG halves in full); with the shared front the part
                           # of a window forward in front of the first cross-attention is executed once for both halves
                           "shared_cfg_prefix": bool(getattr(den, "share_cfg_prefix", False) and den.half_streams)},
        }

    cpu, cpu_timed_out = None, False
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        # the baseline is a report, never a reason to lose the GPU number: it runs under a watchdog (a host with a
        # misbehaving thread pool once spent minutes in it) and any failure is recorded instead of raised
        import concurrent.futures
        pool = concurrent.futures.ThreadPoolExecutor(max_workers=1)
        fut = pool.submit(cpu_baseline, flavour, unet_flops(args.size, args.size, win + n_cond, 2, flavour, n_vis=n_cond), T,
                          os.cpu_count() or 1, unet.state_dict())
        try:
            cpu = fut.result(timeout=180)
        except concurrent.futures.TimeoutError:
            cpu, cpu_timed_out = {"error": "cpu_baseline sample exceeded its 180 s watchdog"}, True
        except Exception as ex:  # noqa: BLE001
            cpu = {"error": repr(ex)}
        pool.shutdown(wait=False)

    if rank == 0:
        line = {
            "metric": "denoised frames/sec @512x512, 12-frame window, 20 DDIM steps",
            "value": value, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": "strong" if workload in ("config4", "config5") else "weak", "vs_baseline": None, "dtype": "f16", "data": "synthetic",
            "config": {"workload": f"{workload}: {flavour}, {args.size}x{args.size}, {T} frames (+{n_cond} vision-condition frame per window), "
                                   f"window {win} overlap 4 -> {n_windows} window(s) x 2 CFG halves = {n_windows * 2} units over {world} GPU(s), "
                                   f"{DENOISE_STEPS} DDIM steps, guidance 3.5",
                       "frames": T, "windows": n_windows, "units_per_gpu_max": max(len(s) for s in shards),
                       # every window forward denoises 12 frames; with overlap 4 the closed-loop schedule needs N windows for
                       # 8N unique frames (N >= 2), so at fixed per-GPU work (one window per GPU) the unique-frame rate of a
                       # PERFECTLY parallel run is 8N/12 of N x the single-window rate -- the algorithm's overlap, not a loss
                       "window_frames_per_s": n_windows * win / (DENOISE_STEPS * ms_per_step / 1e3),
                       "ideal_value_vs_n1": (T / 12.0) if workload == "weak" else None,
                       # strong-scaling workloads: the speed-up this rank count can reach at best = total units / the largest shard
                       "ideal_speedup_vs_1gpu_same_workload": (n_windows * 2) / max(len(s_) for s_ in shards),
                       "weights": "seeded random fp16, SD-1.5 MuseV architecture (1.42 B parameters)",
                       "output_finite": finite, "graphs": bool(graphs), "device_state": dev_state,
                       "hw_queues": os.environ.get("GPU_MAX_HW_QUEUES")},
            "roofline": roofline, "cpu_baseline": cpu, "config4_n1": config4_n1, "multi_gpu": multi,
        }
        if args.rehearse_shared_gpu:
            line["rehearsal"] = f"{world} ranks sharing ONE GPU over gloo: exercises the N > 1 code path only, the timings are not multi-GPU numbers"
        print(json.dumps(line), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()
    if cpu_timed_out:
        sys.stdout.flush()
        os._exit(0)  # the abandoned baseline thread would otherwise keep the process alive


if __name__ == "__main__":
    main()
