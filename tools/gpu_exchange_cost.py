"""What the multi-GPU exchange path costs per step on ONE GPU: config 4 (96 frames, 12 windows x 2 CFG halves = 24 units) run (a) with
local accumulation and (b) through the exchange path -- send-slot copies, one async RCCL all_gather_into_tensor per slot in a 1-rank
`nccl` group, the table-driven gather-reduce (ParallelDenoiser.always_exchange) -- i.e. everything a rank of an N-GPU run does except
the wire time.  Usage: python tools/gpu_exchange_cost.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def main():
    import torch.distributed as dist
    import bench
    from musev_amd.pipelines.parallel_denoise import ParallelDenoiser
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", HSA_ENABLE_IPC_MODE_LEGACY="0")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    unet = bench.build_unet("musev", dev)
    lat = torch.randn(1, 4, 96, 64, 64, generator=torch.Generator().manual_seed(0)).to(dev)
    cond = (0.18215 * torch.randn(1, 4, 1, 64, 64, generator=torch.Generator().manual_seed(2))).to(dev)
    prompt = torch.randn(2, 77, 768, generator=torch.Generator().manual_seed(1)).to(dev)
    out = {}
    for name, exch in (("local", False), ("exchange", True), ("local", False), ("exchange", True)):
        den = ParallelDenoiser(unet)
        den.always_exchange = exch
        marks = {}

        def cb(step, t, l_):
            if step == 1:
                torch.cuda.synchronize()
                marks["t0"] = time.perf_counter()
        den(lat, prompt, num_inference_steps=5, guidance_scale=3.5, condition_latents=cond, callback=cb,
            group=dist.group.WORLD if exch else None)
        torch.cuda.synchronize()
        out.setdefault(name, []).append((time.perf_counter() - marks["t0"]) * 1e3 / 3)
    a, b = min(out["local"]), min(out["exchange"])
    print(f"config 4 on one GPU, ms per step: local accumulation {a:.1f} | exchange path (24 slots: copy + 1-rank RCCL all-gather + table reduce) {b:.1f} "
          f"| +{b - a:.1f} ms = {100 * (b - a) / a:.2f} % ({(b - a) / 24 * 1e3:.0f} us per unit)")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
