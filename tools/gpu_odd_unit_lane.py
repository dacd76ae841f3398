"""What a rank with an ODD number of units pays per step, on ONE GPU: config 4 (96 frames, 12 windows x 2 CFG halves = 24 units) seen
from rank r of an 8-rank run -- 3 units = one two-half window + one lone half -- with local accumulation in place of the exchange
(timing only).  Compares the groups run one after the other with the lone half's graph replayed on a third stream concurrently with
the pair (ParallelDenoiser.odd_unit_lane), and checks that the latents are bit-identical.  Usage: python tools/gpu_odd_unit_lane.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def main():
    import bench
    from musev_amd.pipelines import parallel_denoise as pd
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    unet = bench.build_unet("musev", dev)
    lat = torch.randn(1, 4, 96, 64, 64, generator=torch.Generator().manual_seed(0)).to(dev)
    cond = (0.18215 * torch.randn(1, 4, 1, 64, 64, generator=torch.Generator().manual_seed(2))).to(dev)
    prompt = torch.randn(2, 77, 768, generator=torch.Generator().manual_seed(1)).to(dev)
    real = pd.shard_units
    keep = []  # (a captured graph must not be destroyed while another capture is under way: keep every denoiser until the end)
    for rank in (0, 1):
        pd.shard_units = lambda n, hv, world, rank=rank: [real(n, hv, 8)[rank]]
        outs, times = {}, {}
        for lane in (False, True, False, True):
            den = pd.ParallelDenoiser(unet)
            den.odd_unit_lane = lane
            keep.append(den)
            marks = {}

            def cb(step, t, l_):
                if step == 1:
                    torch.cuda.synchronize()
                    marks["t0"] = time.perf_counter()
            out = den(lat, prompt, num_inference_steps=8, guidance_scale=3.5, condition_latents=cond, callback=cb)
            torch.cuda.synchronize()
            times.setdefault(lane, []).append((time.perf_counter() - marks["t0"]) * 1e3 / 6)
            outs[lane] = out
        a, b = min(times[False]), min(times[True])
        print(f"rank {rank} of 8 (units {[(u.window, u.half) for u in real(12, 2, 8)[rank]]}): ms per step one after the other {a:.2f} | "
              f"lone half on the lane stream {b:.2f} | {100 * (b - a) / a:+.1f} % | latents bit-identical: {bool(torch.equal(outs[False], outs[True]))}",
              flush=True)
    pd.shard_units = real


if __name__ == "__main__":
    main()
