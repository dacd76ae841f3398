"""Cost of integration Level 1: the reference's own denoise loop calling ``unet.forward`` once per window -- the
5-D ``b c t h w`` API with its two layout conversions, eager launches (no hipGraph), both CFG halves in one batch-2 call on one
stream -- next to what ParallelDenoiser does for the same window (rows API, captured graph, two streams).  Config 2 sizes.
Usage: python tools/gpu_level1_timing.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def main():
    import bench
    from musev_amd.pipelines.parallel_denoise import ParallelDenoiser
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    unet = bench.build_unet("musev", dev)
    g = torch.Generator().manual_seed(0)
    x = torch.randn(2, 4, 13, 64, 64, generator=g).to(dev, torch.float16)
    ehs = torch.randn(2, 77, 768, generator=g).to(dev)
    t = torch.tensor(601, device=dev)
    kw = dict(sample_index=torch.arange(1, 13, device=dev), vision_conditon_frames_sample_index=torch.tensor([0], device=dev), sample_frame_rate=8)

    def level1():
        return unet(x, t, encoder_hidden_states=ehs, return_dict=False, **kw)[0]
    for _ in range(3):
        level1()
    torch.cuda.synchronize()
    n = 10
    t0 = time.perf_counter()
    for _ in range(n):
        level1()
    torch.cuda.synchronize()
    ms1 = (time.perf_counter() - t0) * 1e3 / n
    den = ParallelDenoiser(unet)
    lat = torch.randn(1, 4, 12, 64, 64, generator=g).to(dev)
    cond = (0.18215 * torch.randn(1, 4, 1, 64, 64, generator=g)).to(dev)
    marks = {}

    def cb(step, t_, l_):
        if step == 4:
            torch.cuda.synchronize()
            marks["t0"] = time.perf_counter()
    den(lat, ehs, num_inference_steps=25, guidance_scale=3.5, condition_latents=cond, callback=cb)
    torch.cuda.synchronize()
    ms2 = (time.perf_counter() - marks["t0"]) * 1e3 / 20
    print(f"level 1: unet.forward (b c t h w, eager, batch 2, one stream) {ms1:.2f} ms per window forward | "
          f"ParallelDenoiser step (rows, hipGraph, two streams, incl. loop glue) {ms2:.2f} ms | ratio {ms1 / ms2:.2f}")


if __name__ == "__main__":
    main()
