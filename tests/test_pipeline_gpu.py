"""-m gpu: the sliding-window parallel-denoise loop on HIP kernels (musev_amd.pipelines.ParallelDenoiser) against the
oracle loop (oracle/pipeline.py, restating pipeline_controlnet.py:1832-2156) on identical seeds.

Tolerance: |delta latent|max < 1e-2 (north-star bound) over the first steps of the real 20-step DDIM schedule.  A whole
20-step run amplifies the UNet's fp16-level prediction error (~3e-3, test_model_gpu.py) by CFG (x ~4) and by the DDIM
recursion (L2 gain ~3.9), beyond 1e-2 for ANY fp16 implementation -- the per-step bound is the meaningful one, and the
loop glue itself is exact (fp32; checked bit-level in kernel_cases.case_window_loop and the full-size properties below)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

ARCH = dict(block_out_channels=(320, 640), layers_per_block=1,
            down_block_types=("CrossAttnDownBlock3D", "DownBlock3D"), up_block_types=("UpBlock3D", "CrossAttnUpBlock3D"))


def _run(flavour, T, win, ov, steps, with_cond=True, seed=0, scheduler="ddim"):
    from oracle import pipeline as opipe
    from oracle import unet3d
    from musev_amd.models.unet_loader import load_unet_by_name
    from musev_amd.pipelines.parallel_denoise import ParallelDenoiser
    cfg = unet3d.flavour_config(flavour, **ARCH)
    sd = unet3d.init_state_dict(cfg, 3)
    g = torch.Generator().manual_seed(seed)
    h = w = 8
    latents = torch.randn(1, 4, T, h, w, generator=g)
    sched = None
    if scheduler == "euler":
        from musev_amd.schedulers import EulerDiscreteScheduler
        sched = EulerDiscreteScheduler()
        sched.set_timesteps(20)
        latents = latents * sched.init_noise_sigma  # prepare_latents scales the initial noise (pipeline_controlnet.py)
    cond = 0.18215 * torch.randn(1, 4, 1, h, w, generator=g) if with_cond else None
    prompt = torch.randn(2, 77, 768, generator=g)
    kw = dict(num_inference_steps=20, max_steps=steps, guidance_scale=3.5, condition_latents=cond)

    def oracle_unet(x, t, ehs, **k):
        return unet3d.unet3d_forward(sd, cfg, x, t, ehs, **k)

    want = opipe.denoise_loop(oracle_unet, latents, prompt, motion_speed=8.0, context_frames=win, context_overlap=ov,
                              scheduler=scheduler, **kw)
    dev = torch.device("cuda", 0)
    unet = load_unet_by_name(flavour, sd_unet_model=sd, dtype=torch.float16, **ARCH).to(dev)
    den = ParallelDenoiser(unet, scheduler=sched, context_frames=win, context_overlap=ov)
    kw["condition_latents"] = None if cond is None else cond.to(dev)
    got = den(latents.to(dev), prompt.to(dev), motion_speed=8.0, **kw)
    got2 = den(latents.to(dev), prompt.to(dev), motion_speed=8.0, **kw)
    torch.cuda.synchronize()
    return want, got.float().cpu(), got2.float().cpu()


@pytest.mark.parametrize("T,win,ov,steps", [(8, 6, 2, 2), (12, 6, 2, 3), (5, 6, 2, 2)])
def test_loop_parity_first_steps(T, win, ov, steps):
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    want, got, got2 = _run("musev", T, win, ov, steps)
    assert got.shape == want.shape
    assert torch.equal(got, got2), "the loop must be deterministic"
    err = (got - want).abs().max().item()
    assert err < 1e-2, f"|delta latent|max = {err}"
    # the vision-condition frame is re-inserted untouched in front (pipeline_controlnet.py:2149-2156)
    assert torch.equal(got[:, :, 0], want[:, :, 0])


def test_full_size_window_average_property():
    """Config-4 sizes (96 frames, 64x64 latents, window 12 overlap 4 -> 12 windows): when every window predicts the same
    per-frame value, scatter-add / coverage average must reproduce it exactly, and CFG with equal halves + DDIM with
    alpha_t == alpha_prev must leave the latents unchanged (size-independent properties of the loop glue)."""
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from musev_amd import ops
    from musev_amd.pipelines.context import prepare_global_context
    dev = torch.device("cuda", 0)
    c, T, hw, n_cond = 4, 96, 64 * 64, 1
    wins = [w_[0] for w_ in prepare_global_context("uniform", 20, T, 12, 1, 4, 1)]
    assert len(wins) == 12 and wins[-1] == [88, 89, 90, 91, 92, 93, 94, 95, 0, 1, 2, 3]
    g = torch.Generator().manual_seed(5)
    frame_val = torch.randn(2, c, T, hw, generator=g).to(dev)  # the "prediction" of frame f, the same from every window
    acc = torch.zeros(2, c, T, hw, device=dev)
    cnt = torch.zeros(T, device=dev)
    for wd in wins:
        idx = torch.tensor(wd, dtype=torch.int32, device=dev)
        for half in range(2):
            rows = torch.zeros((n_cond + 12) * hw, c, device=dev)
            rows[n_cond * hw:] = frame_val[half][:, idx.long()].permute(1, 2, 0).reshape(-1, c)
            ops.window_scatter_add(rows, idx, n_cond, 1, half, acc, cnt, half == 0)
    counts = torch.tensor([2, 2, 2, 2, 1, 1, 1, 1] * 12, dtype=torch.float32, device=dev)
    assert torch.equal(cnt, counts)
    avg = acc / cnt[None, None, :, None]
    assert (avg - frame_val).abs().max().item() < 1e-6
    # equal CFG halves + alpha_t == alpha_prev: x_prev == x up to fp32 rounding, for any guidance scale
    lat = torch.randn(c, T, hw, generator=g).to(dev)
    acc2 = torch.stack([acc[0], acc[0]]).contiguous()
    x = lat.clone()
    ops.cfg_ddim_step(x, acc2, cnt, 3.5, 0.5, 0.5)
    assert (x - lat).abs().max().item() < 1e-5
