"""The side models either side of the denoise loop: VAE decoder with temporally chunked decode, IP-Adapter
ImageProjModel, the predictor's multi-shot loop.  Oracle: oracle/vae.py (PARITY UNPINNED: AutoencoderKL / ImageProjModel live in
un-vendored packages; upstream semantics restated) and oracle/pipeline.multi_shot_loop (restates
pipeline_controlnet_predictor.py:643-745).

CPU (-m "not gpu"): key / shape inventory, the host-side wiring of the HIP modules through the emulated kernels (tests/emu_ops.py,
which also enforces every kernel's argument contract), chunk-independence of the decode, the shot loop with kernel test doubles.
GPU (-m gpu): the same modules on the real kernels against the oracle, incl. the real SD-1.5 VAE widths."""
import pytest
import torch

import emu_ops
import fake_ops

SMALL = dict(block_out_channels=(64, 128), layers_per_block=1)
REAL = dict()


def _vae_pair(arch, seed, dev="cpu"):
    from oracle import vae as ovae
    from musev_amd.models.vae import AutoencoderKL
    cfg = ovae.vae_config(**arch)
    sd = ovae.init_state_dict(ovae.decoder_param_shapes(cfg), seed)
    m = AutoencoderKL(**arch)
    m.load_state_dict(sd, strict=True)   # exactly the upstream decoder keys: nothing missing, nothing unexpected
    return cfg, sd, m.to(torch.float16).to(dev).eval()


def test_vae_state_dict_accepts_full_and_legacy_checkpoints():
    from oracle import vae as ovae
    from musev_amd.models.vae import AutoencoderKL
    cfg = ovae.vae_config(**SMALL)
    sd = ovae.init_state_dict(ovae.decoder_param_shapes(cfg), 1)
    assert "decoder.mid_block.attentions.0.to_q.weight" in sd and "decoder.up_blocks.0.upsamplers.0.conv.weight" in sd
    assert "decoder.up_blocks.1.upsamplers.0.conv.weight" not in sd   # the last up block has no upsampler
    # a full checkpoint (encode half present) with the pre-0.17 attention names and 1x1-conv shaped attention weights
    legacy = {}
    for k, v in sd.items():
        for new, old in (("to_q", "query"), ("to_k", "key"), ("to_v", "value"), ("to_out.0", "proj_attn")):
            if ".attentions.0." + new + "." in k:
                k = k.replace("." + new + ".", "." + old + ".")
                if k.endswith("weight"):
                    v = v[:, :, None, None]
        legacy[k] = v
    legacy["encoder.conv_in.weight"] = torch.zeros(128, 3, 3, 3)
    legacy["quant_conv.weight"] = torch.zeros(8, 8, 1, 1)
    m = AutoencoderKL(**SMALL)
    m.load_state_dict(legacy, strict=True)
    assert torch.equal(m.state_dict()["decoder.mid_block.attentions.0.to_q.weight"], sd["decoder.mid_block.attentions.0.to_q.weight"])


def test_vae_decode_wiring_on_emulated_kernels(monkeypatch):
    from oracle import vae as ovae
    emu_ops.install(monkeypatch)
    cfg, sd, m = _vae_pair(SMALL, 2)
    m._device_check = False
    z = torch.randn(3, 4, 8, 12, generator=torch.Generator().manual_seed(3))
    want = ovae.vae_decode(sd, cfg, z)
    got = m.decode(z)[0]
    assert got.shape == want.shape == (3, 3, 16, 24)
    err = (got - want).abs().max().item()
    assert err < 2e-2 * max(1.0, want.abs().max().item()), err


def test_decode_latents_is_independent_of_the_chunking(monkeypatch):
    from oracle import vae as ovae
    from musev_amd.pipelines import video
    emu_ops.install(monkeypatch)
    cfg, sd, m = _vae_pair(SMALL, 4)
    m._device_check = False
    lat = 0.18215 * torch.randn(1, 4, 5, 8, 8, generator=torch.Generator().manual_seed(5))
    want = ovae.decode_latents(sd, cfg, lat, decoder_t_segment=2)
    assert torch.allclose(want, ovae.decode_latents(sd, cfg, lat, decoder_t_segment=200), atol=1e-5)
    a = video.decode_latents(m, lat, decoder_t_segment=2)
    b = video.decode_latents(m, lat, decoder_t_segment=200)
    monkeypatch.setattr(video, "_MAX_CALL_BYTES", 16 * 16 * 64 * 2)    # forces one frame per kernel call
    c = video.decode_latents(m, lat, decoder_t_segment=3)
    # (not bit-identical: GroupNorm's row splits and the GEMM split factor are chosen from the call's size)
    assert a.shape == want.shape == (1, 3, 5, 16, 16) and (a - b).abs().max().item() < 2e-3 and (a - c).abs().max().item() < 2e-3
    assert float(a.min()) >= 0.0 and float(a.max()) <= 1.0
    assert (a - want).abs().max().item() < 1e-2


def test_image_proj_wiring_on_emulated_kernels(monkeypatch):
    from oracle import vae as ovae
    from musev_amd.models.vae import ImageProjModel
    emu_ops.install(monkeypatch)
    sd = ovae.init_state_dict(ovae.image_proj_shapes(), 6)
    m = ImageProjModel()
    m.load_state_dict(sd, strict=True)
    m = m.to(torch.float16)
    m._device_check = False
    x = torch.randn(3, 1024, generator=torch.Generator().manual_seed(7))
    want = ovae.image_proj(sd, x)
    got = m(x)
    assert got.shape == want.shape == (3, 4, 768)
    assert (got.float() - want).abs().max().item() < 1e-2
    # the unconditional tokens of classifier-free guidance are the projection of zeros (pipeline_controlnet.py:736-774)
    assert (m(torch.zeros(1, 1024)).float() - ovae.image_proj(sd, torch.zeros(1, 1024))).abs().max().item() < 1e-2


def _loop_doubles(monkeypatch):
    from musev_amd import ops
    from musev_amd.pipelines.parallel_denoise import ParallelDenoiser
    for name in ("window_gather", "window_scatter_add", "window_units_reduce", "cfg_ddim_step", "cfg_affine_step"):
        monkeypatch.setattr(ops, name, getattr(fake_ops, name))
    monkeypatch.setattr(ParallelDenoiser, "_device_check", False)


@pytest.mark.parametrize("fix", [False, True])
def test_multi_shot_loop_matches_the_oracle(monkeypatch, fix):
    """3 shots of 10 frames (2 windows each): shot k+1 is conditioned on the last latent frame of shot k, the re-inserted condition
    frame of shots 1.. is dropped -> 11 + 10 + 10 frames"""
    from oracle import pipeline as opipe
    from musev_amd.pipelines.parallel_denoise import ParallelDenoiser
    from musev_amd.pipelines.video import multi_shot_denoise
    _loop_doubles(monkeypatch)
    g = torch.Generator().manual_seed(8)
    noises = [torch.randn(1, 4, 10, 4, 4, generator=g) for _ in range(3)]
    cond = torch.randn(1, 4, 1, 4, 4, generator=g)
    prompt = torch.randn(2, 7, 16, generator=g)
    den = ParallelDenoiser(fake_ops.FakeUNet(), context_frames=6, context_overlap=2)
    seen = []
    lat, vid = multi_shot_denoise(den, lambda i: noises[i], prompt, condition_latents=cond, max_batch_num=3, fix_condition_images=fix,
                                  num_inference_steps=4, guidance_scale=3.5, on_shot=lambda i, x: seen.append(tuple(x.shape)))
    want = opipe.multi_shot_loop(fake_ops.FakeUNet().nchw, noises, prompt, cond, fix_condition_images=fix, num_inference_steps=4,
                                 guidance_scale=3.5, context_frames=6, context_overlap=2, motion_speed=8.0)
    assert vid is None and lat.shape == want.shape == (1, 4, 31, 4, 4)
    assert seen == [(1, 4, 11, 4, 4), (1, 4, 10, 4, 4), (1, 4, 10, 4, 4)]
    assert (lat - want).abs().max().item() < 5e-3


# ---- GPU ---------------------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("arch,n,h,w", [("small", 3, 8, 12), ("real", 2, 32, 32)])
def test_vae_decode_on_the_gpu(arch, n, h, w):
    """HIP VAE decoder against the oracle: small widths on a non-square latent, and the real SD-1.5 VAE (128 / 256 / 512 / 512, mid
    attention with one 512-wide head over 1024 tokens) at 256x256 px.  Tolerance: 2e-2 of the output range (fp16 activations through
    ~30 layers; the decoded image is then quantised to 8 bits = 3.9e-3 per level)."""
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from oracle import vae as ovae
    cfg, sd, m = _vae_pair(SMALL if arch == "small" else REAL, 11, dev="cuda")
    z = torch.randn(n, 4, h, w, generator=torch.Generator().manual_seed(12))
    with torch.no_grad():
        want = ovae.vae_decode(sd, cfg, z)
    got = m.decode(z.cuda())[0].float().cpu()
    torch.cuda.synchronize()
    err = (got - want).abs().max().item()
    print(f"vae decode {arch}: |delta|max = {err:.3e}, |want|max = {want.abs().max().item():.3f}")
    assert torch.isfinite(got).all()
    assert err < 2e-2 * max(1.0, want.abs().max().item()), err


@pytest.mark.gpu
def test_decode_latents_chunks_and_image_proj_on_the_gpu():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from oracle import vae as ovae
    from musev_amd.models.vae import ImageProjModel
    from musev_amd.pipelines import video
    cfg, sd, m = _vae_pair(SMALL, 13, dev="cuda")
    lat = 0.18215 * torch.randn(1, 4, 7, 8, 8, generator=torch.Generator().manual_seed(14))
    with torch.no_grad():
        want = ovae.decode_latents(sd, cfg, lat)
    a = video.decode_latents(m, lat.cuda(), decoder_t_segment=3)
    b = video.decode_latents(m, lat.cuda(), decoder_t_segment=200)
    torch.cuda.synchronize()
    assert (a - b).abs().max().item() < 2e-3, "the temporal chunking must not change the result (beyond summation order)"
    assert (a.float().cpu() - want).abs().max().item() < 1e-2
    psd = ovae.init_state_dict(ovae.image_proj_shapes(), 15)
    pm = ImageProjModel()
    pm.load_state_dict(psd, strict=True)
    pm = pm.to(torch.float16).cuda()
    x = torch.randn(2, 1024, generator=torch.Generator().manual_seed(16))
    got = pm(x.cuda()).float().cpu()
    assert (got - ovae.image_proj(psd, x)).abs().max().item() < 1e-2


@pytest.mark.gpu
def test_multi_shot_generation_on_the_gpu():
    """two shots through the HIP UNet (2-level SD-1.5-width net), first steps of the schedule, decoded by the HIP VAE: latents against
    the oracle shot loop, frames finite and in [0, 1]"""
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from oracle import pipeline as opipe
    from oracle import unet3d
    from musev_amd.models.unet_loader import load_unet_by_name
    from musev_amd.pipelines.parallel_denoise import ParallelDenoiser
    from musev_amd.pipelines.video import multi_shot_denoise
    arch = dict(block_out_channels=(320, 640), layers_per_block=1, down_block_types=("CrossAttnDownBlock3D", "DownBlock3D"),
                up_block_types=("UpBlock3D", "CrossAttnUpBlock3D"))
    cfg = unet3d.flavour_config("musev", **arch)
    sd = unet3d.init_state_dict(cfg, 3)
    g = torch.Generator().manual_seed(17)
    noises = [torch.randn(1, 4, 6, 8, 8, generator=g) for _ in range(2)]
    cond = 0.18215 * torch.randn(1, 4, 1, 8, 8, generator=g)
    prompt = torch.randn(2, 77, 768, generator=g)
    kw = dict(num_inference_steps=20, max_steps=2, guidance_scale=3.5)
    with torch.no_grad():
        want = opipe.multi_shot_loop(lambda x, t, e, **k: unet3d.unet3d_forward(sd, cfg, x, t, e, **k), noises, prompt, cond,
                                     context_frames=6, context_overlap=2, motion_speed=8.0, **kw)
    dev = torch.device("cuda", 0)
    unet = load_unet_by_name("musev", sd_unet_model=sd, dtype=torch.float16, **arch).to(dev)
    den = ParallelDenoiser(unet, context_frames=6, context_overlap=2)
    _, _, vae = _vae_pair(SMALL, 18, dev="cuda")
    lat, vid = multi_shot_denoise(den, lambda i: noises[i].to(dev), prompt.to(dev), condition_latents=cond.to(dev), max_batch_num=2,
                                  vae=vae, decoder_t_segment=4, motion_speed=8.0, **kw)
    torch.cuda.synchronize()
    assert lat.shape == want.shape == (1, 4, 13, 8, 8) and vid.shape == (1, 3, 13, 16, 16)
    assert (lat.float().cpu() - want).abs().max().item() < 1e-2
    assert torch.isfinite(vid).all() and float(vid.min()) >= 0.0 and float(vid.max()) <= 1.0
