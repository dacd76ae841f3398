"""-m gpu, LAST in collection order on purpose: GPU checks written after round 1's GPU budget was spent (the Euler-discrete
loop, its fused step kernel, the full-size batch-independence property, ReferenceNet2D on HIP kernels).  They exercise code
whose host side is tested on the CPU against the oracle and the reference's recorded outputs with the kernels emulated
(tests/test_emulated_wiring.py, tests/test_oracle_golden.py, tests/test_parallel_sharding.py) but which had not run on a GPU
when they were written; keeping them at the end means a surprise here cannot hide the results of the tests above under
`pytest -x`."""
import os

import numpy as np
import pytest
import torch

from test_pipeline_gpu import _run

pytestmark = pytest.mark.gpu


def test_cfg_affine_step_kernel():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from kernel_cases import case_cfg_affine_step
    res = case_cfg_affine_step()
    torch.cuda.synchronize()
    assert res["ok"], res


def test_loop_parity_euler_first_steps():
    """the reference's default scheduler (EulerDiscreteScheduler, pipeline_controlnet_predictor.py:258-261): first 2 steps
    of the 20-step schedule.  Euler latents live in sigma-scaled space (initial noise x sigma_max = 14.6), so the bound is
    the north-star 1e-2 on the MODEL-INPUT scale: |delta latent| / sqrt(sigma^2 + 1) < 1e-2 at the sigma reached."""
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from musev_amd.schedulers import EulerDiscreteScheduler
    want, got, got2 = _run("musev", 8, 6, 2, 2, scheduler="euler")
    assert torch.equal(got, got2), "the loop must be deterministic"
    s = EulerDiscreteScheduler()
    s.set_timesteps(20)
    scale = (float(s.sigmas[2]) ** 2 + 1) ** 0.5
    err = (got - want).abs().max().item() / scale
    assert err < 1e-2, f"|delta latent|max / sqrt(sigma^2+1) = {err}"
    assert torch.equal(got[:, :, 0], want[:, :, 0])


def _random_full_unet(flavour="musev", seed=3):
    """the 1.42 B-parameter SD-1.5 MuseV architecture with seeded random fp16 weights, built directly on the GPU"""
    from musev_amd.models.layers import bump_pack_epoch
    from musev_amd.models.unet_loader import load_unet_by_name
    dev = torch.device("cuda", 0)
    with torch.device("meta"):
        unet = load_unet_by_name(flavour, dtype=torch.float16)
    unet = unet.to_empty(device=dev)
    gg = torch.Generator(device=dev).manual_seed(seed)
    res_out = ("conv2.weight", "to_out.0.weight", "ff.net.2.weight", "proj_out.weight", "conv4.3.weight")
    with torch.no_grad():
        for name, p in unet.named_parameters():
            if name.endswith("temporal_weight"):
                p.copy_(0.1 + 0.9 * torch.rand(p.shape, generator=gg, device=dev))
            elif p.ndim >= 2:
                p.copy_(torch.randn(p.shape, generator=gg, device=dev) * ((0.3 if name.endswith(res_out) else 1.0) / p[0].numel() ** 0.5))
            elif name.endswith(".weight"):
                p.copy_(1.0 + 0.1 * torch.randn(p.shape, generator=gg, device=dev))
            else:
                p.copy_(0.05 * torch.randn(p.shape, generator=gg, device=dev))
    unet.eval()
    bump_pack_epoch()
    return unet


def test_full_size_batch_independence():
    """BASELINE config-2 size (1.42 B parameters, 64x64 latents, 12 + 1 frames): size-independent property of the network --
    the two CFG halves never interact inside the UNet, so one batch-2 forward must equal the two batch-1 forwards (this is
    also what the two-stream schedule of ParallelDenoiser relies on).  Different batch sizes take different GEMM tile
    shapes and GroupNorm row splits, so the comparison is to 5e-3 (half the parity bound), not bitwise."""
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    unet = _random_full_unet()
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(0)
    t, h, w = 13, 64, 64
    x = torch.randn(2, 4, t, h, w, generator=g).to(dev)
    ehs = torch.randn(2, 77, 768, generator=g).to(dev)
    kw = dict(sample_index=torch.arange(1, t, device=dev), vision_conditon_frames_sample_index=torch.tensor([0], device=dev),
              sample_frame_rate=8, return_dict=False)
    ts = torch.tensor(601, device=dev)
    both = unet(x, ts, encoder_hidden_states=ehs, **kw)[0].float()
    again = unet(x, ts, encoder_hidden_states=ehs, **kw)[0].float()
    assert torch.equal(both, again), "the forward must be deterministic"
    assert torch.isfinite(both).all()
    for i in range(2):
        one = unet(x[i:i + 1], ts, encoder_hidden_states=ehs[i:i + 1], **kw)[0].float()
        err = (one - both[i:i + 1]).abs().max().item()
        assert err < 5e-3, f"CFG half {i}: batch-1 vs batch-2 forward differ by {err}"
    assert (both[0] - both[1]).abs().max().item() > 1e-3, "the halves see different prompts and must differ"


def test_referencenet_matches_reference_golden_and_oracle():
    """ReferenceNet2D on HIP kernels against the oracle and against the feature maps recorded from the
    reference's own ReferenceNet2D (tests/golden/reference_referencenet_hipw.npz)."""
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from golden_cases import REFNET_CASES, refnet_case_inputs
    from oracle import referencenet as oref
    from musev_amd.models.referencenet import load_referencenet_by_name
    case = REFNET_CASES["hipw"]
    cfg = oref.referencenet_config(**case["arch"])
    sd = oref.init_state_dict(cfg, case["weight_seed"])
    x, t, ehs = refnet_case_inputs(case, cfg)
    net = load_referencenet_by_name("musev_referencenet", sd, **case["arch"]).to("cuda")
    down, mid, sa = net(x.to("cuda"), t.to("cuda"), encoder_hidden_states=ehs.to("cuda"), num_frames=case["t"], return_ndim=5)
    torch.cuda.synchronize()
    assert sa is None
    gold = np.load(os.path.join(os.path.dirname(__file__), "golden", "reference_referencenet_hipw.npz"))
    with torch.no_grad():
        odown, omid = oref.referencenet_forward(sd, cfg, x, t, ehs, num_frames=case["t"])
    assert len(down) == len(odown)
    for i, d in enumerate(down):
        want = torch.from_numpy(gold[f"down{i}"])
        assert d.shape == want.shape
        assert (d.float().cpu() - want).abs().max().item() < 1e-2, f"down{i} vs reference"
        assert (d.float().cpu() - odown[i]).abs().max().item() < 1e-2, f"down{i} vs oracle"
    assert (mid.float().cpu() - torch.from_numpy(gold["mid"])).abs().max().item() < 1e-2


def test_conv3x3_direct_kernel():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from kernel_cases import case_conv3x3_direct
    res = case_conv3x3_direct()
    torch.cuda.synchronize()
    assert res["ok"], res


def test_poseguider_matches_reference_golden():
    """PoseGuider on mv_conv3x3_direct_f16 against the output recorded from the reference's own class in
    the configuration scripts/inference/video2video.py builds (320 channels, levels 16/32/96/256)"""
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from golden_cases import POSEGUIDER_CASES, poseguider_case_inputs
    from oracle import poseguider as opg
    from musev_amd.models.controlnet import PoseGuider
    for name, c in POSEGUIDER_CASES.items():
        sd = opg.init_state_dict(opg.param_shapes(c["emb"], c["cond"], c["ch"]), c["weight_seed"])
        net = PoseGuider.from_pretrained(sd, conditioning_embedding_channels=c["emb"], conditioning_channels=c["cond"],
                                         block_out_channels=c["ch"]).half().to("cuda")
        got = net(poseguider_case_inputs(c).to("cuda"))
        torch.cuda.synchronize()
        want = torch.from_numpy(np.load(os.path.join(os.path.dirname(__file__), "golden", f"reference_poseguider_{name}.npz"))["out"])
        assert got.shape == want.shape
        err = (got.float().cpu() - want).abs().max().item()
        assert err < 1e-2, f"{name}: |delta|max = {err}"


_CN_ARCH = dict(block_out_channels=(320, 640, 640), layers_per_block=1,
                down_block_types=("CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "DownBlock2D"))


def test_controlnet_matches_oracle():
    """ControlNetModel on HIP kernels against oracle/controlnet.py -- whose top-level composition restates
    the published diffusers algorithm (unpinned), while its encoder and conditioning embedding are the pinned
    oracle/referencenet.py and oracle/poseguider.py"""
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from oracle import controlnet as ocn
    from musev_amd.models.controlnet import ControlNetModel
    ccfg = ocn.controlnet_config(**_CN_ARCH)
    csd = ocn.init_state_dict(ccfg, 5)
    g = torch.Generator().manual_seed(93)
    n, h, w = 4, 16, 24
    frames = torch.randn(n, 4, h, w, generator=g)
    text = torch.randn(n, 77, 768, generator=g)
    pose = torch.rand(n, 3, 8 * h, 8 * w, generator=g) * 2 - 1
    net = ControlNetModel(**_CN_ARCH)
    net.load_state_dict(csd, strict=True)
    net = net.half().eval().to("cuda")
    for guess, scale in ((False, 1.0), (True, 0.7)):
        odown, omid = ocn.controlnet_forward(csd, ccfg, frames, torch.tensor(601), text, pose, conditioning_scale=scale, guess_mode=guess)
        down, mid = net(frames.to("cuda"), torch.tensor(601, device="cuda"), text.to("cuda"), pose.to("cuda"),
                        conditioning_scale=scale, guess_mode=guess, return_dict=False)
        torch.cuda.synchronize()
        assert len(down) == len(odown)
        for i, (d, o) in enumerate(zip(down, odown)):
            assert d.shape == o.shape
            assert (d.float().cpu() - o).abs().max().item() < 1e-2, f"guess={guess} residual {i}"
        assert (mid.float().cpu() - omid).abs().max().item() < 1e-2


def test_loop_with_controlnet_first_steps():
    """the per-window ControlNet call inside the loop (hipGraph-captured together with the UNet forward, two streams)"""
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from test_pipeline_gpu import ARCH
    from oracle import controlnet as ocn
    from oracle import pipeline as opipe
    from oracle import unet3d
    from musev_amd.models.controlnet import ControlNetModel
    from musev_amd.models.unet_loader import load_unet_by_name
    from musev_amd.pipelines.parallel_denoise import ParallelDenoiser
    cfg = unet3d.flavour_config("musev", **ARCH)
    sd = unet3d.init_state_dict(cfg, 3)
    cn_arch = dict(block_out_channels=(320, 640), layers_per_block=1, down_block_types=("CrossAttnDownBlock2D", "DownBlock2D"))
    ccfg = ocn.controlnet_config(**cn_arch)
    csd = ocn.init_state_dict(ccfg, 5)
    g = torch.Generator().manual_seed(0)
    T, win, ov, h, w = 8, 6, 2, 8, 8
    latents = torch.randn(1, 4, T, h, w, generator=g)
    cond = 0.18215 * torch.randn(1, 4, 1, h, w, generator=g)
    prompt = torch.randn(2, 77, 768, generator=g)
    pose = torch.rand(1, 3, 1 + T, 8 * h, 8 * w, generator=g) * 2 - 1
    kw = dict(num_inference_steps=20, max_steps=2, guidance_scale=3.5, motion_speed=8.0, controlnet_conditioning_scale=0.8)
    want = opipe.denoise_loop(lambda x, t, e, **k: unet3d.unet3d_forward(sd, cfg, x, t, e, **k), latents, prompt,
                              context_frames=win, context_overlap=ov, condition_latents=cond, control_image=pose,
                              controlnet_fn=lambda f, t, tx, ci, s, gm: ocn.controlnet_forward(csd, ccfg, f, t, tx, ci, conditioning_scale=s, guess_mode=gm),
                              **kw)
    dev = torch.device("cuda", 0)
    net = ControlNetModel(**cn_arch)
    net.load_state_dict(csd, strict=True)
    net = net.half().eval().to(dev)
    unet = load_unet_by_name("musev", sd_unet_model=sd, dtype=torch.float16, **ARCH).to(dev)
    den = ParallelDenoiser(unet, context_frames=win, context_overlap=ov)
    got = den(latents.to(dev), prompt.to(dev), condition_latents=cond.to(dev), controlnet=net, control_image=pose.to(dev), **kw)
    got2 = den(latents.to(dev), prompt.to(dev), condition_latents=cond.to(dev), controlnet=net, control_image=pose.to(dev), **kw)
    torch.cuda.synchronize()
    assert torch.equal(got, got2), "the loop must be deterministic"
    err = (got.float().cpu() - want).abs().max().item()
    assert err < 1e-2, f"|delta latent|max = {err}"
