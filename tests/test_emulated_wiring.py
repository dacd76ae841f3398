"""CPU checks of the product's HOST SIDE with the kernels emulated (tests/emu_ops.py): the HIP-backed modules -- weight
packing, QKV fusion, segment descriptors, geometry, conditioning rows, skip bookkeeping, the denoise loop -- run end to
end against the oracle and against the outputs recorded from the reference's own source, with no GPU.

What this does and does not prove: the kernels themselves are NOT exercised here (that is `-m gpu`, through the C ABI
on an MI355X).  The emulation is first tied to the same torch reference expressions the kernels are verified against
(test_emulation_matches_kernel_references), so a wiring error in a module shows up here on the CPU, and a kernel error
shows up in tests/test_kernels_gpu.py on the GPU."""
import os

import numpy as np
import pytest
import torch

import emu_ops
from golden_cases import (POSEGUIDER_CASES, REFNET_CASES, UNET_CASES, case_config, case_inputs, check_written_refer_embs, poseguider_case_inputs,
                          refnet_case_inputs)

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
TOL = 1e-2  # the north-star bound; fp16 storage between emulated ops gives the same error level as the kernels (~3e-3)


@pytest.fixture
def emulated(monkeypatch):
    emu_ops.install(monkeypatch)
    torch.set_num_threads(min(8, os.cpu_count() or 1))
    return monkeypatch


def _widths(emulated, arch):
    """full argument contract (head dims 40/80/160, 64-channel conv sources) for the widths the kernels serve; the
    layout / alignment contract only for the 1/5-width cases"""
    emulated.setattr(emu_ops, "STRICT_WIDTHS", arch["block_out_channels"][0] % 320 == 0)


def _cpu(model):
    model._device_check = False  # instance attribute: the class default (refuse non-HIP tensors) stays
    return model


# ---- 1. the emulation against the reference expressions of tests/kernel_cases.py ----------------------------------------
def test_emulation_matches_kernel_references(emulated):
    import kernel_cases as kc
    emulated.setattr(kc, "DEV", "cpu")
    emulated.setattr(emu_ops, "STRICT_WIDTHS", False)  # small shapes: the arithmetic is what is compared here
    cases = [
        lambda: kc.case_gemm(M=200, N=320, K=640), lambda: kc.case_gemm(M=130, N=160, K=448, two_src=True),
        lambda: kc.case_gemm(M=64, N=64, K=64, epilogue=False), kc.case_gemm_silu, lambda: kc.case_gemm_geglu(M=77, C=64),
        lambda: kc.case_conv3x3(n=2, h=8, w=12, c1=32, cout=64),
        lambda: kc.case_conv3x3(n=2, h=8, w=12, c1=32, c2=16, cout=64),
        lambda: kc.case_conv3x3(n=2, h=8, w=12, c1=32, cout=64, stride=2),
        lambda: kc.case_conv3x3(n=2, h=7, w=9, c1=32, cout=64, stride=2),
        lambda: kc.case_conv3x3(n=2, h=8, w=12, c1=32, cout=64, upsample=True),
        lambda: kc.case_tconv3(b=2, t=5, hw=12, c=64),
        lambda: kc.case_groupnorm(n=3, rows=50, c1=64), lambda: kc.case_groupnorm(n=2, rows=50, c1=64, c2=32, silu=False),
        lambda: kc.case_layernorm(rows=99, c=64),
        lambda: kc.case_attention_self(d=40, b=2, t=3, lq=20, cond_idx=1), lambda: kc.case_attention_cross(d=80, nb=6, t=3, lq=13),
        lambda: kc.case_temporal_attention(b=2, t=5, hw=7, d=40), kc.case_geglu, kc.case_conv_in_out, kc.case_timestep_embedding,
        kc.case_layout_and_misc, kc.case_upsample_nearest, kc.case_window_loop, kc.case_cfg_affine_step, kc.case_conv3x3_direct,
    ]
    for fn in cases:
        res = fn()
        assert res["ok"], res


# ---- 2. UNet3DConditionModel wiring: vs the reference's recorded outputs and vs the oracle -------------------------------
@pytest.mark.parametrize("name", list(UNET_CASES))
def test_unet_wiring_matches_reference_golden(name, emulated):
    """every golden case, including the 1/5-width ones whose head dims (8/16/32) the HIP attention kernels do not
    serve -- the module wiring is width-independent"""
    from oracle import unet3d
    from musev_amd.models.unet_loader import load_unet_by_name
    case = UNET_CASES[name]
    _widths(emulated, case["arch"])
    cfg = case_config(case)
    sd = unet3d.init_state_dict(cfg, case["weight_seed"])
    x, t, ehs, kw = case_inputs(case, cfg)
    want = torch.from_numpy(np.load(os.path.join(GOLDEN, f"reference_unet_{name}.npz"))["out"])
    model = _cpu(load_unet_by_name(case["flavour"], sd_unet_model=sd, dtype=torch.float16, **case["arch"]))
    got = model(x, t, encoder_hidden_states=ehs, return_dict=False, **kw)[0]
    assert got.shape == want.shape and torch.isfinite(got).all()
    err = (got.float() - want).abs().max().item()
    assert err < TOL, f"{name}: |delta|max = {err}"
    if case.get("refer_self_write"):   # "write" mode: the caller's list now holds every spatial block's self-attention input
        check_written_refer_embs(name, kw["refer_self_attn_emb"], np.load(os.path.join(GOLDEN, f"reference_unet_{name}.npz")), TOL)
    # second call: cached K/V projections / packed weights must give the same result
    again = model(x, t, encoder_hidden_states=ehs, return_dict=False, **kw)[0]
    assert torch.equal(got, again)


def test_unet_rejects_cpu_tensors_by_default():
    from oracle import unet3d
    from musev_amd.models.unet_loader import load_unet_by_name
    case = UNET_CASES["musev_narrow_2d"]
    cfg = case_config(case)
    model = load_unet_by_name("musev", sd_unet_model=unet3d.init_state_dict(cfg, 3), dtype=torch.float16, **case["arch"])
    x, t, ehs, kw = case_inputs(case, cfg)
    with pytest.raises(RuntimeError, match="no CPU path"):
        model(x, t, encoder_hidden_states=ehs, **kw)


# ---- 3. ReferenceNet2D wiring and the ReferenceNet -> UNet hand-over ------------------------------------------------------
@pytest.mark.parametrize("name", list(REFNET_CASES))
def test_referencenet_wiring_matches_reference_golden(name, emulated):
    from oracle import referencenet as oref
    from musev_amd.models.referencenet import load_referencenet_by_name
    case = REFNET_CASES[name]
    _widths(emulated, case["arch"])
    cfg = oref.referencenet_config(**case["arch"])
    sd = oref.init_state_dict(cfg, case["weight_seed"])
    x, t, ehs = refnet_case_inputs(case, cfg)
    net = _cpu(load_referencenet_by_name("musev_referencenet", sd, **case["arch"]))
    down, mid, sa = net(x, t, encoder_hidden_states=ehs, num_frames=case["t"], return_ndim=5)
    assert sa is None
    gold = np.load(os.path.join(GOLDEN, f"reference_referencenet_{name}.npz"))
    n_down = len([k for k in gold.files if k.startswith("down")])
    assert len(down) == n_down
    for i, d in enumerate(down):
        want = torch.from_numpy(gold[f"down{i}"])
        assert d.shape == want.shape, (i, d.shape, want.shape)
        err = (d.float() - want).abs().max().item()
        assert err < TOL, f"{name} down{i}: {err}"
    assert (mid.float() - torch.from_numpy(gold["mid"])).abs().max().item() < TOL
    # return_ndim = 4: the same features as (b t) c h w (referencenet.py:1018-1033)
    down4, mid4, _ = net(x, t, encoder_hidden_states=ehs, num_frames=case["t"], return_ndim=4)
    b = x.shape[0] // case["t"]
    for d5, d4 in zip(list(down) + [mid], list(down4) + [mid4]):
        assert torch.equal(d5.permute(0, 2, 1, 3, 4).reshape(b * case["t"], *d4.shape[1:]), d4)


def test_referencenet_features_feed_the_unet(emulated):
    """ReferenceNet2D -> UNet3DConditionModel(down_block_refer_embs=, mid_block_refer_emb=) on the module side against the
    same chain in the oracle (pipeline_controlnet.py:867-964 -> :2045-2067)."""
    from oracle import referencenet as oref
    from oracle import unet3d
    from musev_amd.models.referencenet import load_referencenet_by_name
    from musev_amd.models.unet_loader import load_unet_by_name
    case = UNET_CASES["refnet_narrow"]
    _widths(emulated, case["arch"])
    cfg = case_config(case)
    sd = unet3d.init_state_dict(cfg, case["weight_seed"])
    x, t, ehs, kw = case_inputs(case, cfg)
    rcfg = oref.referencenet_config(block_out_channels=cfg["block_out_channels"])
    rsd = oref.init_state_dict(rcfg, 21)
    g = torch.Generator().manual_seed(77)
    ref_img = torch.randn(1, 4, case["h"], case["w"], generator=g)  # one reference image, shared by both CFG halves
    rtext = torch.randn(1, 77, cfg["cross_attention_dim"], generator=g)
    with torch.no_grad():
        odown, omid = oref.referencenet_forward(rsd, rcfg, ref_img, torch.tensor(0), rtext, num_frames=1)
    okw = dict(kw, down_block_refer_embs=[d.repeat(case["b"], 1, 1, 1, 1) for d in odown],
               mid_block_refer_emb=omid.repeat(case["b"], 1, 1, 1, 1))
    want = unet3d.unet3d_forward(sd, cfg, x, t, ehs, **okw)

    net = _cpu(load_referencenet_by_name("musev_referencenet", rsd, block_out_channels=cfg["block_out_channels"]))
    down, mid, _ = net(ref_img, torch.tensor(0), encoder_hidden_states=rtext, num_frames=1, return_ndim=5)
    hkw = dict(kw, down_block_refer_embs=[d.repeat(case["b"], 1, 1, 1, 1) for d in down],
               mid_block_refer_emb=mid.repeat(case["b"], 1, 1, 1, 1))
    model = _cpu(load_unet_by_name(case["flavour"], sd_unet_model=sd, dtype=torch.float16, **case["arch"]))
    got = model(x, t, encoder_hidden_states=ehs, return_dict=False, **hkw)[0]
    err = (got.float() - want).abs().max().item()
    assert err < TOL, f"|delta|max = {err}"


# ---- 4. the denoise loop over the real module (not the closed-form fake UNet of test_parallel_sharding.py) -----------------
@pytest.mark.parametrize("scheduler,hw", [("ddim", (8, 8)), ("euler", (8, 8)), ("ddim", (10, 6))])   # (10 x 6: forward_upsample_size in the loop)
def test_denoise_loop_over_the_module(scheduler, hw, emulated):
    from oracle import pipeline as opipe
    from oracle import unet3d
    from musev_amd.models.unet_loader import load_unet_by_name
    from musev_amd.pipelines.parallel_denoise import ParallelDenoiser
    arch = UNET_CASES["musev_narrow"]["arch"]
    _widths(emulated, arch)
    cfg = unet3d.flavour_config("musev", **arch)
    sd = unet3d.init_state_dict(cfg, 3)
    g = torch.Generator().manual_seed(0)
    T, win, ov = 8, 6, 2
    h, w = hw
    latents = torch.randn(1, 4, T, h, w, generator=g)
    sched = None
    if scheduler == "euler":
        from musev_amd.schedulers import EulerDiscreteScheduler
        sched = EulerDiscreteScheduler()
        sched.set_timesteps(20)
        latents = latents * sched.init_noise_sigma
    cond = 0.18215 * torch.randn(1, 4, 1, h, w, generator=g)
    prompt = torch.randn(2, 77, 768, generator=g)
    kw = dict(num_inference_steps=20, max_steps=2, guidance_scale=3.5, condition_latents=cond, motion_speed=8.0)
    want = opipe.denoise_loop(lambda x, t, e, **k: unet3d.unet3d_forward(sd, cfg, x, t, e, **k), latents, prompt,
                              context_frames=win, context_overlap=ov, scheduler=scheduler, **kw)
    unet = _cpu(load_unet_by_name("musev", sd_unet_model=sd, dtype=torch.float16, **arch))
    den = ParallelDenoiser(unet, scheduler=sched, context_frames=win, context_overlap=ov)
    den._device_check = False
    got = den(latents, prompt, **kw)
    assert got.shape == want.shape
    # Euler works on sigma-scaled latents (init_noise_sigma ~ 14.6): the bound scales with the latent magnitude
    scale = max(1.0, float(want.abs().max()) / 4.0)
    err = (got.float() - want).abs().max().item()
    assert err < TOL * scale, f"|delta latent|max = {err} (scale {scale})"
    assert torch.equal(got[:, :, 0].float(), want[:, :, 0])


@pytest.mark.parametrize("name", ["musev_hipw", "refnet_hipw", "musev_hipw_refer_self"])
def test_shared_cfg_prefix_is_bit_identical_to_two_forwards(name, emulated):
    """models/runtime.PrefixMemo: the first CFG half's forward records everything in front of the first text cross-attention, the
    second half's forward replays it.  Checked on the emulated kernels: both halves' outputs are bit-identical to two independent
    batch-1 forwards; the `musev` flavour really shares (the timestep / frame embeddings and their projections, conv_in, transformer_in, the first resnet / temporal conv, the first
    block's self-attention and query); flavours / calls whose front depends on per-half tensors (ReferenceNet features,
    refer_self_attn_emb) share only what precedes them."""
    from oracle import unet3d
    from musev_amd.models.runtime import PrefixMemo
    from musev_amd.models.unet_loader import load_unet_by_name
    case = UNET_CASES[name]
    _widths(emulated, case["arch"])
    cfg = case_config(case)
    sd = unet3d.init_state_dict(cfg, case["weight_seed"])
    x, t, ehs, ckw = case_inputs(case, cfg)
    unet = _cpu(load_unet_by_name(case["flavour"], sd_unet_model=sd, dtype=torch.float16, **case["arch"]))
    if case.get("refer_self"):
        unet.insert_spatial_self_attn_idx()
    b, c, tt, h, w = x.shape
    x = x[:1].repeat(2, 1, 1, 1, 1)  # the loop's CFG duplication: both halves see the same latents
    rows = x.permute(0, 2, 3, 4, 1).reshape(b * tt * h * w, c).to(torch.float16).contiguous()
    half = rows.shape[0] // 2

    from musev_amd.pipelines.parallel_denoise import _PER_HALF_KWARGS, ParallelDenoiser

    def forward(i, memo):
        kw = {k: (ParallelDenoiser._slice_half(v, [i], 2) if k in _PER_HALF_KWARGS else v) for k, v in ckw.items()}
        return unet.forward_rows(rows[i * half:(i + 1) * half], 1, tt, h, w, t, ehs[i:i + 1], prefix_memo=memo, **kw)

    plain = [forward(0, None), forward(1, None)]
    memo = PrefixMemo()
    splits = []
    memo.on_split = lambda: splits.append(len(memo.store))
    shared = [forward(0, memo)]
    assert memo.closed and len(splits) == 1
    shared.append(forward(1, memo.replay()))
    assert memo.closed
    for a, b_ in zip(plain, shared):
        assert torch.equal(a, b_)
    if name == "musev_hipw":
        assert memo.hits == len(memo.store) >= 5, (memo.hits, list(memo.store))
    elif name == "refnet_hipw":
        assert memo.hits == len(memo.store) == 2   # the embeddings and conv_in only: the ReferenceNet features come per half right behind


@pytest.mark.parametrize("one_half_per_forward", [True, False])
def test_denoise_loop_slices_refer_self_attn_emb_per_cfg_half(one_half_per_forward, emulated):
    """refer_self_attn_emb ("read", attention.py:261-289) is a list of [2 b, c, t, h, w] tensors batched over the CFG halves: the loop
    must hand each batch-1 half-forward (the default: one half per stream) its own slice, like the other per-half conditioning
    (the keyword was missing from the per-half list and the batch-1 forward raised)."""
    from oracle import pipeline as opipe
    from oracle import unet3d
    from musev_amd.models.unet_loader import load_unet_by_name
    from musev_amd.pipelines import parallel_denoise as pdn
    case = UNET_CASES["musev_hipw_refer_self"]
    _widths(emulated, case["arch"])
    cfg = case_config(case)
    sd = unet3d.init_state_dict(cfg, case["weight_seed"])
    _x, _t, _ehs, ckw = case_inputs(case, cfg)
    refer = [r + 0.25 * torch.arange(2, dtype=r.dtype).view(2, 1, 1, 1, 1) for r in ckw["refer_self_attn_emb"]]  # halves differ
    g = torch.Generator().manual_seed(3)
    T, win, ov, h, w = 6, 4, 2, 16, 16
    latents = torch.randn(1, 4, T, h, w, generator=g)
    cond = 0.18215 * torch.randn(1, 4, 1, h, w, generator=g)
    prompt = torch.randn(2, 77, 768, generator=g)
    kw = dict(num_inference_steps=20, max_steps=1, guidance_scale=3.5, condition_latents=cond, motion_speed=8.0)
    ukw = dict(refer_self_attn_emb=refer, refer_self_attn_emb_mode="read")
    want = opipe.denoise_loop(lambda x, t, e, **k: unet3d.unet3d_forward(sd, cfg, x, t, e, **k), latents, prompt,
                              context_frames=win, context_overlap=ov, unet_kwargs=ukw, **kw)
    unet = _cpu(load_unet_by_name("musev", sd_unet_model=sd, dtype=torch.float16, **case["arch"]))
    unet.insert_spatial_self_attn_idx()
    if one_half_per_forward:  # what the two-stream default, a sharded rank's lone half and the odd-unit lane do on the GPU
        emulated.setattr(pdn, "group_units", lambda units: [(u.window, [u.half]) for u in units])
    den = pdn.ParallelDenoiser(unet, context_frames=win, context_overlap=ov)
    den._device_check = False
    got = den(latents, prompt, unet_kwargs=ukw, **kw)
    err = (got.float() - want).abs().max().item()
    assert err < TOL, f"|delta latent|max = {err}"


def test_referencenet_cfg_glue_feeds_distinct_halves(emulated):
    """pipeline glue (pipeline_controlnet.py:838-859, 867-964): the CFG halves share the reference latents but see different
    cross-attention tokens ([proj(zeros), proj(clip(image))]), so ReferenceNet runs on batch 2 and each half of the UNet gets
    its own features -- module side through musev_amd.pipelines.conditioning against the same chain in the oracle."""
    from oracle import referencenet as oref
    from oracle import unet3d
    from musev_amd.models.referencenet import load_referencenet_by_name
    from musev_amd.models.unet_loader import load_unet_by_name
    from musev_amd.pipelines.conditioning import cfg_refer_image_latents, get_referencenet_emb
    case = UNET_CASES["refnet_narrow"]
    _widths(emulated, case["arch"])
    cfg = case_config(case)
    sd = unet3d.init_state_dict(cfg, case["weight_seed"])
    x, t, ehs, kw = case_inputs(case, cfg)
    rcfg = oref.referencenet_config(block_out_channels=cfg["block_out_channels"])
    rsd = oref.init_state_dict(rcfg, 21)
    g = torch.Generator().manual_seed(78)
    ref_lat = torch.randn(1, 4, case["h"], case["w"], generator=g)             # (b t) = 1 reference image
    ip_tokens = torch.randn(2, 4, cfg["cross_attention_dim"], generator=g)     # [uncond, cond] image-prompt tokens
    both = cfg_refer_image_latents(ref_lat, 1, True)
    assert both.shape[0] == 2 and torch.equal(both[0], both[1])
    with torch.no_grad():
        odown, omid = oref.referencenet_forward(rsd, rcfg, both, torch.tensor(0), ip_tokens, num_frames=1)
    want = unet3d.unet3d_forward(sd, cfg, x, t, ehs, **dict(kw, down_block_refer_embs=odown, mid_block_refer_emb=omid))

    net = _cpu(load_referencenet_by_name("musev_referencenet", rsd, block_out_channels=cfg["block_out_channels"]))
    down, mid, sa = get_referencenet_emb(net, both, 1, ip_tokens, prompt_embeds=ehs)
    assert sa is None and down[0].shape[0] == 2
    assert (down[1][0].float() - down[1][1].float()).abs().max().item() > 1e-3, "the halves see different tokens"
    model = _cpu(load_unet_by_name(case["flavour"], sd_unet_model=sd, dtype=torch.float16, **case["arch"]))
    got = model(x, t, encoder_hidden_states=ehs, return_dict=False, **dict(kw, down_block_refer_embs=down, mid_block_refer_emb=mid))[0]
    err = (got.float() - want).abs().max().item()
    assert err < TOL, f"|delta|max = {err}"
    assert get_referencenet_emb(None, both, 1, ip_tokens, None) == (None, None, None)
    with pytest.raises(ValueError):
        get_referencenet_emb(net, both, 1, ip_tokens[:1], None)                # token batch != reference batch


# ---- 5. PoseGuider wiring -----------------------------------------------------------------------------------
@pytest.mark.parametrize("name", list(POSEGUIDER_CASES))
def test_poseguider_wiring_matches_reference_golden(name, emulated):
    from oracle import poseguider as opg
    from musev_amd.models.controlnet import PoseGuider
    c = POSEGUIDER_CASES[name]
    sd = opg.init_state_dict(opg.param_shapes(c["emb"], c["cond"], c["ch"]), c["weight_seed"])
    x = poseguider_case_inputs(c)
    want = torch.from_numpy(np.load(os.path.join(GOLDEN, f"reference_poseguider_{name}.npz"))["out"])
    fresh = _cpu(PoseGuider(c["emb"], c["cond"], c["ch"]).half())
    assert sorted(fresh.state_dict()) == sorted(sd) and all(tuple(fresh.state_dict()[k].shape) == tuple(v.shape) for k, v in sd.items())
    assert (fresh(x) == 0).all(), "conv_out is zero-initialised: an untrained PoseGuider adds nothing"
    net = _cpu(PoseGuider.from_pretrained(sd, conditioning_embedding_channels=c["emb"], conditioning_channels=c["cond"],
                                          block_out_channels=c["ch"]).half())
    got = net(x)
    assert got.shape == want.shape and got.dtype == x.dtype
    err = (got.float() - want).abs().max().item()
    assert err < TOL, f"{name}: |delta|max = {err}"
    with pytest.raises(ValueError):
        net(x[:, :, 0])                       # not b c f h w
    with pytest.raises(RuntimeError, match="no CPU path"):
        PoseGuider(c["emb"], c["cond"], c["ch"])(x)


# ---- 6. ControlNetModel wiring (top-level composition is unpinned, see oracle/controlnet.py) ------------------
@pytest.mark.parametrize("guess_mode,scale", [(False, 1.0), (False, 0.6), (True, 1.0)])
def test_controlnet_wiring_matches_oracle_and_feeds_the_unet(guess_mode, scale, emulated):
    from oracle import controlnet as ocn
    from oracle import unet3d
    from musev_amd.models.controlnet import ControlNetModel
    from musev_amd.models.unet_loader import load_unet_by_name
    case = UNET_CASES["musev_narrow"]
    _widths(emulated, case["arch"])
    cfg = case_config(case)
    ccfg = ocn.controlnet_config(block_out_channels=cfg["block_out_channels"])
    csd = ocn.init_state_dict(ccfg, 5)
    x, t, ehs, kw = case_inputs(case, cfg)
    b, _, tt, h, w = x.shape
    g = torch.Generator().manual_seed(91)
    frames = x.permute(0, 2, 1, 3, 4).reshape(b * tt, -1, h, w)                       # "b c t h w -> (b t) c h w" (:1236-1238)
    text = ehs.repeat_interleave(tt, dim=0)                                            # align_repeat_tensor_single_dim (:1242-1246)
    pose = torch.rand(b * tt, 3, 8 * h, 8 * w, generator=g) * 2 - 1
    odown, omid = ocn.controlnet_forward(csd, ccfg, frames, t, text, pose, conditioning_scale=scale, guess_mode=guess_mode)
    assert len(odown) == ocn.n_residuals(ccfg) == 12

    net = ControlNetModel(block_out_channels=cfg["block_out_channels"])
    assert sorted(net.state_dict()) == sorted(csd)
    assert all(tuple(net.state_dict()[k].shape) == tuple(v.shape) for k, v in csd.items())
    net.load_state_dict(csd, strict=True)
    net = _cpu(net.half().eval())
    net.controlnet_cond_embedding._device_check = False
    down, mid = net(frames, t, text, pose, conditioning_scale=scale, guess_mode=guess_mode, return_dict=False)
    assert len(down) == len(odown)
    for i, (d, o) in enumerate(zip(down, odown)):
        assert d.shape == o.shape
        assert (d.float() - o).abs().max().item() < TOL, f"residual {i}"
    assert (mid.float() - omid).abs().max().item() < TOL
    out = net(frames, t, text, pose, conditioning_scale=scale, guess_mode=guess_mode)
    assert torch.equal(out.mid_block_res_sample, mid)
    with pytest.raises(NotImplementedError):
        net(frames, t, text, pose, controlnet_cond_latents=frames)
    if guess_mode or scale != 1.0:
        return
    # the residuals into the UNet (pipeline_controlnet.py:2045-2051), module chain vs oracle chain
    sd = unet3d.init_state_dict(cfg, case["weight_seed"])
    want = unet3d.unet3d_forward(sd, cfg, x, t, ehs, **dict(kw, down_block_additional_residuals=odown, mid_block_additional_residual=omid))
    model = _cpu(load_unet_by_name(case["flavour"], sd_unet_model=sd, dtype=torch.float16, **case["arch"]))
    got = model(x, t, encoder_hidden_states=ehs, return_dict=False,
                **dict(kw, down_block_additional_residuals=down, mid_block_additional_residual=mid))[0]
    assert (got.float() - want).abs().max().item() < TOL


@pytest.mark.parametrize("guess_mode", [False, True])
def test_denoise_loop_with_controlnet(guess_mode, emulated):
    """the per-window ControlNet call inside the loop (get_controlnet_emb + the control-frame gather, :1202-1291, 1947-1976)
    with control_guidance_end cutting the last step: module loop vs oracle loop"""
    from oracle import controlnet as ocn
    from oracle import pipeline as opipe
    from oracle import unet3d
    from musev_amd.models.controlnet import ControlNetModel
    from musev_amd.models.unet_loader import load_unet_by_name
    from musev_amd.pipelines.parallel_denoise import ParallelDenoiser
    arch = UNET_CASES["musev_narrow"]["arch"]
    _widths(emulated, arch)
    cfg = unet3d.flavour_config("musev", **arch)
    sd = unet3d.init_state_dict(cfg, 3)
    ccfg = ocn.controlnet_config(block_out_channels=cfg["block_out_channels"])
    csd = ocn.init_state_dict(ccfg, 5)
    g = torch.Generator().manual_seed(0)
    T, win, ov, h, w = 8, 6, 2, 8, 8
    latents = torch.randn(1, 4, T, h, w, generator=g)
    cond = 0.18215 * torch.randn(1, 4, 1, h, w, generator=g)
    prompt = torch.randn(2, 77, 768, generator=g)
    pose = torch.rand(1, 3, 1 + T, 8 * h, 8 * w, generator=g) * 2 - 1
    # first 3 steps of the 20-step schedule (the per-step bound, see tests/test_pipeline_gpu.py); keep = [1, 1, 0, ...]
    kw = dict(num_inference_steps=20, max_steps=3, guidance_scale=3.5, condition_latents=cond, motion_speed=8.0, control_image=pose,
              controlnet_conditioning_scale=0.8, control_guidance_end=0.1, guess_mode=guess_mode)
    calls = []

    def cn_fn(frames, t, text, cimg, scale, guess):
        calls.append(float(scale))
        return ocn.controlnet_forward(csd, ccfg, frames, t, text, cimg, conditioning_scale=scale, guess_mode=guess)

    want = opipe.denoise_loop(lambda x, t, e, **k: unet3d.unet3d_forward(sd, cfg, x, t, e, **k), latents, prompt,
                              context_frames=win, context_overlap=ov, controlnet_fn=cn_fn, **kw)
    assert 0.0 in calls and 0.8 in calls
    plain = opipe.denoise_loop(lambda x, t, e, **k: unet3d.unet3d_forward(sd, cfg, x, t, e, **k), latents, prompt,
                               context_frames=win, context_overlap=ov, **{k: v for k, v in kw.items() if k in
                                                                         ("num_inference_steps", "max_steps", "guidance_scale", "condition_latents", "motion_speed")})
    assert (want - plain).abs().max().item() > 5e-2, "the ControlNet must matter for the check to mean anything"
    net = ControlNetModel(block_out_channels=cfg["block_out_channels"])
    net.load_state_dict(csd, strict=True)
    net = _cpu(net.half().eval())
    unet = _cpu(load_unet_by_name("musev", sd_unet_model=sd, dtype=torch.float16, **arch))
    den = ParallelDenoiser(unet, context_frames=win, context_overlap=ov)
    den._device_check = False
    got = den(latents, prompt, controlnet=net, **kw)
    err = (got.float() - want).abs().max().item()
    assert err < TOL, f"|delta latent|max = {err}"


# ---- 7. MusevControlNetPipeline: the reference's __call__ surface over the HIP loop ----------------------------------------------
def test_pipeline_call_keeps_the_reference_keyword_list():
    """every keyword of the reference's MusevControlNetPipeline.__call__ (pipeline_controlnet.py:1295-1420; names, order and
    literal defaults recorded from its source by make_reference_goldens.py --signature) exists here with the same default, so a
    caller written against the reference (the predictor, pipeline_controlnet_predictor.py:643-745) needs no edit"""
    import inspect
    import json
    from musev_amd.pipelines.pipeline_controlnet import MusevControlNetPipeline
    ref = json.load(open(os.path.join(GOLDEN, "reference_pipeline_signature.json")))["args"]
    sig = inspect.signature(MusevControlNetPipeline.__call__)
    params = [p for n, p in sig.parameters.items() if n != "self"]
    names = [p.name for p in params]
    assert names[:len(ref)] == [a["name"] for a in ref], "keyword order must match the reference (positional callers)"
    for a, p in zip(ref, params):
        if a.get("required"):
            assert p.default is inspect.Parameter.empty, a
        else:
            assert p.default == a["default"], (a, p.default)


def test_pipeline_call_runs_the_loop_like_the_oracle(emulated):
    """prompt_embeds / negative_prompt_embeds, explicit latents and condition latents, DDIM, 2 steps, two windows: the adapter's
    output latents equal the oracle loop's; strings without a text encoder and unsupported branches raise"""
    from oracle import pipeline as opipe
    from oracle import unet3d
    from musev_amd.models.unet_loader import load_unet_by_name
    from musev_amd.pipelines.parallel_denoise import ParallelDenoiser
    from musev_amd.pipelines.pipeline_controlnet import MusevControlNetPipeline
    arch = UNET_CASES["musev_narrow"]["arch"]
    _widths(emulated, arch)
    cfg = unet3d.flavour_config("musev", **arch)
    sd = unet3d.init_state_dict(cfg, 3)
    g = torch.Generator().manual_seed(0)
    T, win, ov, h, w = 8, 6, 2, 8, 8
    latents = torch.randn(1, 4, T, h, w, generator=g)
    cond = 0.18215 * torch.randn(1, 4, 1, h, w, generator=g)
    prompt = torch.randn(2, 77, 768, generator=g)
    want = opipe.denoise_loop(lambda x, t, e, **k: unet3d.unet3d_forward(sd, cfg, x, t, e, **k), latents, prompt, num_inference_steps=2,
                              guidance_scale=3.5, condition_latents=cond, motion_speed=8.0, context_frames=win, context_overlap=ov)
    unet = _cpu(load_unet_by_name("musev", sd_unet_model=sd, dtype=torch.float16, **arch))
    emulated.setattr(ParallelDenoiser, "_device_check", False)
    pipe = MusevControlNetPipeline(unet=unet)
    out = pipe(video_length=T, prompt_embeds=prompt[1:], negative_prompt_embeds=prompt[:1], latents=latents, condition_latents=cond,
               num_inference_steps=2, guidance_scale=3.5, context_frames=win, context_overlap=ov, output_type="latent")
    assert out.videos is None and out.latents.shape == want.shape
    # two 500-timestep DDIM jumps of a random network blow the latents up: the bound scales with their magnitude (as in
    # test_denoise_loop_over_the_module)
    scale = max(1.0, float(want.abs().max()) / 4.0)
    err = (out.latents.float() - want).abs().max().item()
    assert err < TOL * scale, (err, scale)
    # the same through the tuple return and [negative | positive] embeddings in one tensor
    tup = pipe(T, prompt_embeds=prompt, latents=latents, condition_latents=cond, num_inference_steps=2, guidance_scale=3.5,
               context_frames=win, context_overlap=ov, output_type="latent", return_dict=False)
    assert torch.equal(tup[1], out.latents)
    with pytest.raises(ValueError):
        pipe(T, prompt="a cat", latents=latents, num_inference_steps=2)          # strings need a text encoder callable
    # img2img / video2video start (pipeline_controlnet.py:1627-1633, 283-430): `image` + `latents` + `strength` -> the schedule starts
    # at step N - int(N * strength) from the caller's latents; `image` alone -> its VAE latents noised to the first timestep
    N, strength = 4, 0.5
    want2 = opipe.denoise_loop(lambda x, t, e, **k: unet3d.unet3d_forward(sd, cfg, x, t, e, **k), latents, prompt, num_inference_steps=N,
                               guidance_scale=3.5, condition_latents=cond, motion_speed=8.0, context_frames=win, context_overlap=ov,
                               start_step=N - int(N * strength))
    frames = torch.zeros(1, 3, T, 64, 64)
    with pytest.raises(ValueError):
        pipe(T, prompt_embeds=prompt, image=frames, height=64, width=64, num_inference_steps=N)   # `image` alone needs a VAE encoder
    got2 = pipe(T, prompt_embeds=prompt, latents=latents, image=frames, strength=strength, condition_latents=cond, num_inference_steps=N,
                guidance_scale=3.5, context_frames=win, context_overlap=ov, output_type="latent").latents
    scale2 = max(1.0, float(want2.abs().max()) / 4.0)
    assert (got2.float() - want2).abs().max().item() < TOL * scale2
    init = 0.18215 * torch.randn(1, 4, T, h, w, generator=torch.Generator().manual_seed(5))
    pipe2 = MusevControlNetPipeline(unet=unet, vae_encode=lambda img: init)
    gen = torch.Generator().manual_seed(9)
    got3 = pipe2(T, prompt_embeds=prompt, image=frames, height=8 * h, width=8 * w, condition_latents=cond, num_inference_steps=N,
                 guidance_scale=3.5, context_frames=win, context_overlap=ov, output_type="latent", generator=gen).latents
    from musev_amd.schedulers import DDIMScheduler
    from musev_amd.utils.noise_util import prepare_noise_latents
    sch = DDIMScheduler()
    sch.set_timesteps(N)
    noise = prepare_noise_latents((1, 4, T, h, w), dtype=torch.float32, device=torch.device("cpu"), generator=torch.Generator().manual_seed(9))
    want3 = opipe.denoise_loop(lambda x, t, e, **k: unet3d.unet3d_forward(sd, cfg, x, t, e, **k), sch.add_noise(init, noise, 0), prompt,
                               num_inference_steps=N, guidance_scale=3.5, condition_latents=cond, motion_speed=8.0, context_frames=win,
                               context_overlap=ov)
    assert (got3.float() - want3).abs().max().item() < TOL * max(1.0, float(want3.abs().max()) / 4.0)
