"""CPU (-m "not gpu"): the oracle against golden vectors produced by EXECUTING THE REFERENCE'S OWN SOURCE
(tests/golden/make_reference_goldens.py, third-party deps replaced by tests/golden/refshim.py).  fp32 vs fp32 on the
same host: tolerance 2e-4 absolute on O(1) outputs (different but equivalent op orders / SDPA kernels)."""
import json
import os

import numpy as np
import pytest
import torch

from golden_cases import UNET_CASES, UNET_CASES_AT_SIZE, case_config, case_inputs, check_written_refer_embs

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


AT_SIZE = os.environ.get("MUSEV_GOLDEN_AT_SIZE", "0") == "1"


@pytest.mark.parametrize("name", list(UNET_CASES) + [
    pytest.param(n, marks=pytest.mark.skipif(not AT_SIZE, reason="BASELINE-size oracle replay: minutes of CPU and ~20 GB per "
                                                                   "case; opt in with MUSEV_GOLDEN_AT_SIZE=1 (last run: DESIGN.md 4)"))
    for n in UNET_CASES_AT_SIZE])
def test_oracle_unet_matches_reference(name):
    from oracle import unet3d
    case = dict(UNET_CASES, **UNET_CASES_AT_SIZE)[name]
    cfg = case_config(case)
    sd = unet3d.init_state_dict(cfg, case["weight_seed"])
    x, t, ehs, kw = case_inputs(case, cfg)
    with torch.no_grad():
        got = unet3d.unet3d_forward(sd, cfg, x, t, ehs, **kw)
    g = np.load(os.path.join(GOLD, f"reference_unet_{name}.npz"))
    want = torch.from_numpy(g["out"])
    assert got.shape == want.shape
    err = (got - want).abs().max().item()
    print(f"{name}: oracle vs reference |delta|max = {err:.3e}")
    assert err < 2e-4, f"{name}: oracle deviates from the reference by {err}"
    if case.get("refer_self_write"):   # the list the forward filled ("write" mode) against the reference's (stored as fp16)
        print(f"{name}: written refer_self_attn_emb |delta|max = {check_written_refer_embs(name, kw['refer_self_attn_emb'], g, 2e-3):.3e}")
    if "cfg_flag_max_abs_diff" in g:
        # the reference's do_classifier_free_guidance recompute (attention.py:319-334) must be dead code
        assert float(g["cfg_flag_max_abs_diff"]) == 0.0


def test_context_matches_reference():
    from oracle import pipeline as opipe
    from musev_amd.pipelines import context as pctx
    with open(os.path.join(GOLD, "reference_context.json")) as f:
        ref = json.load(f)
    for e in ref["table"]:
        args = (e["schedule"], 20, e["time_size"], e["context_frames"], e["context_stride"], e["context_overlap"], e["context_batch_size"])
        assert opipe.prepare_global_context(*args) == e["global_context"], e
        assert pctx.prepare_global_context(*args) == e["global_context"], e
    for k, v in ref["ordered_halving"].items():
        assert opipe.ordered_halving(int(k)) == v
        assert pctx.ordered_halving(int(k)) == v
    for s, wins in ref["uniform_steps"].items():
        assert [list(w) for w in opipe.uniform(int(s), 20, 48, 12, 3, 4)] == wins
        assert [list(w) for w in pctx.uniform(int(s), 20, 48, 12, 3, 4)] == wins


def test_context_known_answers():
    """T=96, window 12, overlap 4 -> 12 windows, the last wraps to frame 0; coverage 2,2,2,2,1,1,1,1"""
    from musev_amd.pipelines.context import prepare_global_context
    gc = prepare_global_context("uniform", 20, 96, 12, 1, 4, 1)
    wins = [c[0] for c in gc]
    assert len(wins) == 12
    assert wins[0] == list(range(12)) and wins[1] == list(range(8, 20))
    assert wins[-1] == [88, 89, 90, 91, 92, 93, 94, 95, 0, 1, 2, 3]
    cov = [0] * 96
    for wd in wins:
        for i in wd:
            cov[i] += 1
    assert cov == [2, 2, 2, 2, 1, 1, 1, 1] * 12
    assert len(prepare_global_context("uniform", 20, 48, 12, 1, 4, 1)) == 6
    assert len(prepare_global_context("uniform", 20, 24, 12, 1, 4, 1)) == 3
    assert prepare_global_context("uniform", 20, 12, 12, 1, 4, 1) == [[list(range(12))]]
    assert prepare_global_context("uniform", 20, 5, 12, 1, 4, 1) == [[list(range(5))]]   # ragged: shorter than one window


def test_ddim_matches_reference():
    from oracle import pipeline as opipe
    from musev_amd.schedulers import DDIMScheduler
    g = np.load(os.path.join(GOLD, "reference_ddim.npz"))
    o = opipe.DDIMOracle()
    o.set_timesteps(20)
    p = DDIMScheduler()
    p.set_timesteps(20)
    assert o.timesteps.tolist() == g["timesteps"].tolist() == p.timesteps.tolist()
    assert np.array_equal(o.alphas_cumprod.numpy(), g["alphas_cumprod"])
    assert np.array_equal(p.alphas_cumprod.numpy(), g["alphas_cumprod"])
    gen = torch.Generator().manual_seed(77)
    x = torch.randn(1, 4, 6, 8, 8, generator=gen)
    eps = torch.randn(1, 4, 6, 8, 8, generator=gen)
    for t in (951, 501, 51, 1):
        want = torch.from_numpy(g[f"t{t}"])
        got = o.step(eps, t, x)
        assert (got - want).abs().max().item() < 1e-5, t
        assert p.alphas_for(t) == o.alphas(t)


def test_ddim_v_prediction_zero_snr_matches_reference():
    """the predictor's `enable_zero_snr` scheduler (pipeline_controlnet_predictor.py:270-282): v-prediction, zero-terminal-SNR betas,
    "trailing" spacing -- oracle step and the product's affine coefficients against the reference's own step (scheduling_ddim.py:224-264)"""
    from oracle import pipeline as opipe
    from musev_amd.schedulers import DDIMScheduler
    g = np.load(os.path.join(GOLD, "reference_ddim_vpred.npz"))
    kw = dict(beta_start=0.00085, beta_end=0.012, beta_schedule="linear", steps_offset=1, prediction_type="v_prediction",
              rescale_betas_zero_snr=True, timestep_spacing="trailing", set_alpha_to_one=True)   # (diffusers' default: the predictor does not pass it)
    o = opipe.DDIMOracle(**kw)
    o.set_timesteps(20)
    p = DDIMScheduler(clip_sample=False, **kw)
    p.set_timesteps(20)
    assert o.timesteps.tolist() == g["timesteps"].tolist() == p.timesteps.tolist()
    assert np.allclose(o.alphas_cumprod.numpy(), g["alphas_cumprod"], rtol=0, atol=1e-7)
    assert np.allclose(p.alphas_cumprod.numpy(), g["alphas_cumprod"], rtol=0, atol=1e-7)
    assert float(p.alphas_cumprod[-1]) == 0.0   # zero terminal SNR
    gen = torch.Generator().manual_seed(77)
    x = torch.randn(1, 4, 6, 8, 8, generator=gen)
    v = torch.randn(1, 4, 6, 8, 8, generator=gen)
    for key in (k for k in g.files if k.startswith("t") and k[1:].isdigit()):
        t = int(key[1:])
        want = torch.from_numpy(g[key])
        assert (o.step(v, t, x) - want).abs().max().item() < 1e-5, t
        cx, ce = p.v_coefficients(*p.alphas_for(t))           # what loop_update hands mv_cfg_affine_step
        assert (cx * x + ce * v - want).abs().max().item() < 1e-5, t


def test_datautil_semantics():
    """the index helpers the loop relies on (data_util.py:242-292,413-437,605-652) and the AdaIN no-op (:550-602)"""
    from oracle import unet3d
    g = np.load(os.path.join(GOLD, "reference_datautil.npz"))
    assert bool(g["adain_is_identity"])
    gen = torch.Generator().manual_seed(88)
    d1 = torch.randn(2, 4, 1, 3, 3, generator=gen)
    d2 = torch.randn(2, 4, 5, 3, 3, generator=gen)
    cat = torch.zeros(2, 4, 6, 3, 3)
    cat.index_copy_(2, torch.tensor([0]), d1)
    cat.index_copy_(2, torch.arange(1, 6), d2)
    assert np.array_equal(cat.numpy(), g["cat"])
    assert np.array_equal(cat.index_select(2, torch.arange(1, 6)).numpy(), g["sel"])
    assert np.array_equal(unet3d.align_repeat(torch.arange(6.0).reshape(2, 3), 8, dim=0).numpy(), g["rep"])


def test_condition_index_matches_reference():
    """where the vision-condition frames sit: oracle.pipeline.condition_indices / concat_with_index against
    tests/golden/reference_condition_index.json -- ``prepare_condition_latents_and_index`` (pipeline_controlnet.py:966-1040, its own
    source executed) and the window input / selection / final re-insert built with the reference's data_util functions as the loop
    calls them (:1914-1946, :2068-2071, :2149-2156).  Incl. the reference's literal head + tail behaviour: with [0, -1] and one
    window the tail condition frame is overwritten by the last generated frame and slot 1 stays zero; a window shorter than the
    video raises IndexError."""
    import json
    from oracle import pipeline as opipe
    gold = json.load(open(os.path.join(GOLD, "reference_condition_index.json")))
    assert {"front_1", "head_tail", "tail_only"} <= set(gold)
    for name, g in gold.items():
        n_cond, T = g["n_cond"], g["video_length"]
        vis, lat_idx = opipe.condition_indices(n_cond, T, g["given"])
        assert vis.tolist() == g["vision_condition_latent_index"], name
        assert lat_idx.tolist() == g["latent_index"], name
        cond = 100.0 + torch.arange(n_cond, dtype=torch.float32).view(1, 1, n_cond, 1, 1).expand(1, 2, n_cond, 1, 3).contiguous()
        lat = 1.0 + torch.arange(T, dtype=torch.float32).view(1, 1, T, 1, 1).expand(1, 2, T, 1, 3).contiguous()
        sub = torch.arange(T) + n_cond
        if "window_input_error" in g:
            with pytest.raises((IndexError, RuntimeError)):
                opipe.concat_with_index(torch.cat([cond] * 2), vis, torch.cat([lat] * 2), sub)
        else:
            full = opipe.concat_with_index(torch.cat([cond] * 2), vis, torch.cat([lat] * 2), sub)
            assert full[0, 0, :, 0, 0].tolist() == g["window_input_frames"], name
            assert full.index_select(2, sub)[0, 0, :, 0, 0].tolist() == g["selected"], name
        assert opipe.concat_with_index(cond, vis, lat, lat_idx)[0, 0, :, 0, 0].tolist() == g["final_frames"], name
        win = min(4, T)
        if g["short_window"] == "ok":
            opipe.concat_with_index(torch.cat([cond] * 2), vis, torch.cat([lat] * 2)[:, :, :win], torch.arange(win) + n_cond)
        else:
            with pytest.raises((IndexError, RuntimeError)):
                opipe.concat_with_index(torch.cat([cond] * 2), vis, torch.cat([lat] * 2)[:, :, :win], torch.arange(win) + n_cond)


def test_euler_matches_reference():
    """oracle.pipeline.EulerOracle and the host tables of musev_amd.schedulers.EulerDiscreteScheduler against the reference's
    own EulerDiscreteScheduler (step override musev/schedulers/scheduling_euler_discrete.py:47-167 executed on the refshim
    restatement of the diffusers base): timesteps, sigmas, init_noise_sigma, scale_model_input and four steps per spacing."""
    import os

    import numpy as np
    from oracle import pipeline as opipe
    from musev_amd.schedulers import EulerDiscreteScheduler
    gold = np.load(os.path.join(os.path.dirname(__file__), "golden", "reference_euler.npz"))
    for spacing, offset in (("linspace", 0), ("leading", 1), ("trailing", 0)):
        o = opipe.EulerOracle(timestep_spacing=spacing, steps_offset=offset)
        o.set_timesteps(20)
        h = EulerDiscreteScheduler(timestep_spacing=spacing, steps_offset=offset)
        h.set_timesteps(20)
        for sched in (o, h):
            assert np.array_equal(sched.timesteps.numpy(), gold[f"{spacing}_timesteps"])
            assert np.array_equal(sched.sigmas.numpy(), gold[f"{spacing}_sigmas"])
            assert abs(float(sched.init_noise_sigma) - float(gold[f"{spacing}_init_noise_sigma"])) < 1e-6
        x = torch.from_numpy(gold[f"{spacing}_x0"])
        for i in range(4):
            want_in = torch.from_numpy(gold[f"{spacing}_scaled{i}"])
            assert torch.allclose(o.scale_model_input(x, i), want_in, rtol=1e-6, atol=1e-6)
            assert torch.allclose(x * h.input_scale(i), want_in, rtol=1e-6, atol=1e-6)
            assert torch.allclose(h.scale_model_input(x, h.timesteps[i]), want_in, rtol=1e-6, atol=1e-6)
            eps = torch.from_numpy(gold[f"{spacing}_eps{i}"])
            x = o.step(eps, i, x)
            want = torch.from_numpy(gold[f"{spacing}_x{i + 1}"])
            assert torch.allclose(x, want, rtol=1e-5, atol=1e-5), (spacing, i, (x - want).abs().max())
            # the affine form the fused kernel evaluates: x + (sigma_next - sigma) * eps
            prev = torch.from_numpy(gold[f"{spacing}_x{i}"] if i else gold[f"{spacing}_x0"])
            affine = prev + (float(h.sigmas[i + 1]) - float(h.sigmas[i])) * eps
            assert torch.allclose(affine, want, rtol=1e-5, atol=2e-5), (spacing, i, (affine - want).abs().max())


@pytest.mark.parametrize("name", ["narrow", "narrow_2ref", "hipw"])
def test_oracle_referencenet_matches_reference(name):
    """oracle.referencenet.referencenet_forward against the 12 + 1 feature maps recorded from the reference's own
    ReferenceNet2D (musev/models/referencenet.py:640-1143, block-embedding mode) on the same seeded weights and inputs."""
    import os

    import numpy as np
    from golden_cases import REFNET_CASES, refnet_case_inputs
    from oracle import referencenet as oref
    case = REFNET_CASES[name]
    cfg = oref.referencenet_config(**case["arch"])
    sd = oref.init_state_dict(cfg, case["weight_seed"])
    x, t, ehs = refnet_case_inputs(case, cfg)
    with torch.no_grad():
        down, mid = oref.referencenet_forward(sd, cfg, x, t, ehs, num_frames=case["t"])
    gold = np.load(os.path.join(os.path.dirname(__file__), "golden", f"reference_referencenet_{name}.npz"))
    assert len(down) == sum(1 for k in gold.files if k.startswith("down"))
    for i, d in enumerate(down):
        want = torch.from_numpy(gold[f"down{i}"])
        assert d.shape == want.shape
        assert torch.allclose(d, want, rtol=1e-4, atol=2e-4), (name, i, (d - want).abs().max())
    assert torch.allclose(mid, torch.from_numpy(gold["mid"]), rtol=1e-4, atol=2e-4)


# ---- loop utilities: guidance schedule + initial noise, oracle AND product host code vs the reference ----
def test_guidance_schedule_matches_reference():
    """musev/utils/timesteps_util.py executed by tests/golden/make_reference_goldens.py -> reference_loop_utils.json"""
    from oracle.pipeline import guidance_schedule
    from musev_amd.utils.timesteps_util import generate_parameters_with_timesteps
    table = json.load(open(os.path.join(GOLD, "reference_loop_utils.json")))
    assert len(table) >= 10
    for row in table:
        kw = row["args"]
        if isinstance(row["out"], dict):  # the reference raises ValueError for an unknown method
            with pytest.raises(ValueError):
                guidance_schedule(kw["start"], kw["num"], kw.get("stop"), kw.get("method", "linear"), kw.get("n_fix_start", 3))
            with pytest.raises(ValueError):
                generate_parameters_with_timesteps(**kw)
            continue
        got_o = guidance_schedule(kw["start"], kw["num"], kw.get("stop"), kw.get("method", "linear"), kw.get("n_fix_start", 3))
        got_p = generate_parameters_with_timesteps(**kw)
        assert [float(v) for v in got_o] == row["out"], kw
        assert [float(v) for v in got_p] == row["out"], kw


def test_initial_noise_matches_reference():
    """musev/utils/noise_util.py random_noise / video_fusion_noise on seeded generators: the same draws in the same order
    (bit-exact), for the oracle restatement and for the product's musev_amd.utils.noise_util"""
    from golden_cases import NOISE_CASES
    from oracle.pipeline import fusion_noise
    from musev_amd.utils import noise_util as pn
    gold = np.load(os.path.join(GOLD, "reference_loop_utils.npz"))
    for name, c in NOISE_CASES.items():
        shape = tuple(c["shape"])

        def gens():
            return [torch.Generator().manual_seed(sd) for sd in c["seeds"]] if c.get("per_item") else torch.Generator().manual_seed(c["seeds"][0])

        want = torch.from_numpy(gold[name])
        if c["kind"] == "random":
            got = pn.random_noise(shape=shape, dtype=torch.float32, device="cpu", generator=gens())
            assert torch.equal(got, want), name
            continue
        common = None
        if c.get("common_seed") is not None:
            common = torch.randn(shape[0], shape[1], 1, shape[3], shape[4], generator=torch.Generator().manual_seed(c["common_seed"]))
        got = pn.video_fusion_noise(shape=shape, dtype=torch.float32, device="cpu", generator=gens(), w_ind_noise=c["w"],
                                    initial_common_noise=common)
        assert torch.equal(got, want), name
        assert torch.equal(fusion_noise(shape, gens(), c["w"], common), want), name
        # the text2video branch of prepare_latents: the same noise times init_noise_sigma
        lat = pn.prepare_noise_latents(shape, dtype=torch.float32, device="cpu", generator=gens(), noise_type="video_fusion",
                                       w_ind_noise=c["w"], initial_common_latent=common, init_noise_sigma=14.6146)
        assert torch.equal(lat, want * 14.6146), name
    with pytest.raises(ValueError):
        pn.video_fusion_noise(shape=(2, 4, 3, 2, 2), dtype=torch.float32, device="cpu", generator=[torch.Generator()])
    # image-based video noise (pipeline_controlnet.py:325-344): sqrt(w) * mean_t(cond) + sqrt(1 - w) * noise
    noise = torch.randn(1, 4, 5, 3, 3, generator=torch.Generator().manual_seed(1))
    cond = torch.randn(1, 4, 2, 3, 3, generator=torch.Generator().manual_seed(2))
    mixed = pn.img_based_video_noise(noise, cond, img_weight=1e-3)
    ref = 1e-3 ** 0.5 * cond.mean(dim=2, keepdim=True).repeat(1, 1, 5, 1, 1) + (1 - 1e-3) ** 0.5 * noise
    assert torch.equal(mixed, ref)


def test_euler_step_noise_consumption_matches_reference():
    """the reference's Euler step draws one noise tensor of the model-output shape per step from the caller's generator even
    though s_churn = 0 discards it (scheduling_euler_discrete.py:120-131): after 3 steps the generator must be in the same
    state, for both noise types"""
    from musev_amd.schedulers import DDIMScheduler, EulerDiscreteScheduler
    gold = np.load(os.path.join(GOLD, "reference_loop_utils.npz"))
    for noise_type in ("random", "video_fusion"):
        s = EulerDiscreteScheduler()
        s.set_timesteps(20)
        g = torch.Generator().manual_seed(123)
        for _ in range(3):
            s.consume_step_noise((1, 4, 6, 8, 8), torch.float32, "cpu", g, noise_type, 0.5)
        assert torch.equal(torch.randn(8, generator=g), torch.from_numpy(gold[f"euler_rng_after_{noise_type}"])), noise_type
    g = torch.Generator().manual_seed(123)
    state = g.get_state().clone()
    d = DDIMScheduler()
    d.consume_step_noise((1, 4, 6, 8, 8), torch.float32, "cpu", g)   # eta = 0: the reference's DDIM step draws nothing
    EulerDiscreteScheduler().consume_step_noise((1, 4, 6, 8, 8), torch.float32, "cpu", None)
    assert torch.equal(g.get_state(), state)


# ---- PoseGuider ---------------------------------------------------------------------------------------
from golden_cases import POSEGUIDER_CASES, poseguider_case_inputs  # noqa: E402


@pytest.mark.parametrize("name", list(POSEGUIDER_CASES))
def test_oracle_poseguider_matches_reference(name):
    """oracle/poseguider.py vs musev/models/controlnet.py PoseGuider.forward executed on the same seeded weights (loaded with
    strict=True into the reference class: the key / shape inventory is pinned too)"""
    from oracle import poseguider as opg
    c = POSEGUIDER_CASES[name]
    gold = np.load(os.path.join(GOLD, f"reference_poseguider_{name}.npz"))
    assert bool(gold["fresh_is_zero"])  # zero_module(conv_out): an untrained PoseGuider adds nothing
    sd = opg.init_state_dict(opg.param_shapes(c["emb"], c["cond"], c["ch"]), c["weight_seed"])
    got = opg.poseguider_forward(sd, poseguider_case_inputs(c))
    want = torch.from_numpy(gold["out"])
    assert got.shape == want.shape
    assert (got - want).abs().max().item() < 2e-5


def test_euler_from_config_and_step_match_reference_steps():
    """drop-in surface the predictor uses (pipeline_controlnet_predictor.py:258-261): ``EulerDiscreteScheduler.from_config(other.config)``
    (unknown keys ignored) and a plain-tensor ``step`` -- checked against the steps recorded from the reference's scheduler, and the
    per-step noise is drawn with the MODEL OUTPUT's dtype (scheduling_euler_discrete.py:120-131)"""
    from types import SimpleNamespace
    from musev_amd.schedulers import DDIMScheduler, EulerDiscreteScheduler
    gold = np.load(os.path.join(GOLD, "reference_euler.npz"))
    ddim = DDIMScheduler()
    cfg = dict(vars(ddim.config), timestep_spacing="leading", steps_offset=1, clip_sample=False, set_alpha_to_one=False)
    for config in (cfg, SimpleNamespace(**cfg)):
        h = EulerDiscreteScheduler.from_config(config)
        assert h.config.timestep_spacing == "leading" and h.config.steps_offset == 1 and not hasattr(h.config, "clip_sample")
        h.set_timesteps(20)
        assert np.array_equal(h.sigmas.numpy(), gold["leading_sigmas"])
    keys = [k for k in gold.files if k.startswith("leading_")]
    x = torch.from_numpy(gold["leading_x0"])
    eps_keys = sorted(k for k in keys if k.startswith("leading_eps"))
    out_keys = sorted(k for k in keys if k.startswith("leading_x") and k != "leading_x0")
    if eps_keys and len(eps_keys) == len(out_keys):
        for i, (ek, ok) in enumerate(zip(eps_keys, out_keys)):
            prev, _ = h.step(torch.from_numpy(gold[ek]), h.timesteps[i], x)
            assert (prev - torch.from_numpy(gold[ok])).abs().max().item() < 1e-5, i
            x = prev
    # generator consumption follows the model output's dtype: an fp16 model output consumes what the reference's fp16 draw consumes
    s = EulerDiscreteScheduler()
    s.set_timesteps(20)
    g1, g2 = torch.Generator().manual_seed(5), torch.Generator().manual_seed(5)
    mo = torch.zeros(1, 4, 6, 8, 8, dtype=torch.float16)
    s.step(mo, s.timesteps[0], torch.zeros_like(mo, dtype=torch.float32), generator=g1)
    torch.randn(mo.shape, generator=g2, dtype=torch.float16)
    assert torch.equal(g1.get_state(), g2.get_state())


# ---- the predictor's multi-shot loop, pinned against the reference function's own source -------------------------
@pytest.mark.parametrize("name", ["n1", "n2", "n1_fixed", "one_shot"])
def test_multi_shot_loop_matches_the_predictor_source(name, monkeypatch):
    """tests/golden/reference_multi_shot.npz was recorded by executing DiffusersPipelinePredictor.run_pipe_text2video (cut out of
    pipeline_controlnet_predictor.py with ast) against a stub pipeline; the oracle's multi_shot_loop and the product's
    multi_shot_denoise, driven by the same stub in place of the denoise loop, must hand every shot the same condition latents and
    return the same concatenation (the stub's video is its first three latent channels)."""
    from golden_cases import MULTI_SHOT_CASES, multi_shot_stub_outputs
    from oracle import pipeline as opipe
    from musev_amd.pipelines.video import multi_shot_denoise
    gold = np.load(os.path.join(GOLD, "reference_multi_shot.npz"))
    n_cond, T, shots, fixed = MULTI_SHOT_CASES[name]
    cond0 = torch.from_numpy(gold[f"{name}_cond0"])
    want = torch.from_numpy(gold[f"{name}_video"])
    noises = [torch.zeros(1, 4, T, 3, 4) for _ in range(shots)]

    # oracle: denoise_loop replaced by the stub
    calls = []

    def stub_loop(unet_fn, noise, prompt_embeds, *, condition_latents=None, **kw):
        calls.append(condition_latents.clone())
        return multi_shot_stub_outputs(condition_latents, noise.shape[2], len(calls) - 1)
    monkeypatch.setattr(opipe, "denoise_loop", stub_loop)
    got = opipe.multi_shot_loop(None, noises, None, cond0, n_vision_condition=n_cond, fix_condition_images=fixed)
    assert torch.equal(got[:, :3], want)
    for i, c in enumerate(calls):
        assert torch.equal(c, torch.from_numpy(gold[f"{name}_call{i}_cond"])), i

    # product: the denoiser replaced by the stub
    calls2 = []

    def stub_denoiser(noise, prompt_embeds, *, condition_latents=None, **kw):
        calls2.append(condition_latents.clone())
        return multi_shot_stub_outputs(condition_latents, noise.shape[2], len(calls2) - 1)
    lat, vid = multi_shot_denoise(stub_denoiser, lambda i: noises[i], None, condition_latents=cond0, n_vision_condition=n_cond,
                                  max_batch_num=shots, fix_condition_images=fixed)
    assert vid is None and torch.equal(lat[:, :3], want)
    for i, c in enumerate(calls2):
        assert torch.equal(c, torch.from_numpy(gold[f"{name}_call{i}_cond"])), i
