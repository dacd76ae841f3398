#!/usr/bin/env python3
"""Generates tests/golden/*.npz / *.json by EXECUTING THE REFERENCE'S OWN SOURCE (/root/reference/musev) in this
container, with the un-vendored third-party packages replaced by tests/golden/refshim.py.

Run:  python tests/golden/make_reference_goldens.py          (needs /root/reference; not available on the GPU box)

What is pinned by these fixtures (consumed by tests/test_oracle_golden.py and tests/test_model_gpu.py):
  * reference_context.json      musev/pipelines/context.py  prepare_global_context / uniform / ordered_halving
  * reference_unet_<case>.npz   musev/models/unet_3d_condition.py UNet3DConditionModel.forward (+ unet_3d_blocks,
                                resnet.TemporalConvLayer, temporal_transformer, transformer_2d, attention,
                                attention_processor: reference-only self-attn, IP-Adapter cross-attn,
                                ReferEmbFuseAttention, data_util helpers) on seeded weights/inputs.  The weights are
                                the oracle's seeded state dict, loaded with strict=True -> the key/shape inventory of
                                oracle.unet3d.param_shapes is checked against the reference constructor as well.
  * reference_ddim.npz          musev/schedulers/scheduling_ddim.py DDIMScheduler.step (eta = 0, epsilon)
  * reference_euler.npz         musev/schedulers/scheduling_euler_discrete.py EulerDiscreteScheduler.step (s_churn = 0)
  * reference_datautil.npz      musev/data/data_util.py index helpers used by the loop
  * reference_referencenet_*.npz musev/models/referencenet.py ReferenceNet2D.forward (block-embedding mode): 12 + 1 feature maps
  * reference_loop_utils.json / .npz  musev/utils/timesteps_util.py generate_parameters_with_timesteps (guidance schedule) and
                                musev/utils/noise_util.py random_noise / video_fusion_noise (initial latents)
  * reference_poseguider_*.npz  musev/models/controlnet.py PoseGuider.forward (the pose conditioning of musev_referencenet_pose)
  * reference_unet_musev_cfg2.npz / reference_unet_refnet_cfg3.npz  (``--at-size``) the same forward at the sizes of BASELINE.json
                                configs 2 and 3: full SD-1.5 widths, B 2, T 13, 64x64 latents
  * reference_unet_refnet_pose_cfg5.npz  (``--at-size-cfg5``) the `musev_referencenet_pose` forward at config 5's size (96x96 latents) with
                                ControlNet residuals + PoseGuider embedding
  * reference_multi_shot.npz    musev/pipelines/pipeline_controlnet_predictor.py run_pipe_text2video's shot loop (its own source, pipeline stubbed)
  * reference_pipeline_signature.json  keyword list of MusevControlNetPipeline.__call__ (pipeline_controlnet.py:1295-1420), read with ast
Only seeds, configs and OUTPUTS are stored (inputs and weights are regenerated from the seeds by the tests).
"""
from __future__ import annotations

import json
import logging
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

import refshim  # noqa: E402

refshim.install("/root/reference")
logging.disable(logging.CRITICAL)

from oracle import unet3d  # noqa: E402

# ---- the golden UNet cases (shared with the tests through golden_cases.py) ----
from golden_cases import UNET_CASES, UNET_CASES_AT_SIZE, UNET_CASES_AT_SIZE_CFG5, case_config, case_inputs, FLAVOUR_CTOR_KWARGS  # noqa: E402


def gen_context():
    from musev.pipelines import context as rc
    table = []
    for sched in ("uniform", "uniform_v2"):
        for (T, size, ov, stride, bs) in [(96, 12, 4, 1, 1), (48, 12, 4, 1, 1), (24, 12, 4, 1, 1), (12, 12, 4, 1, 1), (7, 12, 4, 1, 1),
                                          (13, 12, 4, 1, 1), (64, 12, 4, 1, 1), (100, 16, 4, 3, 2), (37, 8, 2, 2, 1), (20, 12, 4, 1, 3)]:
            gc = rc.prepare_global_context(sched, 20, T, size, stride, ov, bs)
            table.append(dict(schedule=sched, time_size=T, context_frames=size, context_overlap=ov, context_stride=stride,
                              context_batch_size=bs, global_context=gc))
    halving = {str(v): rc.ordered_halving(v) for v in (0, 1, 2, 3, 5, 8, 13, 19, 1000)}
    stepped = {str(s): [list(map(int, wdw)) for wdw in rc.uniform(s, 20, 48, 12, 3, 4)] for s in (0, 1, 2, 7)}
    with open(os.path.join(HERE, "reference_context.json"), "w") as f:
        json.dump(dict(table=table, ordered_halving=halving, uniform_steps=stepped), f)
    print("context:", len(table), "entries")


def gen_unet(cases=None):
    import time
    from musev.models.unet_3d_condition import UNet3DConditionModel
    for name, case in (UNET_CASES if cases is None else cases).items():
        t_start = time.time()
        cfg = case_config(case)
        sd = unet3d.init_state_dict(cfg, case["weight_seed"])
        ctor = dict(FLAVOUR_CTOR_KWARGS[case["flavour"]])
        ctor.update(block_out_channels=tuple(cfg["block_out_channels"]), layers_per_block=cfg["layers_per_block"],
                    down_block_types=tuple(cfg["down_block_types"]), up_block_types=tuple(cfg["up_block_types"]),
                    cross_attention_dim=cfg["cross_attention_dim"], attention_head_dim=cfg["attention_head_dim"])
        if cfg.get("need_t2i_ip_adapter_face"):
            ctor["need_t2i_ip_adapter_face"] = True
        model = UNet3DConditionModel(**ctor).eval()
        missing, unexpected = model.load_state_dict(sd, strict=True)
        x, t, ehs, kw = case_inputs(case, cfg)
        if case.get("refer_self"):
            model.insert_spatial_self_attn_idx()
        with torch.no_grad():
            out = model(x, t, encoder_hidden_states=ehs, return_dict=False, **kw)[0]
            extra = {}
            if case.get("check_cfg_flag"):
                out_cfg = model(x, t, encoder_hidden_states=ehs, return_dict=False, do_classifier_free_guidance=True, **kw)[0]
                extra["cfg_flag_max_abs_diff"] = np.float32((out - out_cfg).abs().max().item())
        if case.get("refer_self_write"):   # the list the reference's forward filled (fp16 storage: these are LayerNorm outputs, O(1))
            for i, e in enumerate(kw["refer_self_attn_emb"]):
                if e is not None:
                    extra[f"emb{i}"] = e.numpy().astype(np.float16)
        np.savez_compressed(os.path.join(HERE, f"reference_unet_{name}.npz"), out=out.numpy().astype(np.float32), **extra)
        print("unet", name, tuple(out.shape), "absmax", out.abs().max().item(), {k: float(v) for k, v in extra.items() if np.ndim(v) == 0},
              f"{time.time() - t_start:.0f} s", flush=True)
        del model, sd


def gen_ddim():
    # musev/schedulers/__init__.py imports every sampler (DPM-Solver, Euler, LCM ...: out of scope); load the DDIM module
    # file alone by registering a bare package object for musev.schedulers first
    import types
    import musev
    pkg = types.ModuleType("musev.schedulers")
    pkg.__path__ = [os.path.join(os.path.dirname(musev.__file__), "schedulers")]
    sys.modules["musev.schedulers"] = pkg
    from musev.schedulers.scheduling_ddim import DDIMScheduler
    s = DDIMScheduler(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
                      clip_sample=False, set_alpha_to_one=False, steps_offset=1)
    s.set_timesteps(20)
    g = torch.Generator().manual_seed(77)
    x = torch.randn(1, 4, 6, 8, 8, generator=g)
    eps = torch.randn(1, 4, 6, 8, 8, generator=g)
    outs = {}
    for t in (951, 501, 51, 1):
        outs[f"t{t}"] = s.step(eps, t, x).prev_sample.numpy()
    np.savez_compressed(os.path.join(HERE, "reference_ddim.npz"), timesteps=s.timesteps.numpy(),
                        alphas_cumprod=s.alphas_cumprod.numpy(), **outs)
    print("ddim:", s.timesteps.tolist()[:3], "...")
    # the predictor's zero-SNR scheduler (pipeline_controlnet_predictor.py:270-282): v-prediction through the reference's own step
    # (scheduling_ddim.py:224-231); the beta rescale and the "trailing" spacing come from the stand-in base class (upstream, unpinned)
    s = DDIMScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule="linear", clip_sample=False, steps_offset=1,
                      prediction_type="v_prediction", rescale_betas_zero_snr=True, timestep_spacing="trailing")
    s.set_timesteps(20)
    outs = {f"t{int(t)}": s.step(eps, int(t), x).prev_sample.numpy() for t in s.timesteps[[0, 7, 18, 19]]}
    np.savez_compressed(os.path.join(HERE, "reference_ddim_vpred.npz"), timesteps=s.timesteps.numpy(),
                        alphas_cumprod=s.alphas_cumprod.numpy(), **outs)
    print("ddim v-prediction / zero SNR:", s.timesteps.tolist()[:3], "...", float(s.alphas_cumprod[-1]))


def gen_euler():
    """musev/schedulers/scheduling_euler_discrete.py (the reference's step override; base class = refshim restatement of
    upstream diffusers): 20-step schedules in the three spacings, four consecutive steps each, both noise types (the drawn
    noise must not enter the sample at s_churn = 0)."""
    import types
    import musev
    pkg = types.ModuleType("musev.schedulers")
    pkg.__path__ = [os.path.join(os.path.dirname(musev.__file__), "schedulers")]
    sys.modules["musev.schedulers"] = pkg
    from musev.schedulers.scheduling_euler_discrete import EulerDiscreteScheduler
    outs = {}
    for spacing, offset in (("linspace", 0), ("leading", 1), ("trailing", 0)):
        s = EulerDiscreteScheduler(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
                                   timestep_spacing=spacing, steps_offset=offset)
        s.set_timesteps(20)
        g = torch.Generator().manual_seed(79)
        x = torch.randn(1, 4, 6, 8, 8, generator=g) * float(s.init_noise_sigma)
        outs[f"{spacing}_timesteps"] = s.timesteps.numpy()
        outs[f"{spacing}_sigmas"] = s.sigmas.numpy()
        outs[f"{spacing}_init_noise_sigma"] = np.float32(float(s.init_noise_sigma))
        outs[f"{spacing}_x0"] = x.numpy()
        for i in range(4):
            t = s.timesteps[i]
            xin = s.scale_model_input(x, t)
            outs[f"{spacing}_scaled{i}"] = xin.numpy()
            eps = torch.randn(1, 4, 6, 8, 8, generator=g)
            outs[f"{spacing}_eps{i}"] = eps.numpy()
            x = s.step(eps, t, x, generator=torch.Generator().manual_seed(5 + i),
                       noise_type="video_fusion" if i % 2 else "random").prev_sample
            outs[f"{spacing}_x{i + 1}"] = x.numpy()
    np.savez_compressed(os.path.join(HERE, "reference_euler.npz"), **outs)
    print("euler: spacings linspace/leading/trailing, sigma_max", float(outs["linspace_sigmas"].max()))


def gen_datautil():
    from musev.data import data_util as du
    g = torch.Generator().manual_seed(88)
    d1 = torch.randn(2, 4, 1, 3, 3, generator=g)
    d2 = torch.randn(2, 4, 5, 3, 3, generator=g)
    cat = du.batch_concat_two_tensor_with_index(d1, torch.tensor([0]), d2, torch.arange(1, 6), dim=2)
    sel = du.batch_index_select(cat, dim=2, index=torch.arange(1, 6))
    rep = du.align_repeat_tensor_single_dim(torch.arange(6.0).reshape(2, 3), 8, dim=0)
    adain_in = torch.randn(10, 4, 3, 3, generator=g)
    adain_out = du.batch_adain_conditioned_tensor(adain_in, num_frames=5, need_style_fidelity=False,
                                                  src_index=torch.arange(1, 5), dst_index=torch.tensor([0]))
    np.savez_compressed(os.path.join(HERE, "reference_datautil.npz"), cat=cat.numpy(), sel=sel.numpy(), rep=rep.numpy(),
                        adain_is_identity=np.array(bool(torch.equal(adain_in, adain_out))))
    print("datautil: adain identity =", bool(torch.equal(adain_in, adain_out)))


def gen_referencenet():
    """musev/models/referencenet.py ReferenceNet2D.forward (need_block_embs=True, need_self_attn_block_embs=False) with the
    oracle's seeded weights loaded strict=True: pins the key / shape inventory and the 12 + 1 feature maps."""
    from musev.models.referencenet import ReferenceNet2D
    from golden_cases import REFNET_CASES, refnet_case_inputs
    from oracle import referencenet as oref
    for name, case in REFNET_CASES.items():
        cfg = oref.referencenet_config(**case["arch"])
        sd = oref.init_state_dict(cfg, case["weight_seed"])
        kw = dict(block_out_channels=tuple(cfg["block_out_channels"]), layers_per_block=cfg["layers_per_block"],
                  down_block_types=tuple(cfg["down_block_types"]), cross_attention_dim=cfg["cross_attention_dim"],
                  attention_head_dim=cfg["attention_head_dim"], need_block_embs=True, need_self_attn_block_embs=False)
        if "up_block_types" not in case["arch"] and len(cfg["block_out_channels"]) != 4:
            n_up = len(cfg["block_out_channels"])
            kw["up_block_types"] = ("UpBlock2D",) + ("CrossAttnUpBlock2D",) * (n_up - 1)
        net = ReferenceNet2D(**kw).eval()
        net.load_state_dict(sd, strict=True)
        x, t, ehs = refnet_case_inputs(case, cfg)
        with torch.no_grad():
            down, mid, sa = net(x, t, encoder_hidden_states=ehs, num_frames=case["t"], return_ndim=5)
        assert sa is None
        out = {f"down{i}": d.numpy().astype(np.float32) for i, d in enumerate(down)}
        out["mid"] = mid.numpy().astype(np.float32)
        np.savez_compressed(os.path.join(HERE, f"reference_referencenet_{name}.npz"), **out)
        print("referencenet", name, len(down), [tuple(d.shape) for d in down[:2]], tuple(mid.shape), "absmax", float(mid.abs().max()))


# the argument table is shared with tests/test_oracle_golden.py through golden_cases.py
def gen_loop_utils():
    from golden_cases import GUIDANCE_CASES, NOISE_CASES
    from musev.utils import noise_util as rn
    from musev.utils import timesteps_util as rt
    table = []
    for kw in GUIDANCE_CASES:
        try:
            out = [float(v) for v in rt.generate_parameters_with_timesteps(**kw)]
        except ValueError as ex:
            out = {"raises": "ValueError"}
        table.append({"args": kw, "out": out})
    with open(os.path.join(HERE, "reference_loop_utils.json"), "w") as fjs:
        json.dump(table, fjs, indent=0)
    arrays = {}
    for name, c in NOISE_CASES.items():
        shape = tuple(c["shape"])
        gen = [torch.Generator().manual_seed(sd) for sd in c["seeds"]] if c.get("per_item") else torch.Generator().manual_seed(c["seeds"][0])
        if c["kind"] == "random":
            out = rn.random_noise(shape=shape, dtype=torch.float32, device="cpu", generator=gen)
        else:
            common = None
            if c.get("common_seed") is not None:
                common = torch.randn(shape[0], shape[1], 1, shape[3], shape[4], generator=torch.Generator().manual_seed(c["common_seed"]))
            out = rn.video_fusion_noise(shape=shape, dtype=torch.float32, device="cpu", generator=gen, w_ind_noise=c["w"],
                                        initial_common_noise=common)
        arrays[name] = out.numpy()
    # generator consumption of the reference's Euler step (it draws a noise tensor per step even at s_churn = 0)
    import types
    import musev
    pkg = types.ModuleType("musev.schedulers")
    pkg.__path__ = [os.path.join(os.path.dirname(musev.__file__), "schedulers")]
    sys.modules.setdefault("musev.schedulers", pkg)
    from musev.schedulers.scheduling_euler_discrete import EulerDiscreteScheduler
    for noise_type in ("random", "video_fusion"):
        sch = EulerDiscreteScheduler(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear")
        sch.set_timesteps(20)
        g = torch.Generator().manual_seed(123)
        x = torch.randn(1, 4, 6, 8, 8, generator=torch.Generator().manual_seed(7)) * float(sch.init_noise_sigma)
        for i in range(3):
            t = sch.timesteps[i]
            eps = torch.randn(1, 4, 6, 8, 8, generator=torch.Generator().manual_seed(70 + i))
            x = sch.step(eps, t, sch.scale_model_input(x, t), generator=g, noise_type=noise_type, w_ind_noise=0.5).prev_sample
        arrays[f"euler_rng_after_{noise_type}"] = torch.randn(8, generator=g).numpy()
    np.savez_compressed(os.path.join(HERE, "reference_loop_utils.npz"), **arrays)
    print("loop utils:", len(table), "guidance cases,", len(arrays), "noise cases")


def gen_poseguider():
    """musev/models/controlnet.py PoseGuider.forward with the oracle's seeded weights loaded strict=True (pins the key /
    shape inventory) -- and the class's own initialisation: conv_out is zero, so a fresh PoseGuider outputs zeros."""
    from golden_cases import POSEGUIDER_CASES, poseguider_case_inputs
    from musev.models.controlnet import PoseGuider
    from oracle import poseguider as opg
    for name, c in POSEGUIDER_CASES.items():
        net = PoseGuider(conditioning_embedding_channels=c["emb"], conditioning_channels=c["cond"], block_out_channels=c["ch"]).eval()
        x = poseguider_case_inputs(c)
        with torch.no_grad():
            fresh = net(x)
        sd = opg.init_state_dict(opg.param_shapes(c["emb"], c["cond"], c["ch"]), c["weight_seed"])
        net.load_state_dict(sd, strict=True)
        with torch.no_grad():
            out = net(x)
        np.savez_compressed(os.path.join(HERE, f"reference_poseguider_{name}.npz"), out=out.numpy(),
                            fresh_is_zero=np.array(bool((fresh == 0).all())))
        print("poseguider", name, tuple(out.shape), "absmax", float(out.abs().max()), "fresh zero:", bool((fresh == 0).all()))


def gen_multi_shot():
    from golden_cases import MULTI_SHOT_CASES, multi_shot_stub_outputs
    """musev/pipelines/pipeline_controlnet_predictor.py:356-745 -- ``DiffusersPipelinePredictor.run_pipe_text2video``: the function's
    OWN SOURCE (cut out of the file with ast; the module itself imports cv2 / omegaconf / h5py / mmcm, none of which exist here) is
    executed against a stub pipeline that returns deterministic frames, with the vision-condition latents given (no first-frame
    generation); recorded: the condition latents every call received and the concatenated result."""
    import ast
    import textwrap
    import typing
    path = "/root/reference/musev/pipelines/pipeline_controlnet_predictor.py"
    src = open(path).read()
    tree = ast.parse(src)
    cls = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "DiffusersPipelinePredictor")
    fn = next(n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name == "run_pipe_text2video")
    code = textwrap.dedent(ast.get_source_segment(src, fn))

    class _Log:
        def debug(self, *a, **k):
            pass
    ns = dict(np=np, torch=torch, logger=_Log(), set_all_seed=lambda s: (None, None), hist_match_video_bcthw=None,
              batch_dynamic_crop_resize_images_v2=None, **{k: getattr(typing, k) for k in ("Union", "List", "Optional", "Callable", "Dict", "Any", "Tuple")})
    exec(compile(code, path, "exec"), ns)
    run = ns["run_pipe_text2video"]
    out = {}
    for name, (n_cond, T, shots, fixed) in MULTI_SHOT_CASES.items():
        calls = []

        class _Out:
            pass

        class _Pipe:
            referencenet = None
            ip_adapter_image_proj = None
            facein_image_proj = None

            def __call__(self, **kw):
                cond = kw["condition_latents"]
                calls.append(dict(cond=cond.clone(), video_length=kw["video_length"]))
                lat = multi_shot_stub_outputs(cond, kw["video_length"], len(calls) - 1)
                o = _Out()
                o.latents = lat
                o.videos = lat[:, :3].numpy()        # "np" output: the stub's video IS its first three latent channels
                return o

        class _Self:
            pipeline = _Pipe()
        g = torch.Generator().manual_seed(60 + n_cond)
        cond0 = torch.randn(1, 4, n_cond, 3, 4, generator=g)
        video = run(_Self(), video_length=T, prompt="p", height=24, width=32, condition_latents=cond0, n_vision_condition=n_cond,
                    max_batch_num=shots, fix_condition_images=fixed, video_num_inference_steps=4)
        out[f"{name}_cond0"] = cond0.numpy()
        out[f"{name}_video"] = np.asarray(video)
        for i, c in enumerate(calls):
            out[f"{name}_call{i}_cond"] = c["cond"].numpy()
            assert c["video_length"] == T
    np.savez_compressed(os.path.join(HERE, "reference_multi_shot.npz"), **out)
    print("multi shot:", {k: v.shape for k, v in out.items() if k.endswith("_video")})


CONDITION_INDEX_CASES = {
    # name: (n_cond, video_length, vision_condition_latent_index)
    "front_1": (1, 12, None), "front_2": (2, 6, None), "head_tail": (2, 12, [0, -1]), "tail_only": (1, 5, [-1]),
    "head_tail_short": (2, 6, [0, -1]), "explicit": (2, 5, [0, 3]),
}


def gen_condition_index():
    """musev/pipelines/pipeline_controlnet.py:966-1040 -- ``MusevControlNetPipeline.prepare_condition_latents_and_index``: the method's
    OWN SOURCE (cut out with ast: the module imports the un-vendored diffusers pipeline base) executed with a stub ``self``; recorded:
    the two index vectors it returns, and -- built with the reference's own data_util functions exactly as the loop calls them
    (:1914-1946 one window over all frames, :2068-2071, :2149-2156) -- the window input, the selected prediction rows and the final
    re-insert for a ramp of frame numbers."""
    import ast
    import json
    import textwrap
    import typing
    from einops import rearrange
    from musev.data import data_util as du
    path = "/root/reference/musev/pipelines/pipeline_controlnet.py"
    src = open(path).read()
    tree = ast.parse(src)
    cls = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "MusevControlNetPipeline")
    fn = next(n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name == "prepare_condition_latents_and_index")
    code = textwrap.dedent(ast.get_source_segment(src, fn))

    class _Log:
        def debug(self, *a, **k):
            pass

    class _Self:
        print_idx = 1
    ns = dict(np=np, torch=torch, logger=_Log(), rearrange=rearrange, **{k: getattr(typing, k) for k in ("Union", "List", "Optional", "Callable", "Dict", "Any", "Tuple")})
    exec(compile(code, path, "exec"), ns)
    prep = ns["prepare_condition_latents_and_index"]
    out = {}
    for name, (n_cond, T, vis) in CONDITION_INDEX_CASES.items():
        cond = 100.0 + torch.arange(n_cond, dtype=torch.float32).view(1, 1, n_cond, 1, 1).expand(1, 2, n_cond, 1, 3).contiguous()
        lat = 1.0 + torch.arange(T, dtype=torch.float32).view(1, 1, T, 1, 1).expand(1, 2, T, 1, 3).contiguous()
        c_lat, lat_idx, vis_idx = prep(_Self(), condition_images=None, condition_latents=cond.clone(), video_length=T, batch_size=1,
                                       dtype=torch.float32, device="cpu", latent_index=None, vision_condition_latent_index=vis)
        rec = dict(n_cond=n_cond, video_length=T, given=vis, vision_condition_latent_index=vis_idx.tolist(), latent_index=lat_idx.tolist())
        # the loop's window input for ONE window over all T frames (context = [range(T)]), CFG on (:1902-1946)
        x = torch.cat([lat] * 2)
        sub = torch.LongTensor(torch.arange(T) + n_cond)                      # sub_latent_index_c (:1914-1920)
        try:
            full = du.batch_concat_two_tensor_with_index(data1=torch.cat([c_lat] * 2), data1_index=vis_idx, data2=x, data2_index=sub, dim=2)
            rec["window_input_frames"] = full[0, 0, :, 0, 0].tolist()          # which frame sits in every slot (0 = never written)
            rec["selected"] = du.batch_index_select(full, dim=2, index=sub)[0, 0, :, 0, 0].tolist()      # :2068-2071
        except (IndexError, RuntimeError) as ex:
            rec["window_input_error"] = type(ex).__name__
        final = du.batch_concat_two_tensor_with_index(data1=c_lat, data1_index=vis_idx, data2=lat, data2_index=lat_idx, dim=2)  # :2149-2156
        rec["final_frames"] = final[0, 0, :, 0, 0].tolist()
        # ... and for a window of 4 frames out of T (several windows): global positions past n_cond + win do not exist in its input
        win = min(4, T)
        try:
            du.batch_concat_two_tensor_with_index(data1=torch.cat([c_lat] * 2), data1_index=vis_idx, data2=x[:, :, :win],
                                                  data2_index=torch.LongTensor(torch.arange(win) + n_cond), dim=2)
            rec["short_window"] = "ok"
        except (IndexError, RuntimeError) as ex:
            rec["short_window"] = type(ex).__name__
        out[name] = rec
    with open(os.path.join(HERE, "reference_condition_index.json"), "w") as f:
        f.write("{\n" + ",\n".join(f' "{k}": {json.dumps(v)}' for k, v in out.items()) + "\n}\n")
    print("condition index:", {k: (v["vision_condition_latent_index"], v.get("window_input_frames"), v["short_window"]) for k, v in out.items()})


def gen_pipeline_signature():
    """the keyword list (names, order, literal defaults) of MusevControlNetPipeline.__call__ and of the predictor's shot loop call
    site, read from the reference's SOURCE with ast (the module itself imports the un-vendored diffusers pipeline base)"""
    import ast
    src = open("/root/reference/musev/pipelines/pipeline_controlnet.py").read()
    tree = ast.parse(src)
    cls = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "MusevControlNetPipeline")
    fn = next(n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name == "__call__")
    args = fn.args.args[1:]  # without self
    defaults = [None] * (len(args) - len(fn.args.defaults)) + list(fn.args.defaults)
    out = []
    for a, d in zip(args, defaults):
        if d is None:
            out.append({"name": a.arg, "required": True})
        else:
            try:
                out.append({"name": a.arg, "default": ast.literal_eval(d)})
            except ValueError:
                out.append({"name": a.arg, "default_src": ast.unparse(d)})
    with open(os.path.join(HERE, "reference_pipeline_signature.json"), "w") as f:
        json.dump({"source": "musev/pipelines/pipeline_controlnet.py MusevControlNetPipeline.__call__", "lineno": fn.lineno, "args": out}, f, indent=0)
    print("pipeline signature:", len(out), "keywords, def at line", fn.lineno)


if __name__ == "__main__":
    if "--case" in sys.argv:  # one UNet case by name
        name = sys.argv[sys.argv.index("--case") + 1]
        gen_unet({name: dict(UNET_CASES, **UNET_CASES_AT_SIZE, **UNET_CASES_AT_SIZE_CFG5)[name]})
        sys.exit(0)
    if "--signature" in sys.argv:
        gen_pipeline_signature()
        sys.exit(0)
    if "--multi-shot" in sys.argv:
        gen_multi_shot()
        sys.exit(0)
    if "--ddim" in sys.argv:
        gen_ddim()
        sys.exit(0)
    if "--condition-index" in sys.argv:
        gen_condition_index()
        sys.exit(0)
    if "--at-size" in sys.argv:  # the BASELINE-size UNet cases only (minutes of CPU, ~25 GB)
        gen_unet(UNET_CASES_AT_SIZE)
        sys.exit(0)
    if "--at-size-cfg5" in sys.argv:  # the config-5-size case (96x96 latents; ~100 TFLOP of CPU work, ~45 GB)
        torch.set_num_threads(os.cpu_count() or 1)
        gen_unet(UNET_CASES_AT_SIZE_CFG5)
        sys.exit(0)
    gen_pipeline_signature()
    gen_condition_index()
    gen_multi_shot()
    gen_poseguider()
    gen_loop_utils()
    gen_context()
    gen_ddim()
    gen_euler()
    gen_datautil()
    gen_referencenet()
    gen_unet()
