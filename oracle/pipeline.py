"""ORACLE -- test infrastructure only (see oracle/unet3d.py for the rules).

Plain-PyTorch/numpy CPU restatement of the sliding-window "visual conditioned parallel denoise" loop:
  * window schedulers      musev/pipelines/context.py:12-149
  * DDIM scheduler         musev/schedulers/scheduling_ddim.py:136-302 (+ diffusers DDIMScheduler base: betas,
                           alphas_cumprod, set_timesteps("leading"), SURVEY.md 8c -- un-vendored, UNPINNED)
  * denoise loop           musev/pipelines/pipeline_controlnet.py:1832-2147 (the parts in scope: window gather, CFG
                           duplication, vision-condition prepend, UNet call, scatter-add / counter average, CFG
                           combine, scheduler step, final cond re-insert :2149-2156)
Pinned: `uniform` / `drop_last_repeat_context` / `prepare_global_context` against the reference's own context.py
executed with the mmcm import stubbed, and DDIMScheduler.step against the reference's scheduling_ddim.py executed
with a stand-in diffusers base (tests/golden/make_reference_goldens.py).
"""
from __future__ import annotations

import math
from typing import Callable, List, Optional, Sequence

import numpy as np
import torch

Tensor = torch.Tensor


# ---- context.py ------------------------------------------------------------------------------------------
def ordered_halving(val: int) -> float:
    """context.py:12-17: bit-reversed fraction of a 64-bit integer."""
    return int(f"{val:064b}"[::-1], 2) / (1 << 64)


def uniform(step: int, num_steps: Optional[int], num_frames: int, context_size: int, context_stride: int = 3,
            context_overlap: int = 4, closed_loop: bool = True):
    """context.py:21-48."""
    if num_frames <= context_size:
        yield list(range(num_frames))
        return
    context_stride = min(context_stride, int(np.ceil(np.log2(num_frames / context_size))) + 1)
    for context_step in 1 << np.arange(context_stride):
        pad = int(round(num_frames * ordered_halving(step)))
        for j in range(int(ordered_halving(step) * context_step) + pad,
                       num_frames + pad + (0 if closed_loop else -context_overlap),
                       (context_size * context_step - context_overlap)):
            yield [e % num_frames for e in range(j, j + context_size * context_step, context_step)]


def generate_sample_idxs(total: int, window_size: int, step: int, sample_rate: int = 1, drop_last: bool = False):
    """mmcm.utils.itertools_util.generate_sample_idxs (un-vendored; semantics inferred from the call site
    context.py:60-66 and drop_last_repeat_context :105-117 -- UNPINNED)."""
    out = []
    s = 0
    while s < total:
        e = min(s + window_size * sample_rate, total)
        idx = list(range(s, e, sample_rate))
        if len(idx) < window_size and drop_last:
            break
        out.append(idx)
        s += step
    return out


def uniform_v2(step, num_steps, num_frames, context_size, context_stride=3, context_overlap=4, closed_loop=True):
    return generate_sample_idxs(num_frames, context_size, context_size - context_overlap, 1, False)


def drop_last_repeat_context(contexts: List[List[int]]) -> List[List[int]]:
    """context.py:105-117."""
    if len(contexts) >= 2 and contexts[-1][-1] == contexts[-2][-1]:
        return contexts[:-1]
    return contexts


def prepare_global_context(context_schedule: str, num_inference_steps: int, time_size: int, context_frames: int,
                           context_stride: int, context_overlap: int, context_batch_size: int) -> List[List[List[int]]]:
    """context.py:120-149 (always called with step = 0, pipeline_controlnet.py:1832-1840)."""
    sched: Callable = {"uniform": uniform, "uniform_v2": uniform_v2}[context_schedule]
    queue = list(sched(0, num_inference_steps, time_size, context_frames, context_stride, context_overlap))
    queue = drop_last_repeat_context(queue)
    n = math.ceil(len(queue) / context_batch_size)
    return [queue[i * context_batch_size:(i + 1) * context_batch_size] for i in range(n)]


# ---- DDIM ------------------------------------------------------------------------------------------------
class DDIMOracle:
    """SD-1.5 scheduler config (scaled_linear 0.00085 -> 0.012, 1000 steps, steps_offset 1, clip_sample False,
    set_alpha_to_one False, epsilon prediction, "leading" spacing), eta = 0."""

    def __init__(self, num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, steps_offset=1,
                 set_alpha_to_one=False, beta_schedule="scaled_linear", prediction_type="epsilon", timestep_spacing="leading",
                 rescale_betas_zero_snr=False):
        if beta_schedule == "linear":
            betas = torch.linspace(beta_start, beta_end, num_train_timesteps, dtype=torch.float32)
        else:
            betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
        if rescale_betas_zero_snr:  # diffusers rescale_zero_terminal_snr (un-vendored: restated from upstream, unpinned)
            abs_ = torch.cumprod(1.0 - betas, dim=0).sqrt()
            a0, aT = abs_[0].clone(), abs_[-1].clone()
            ab = ((abs_ - aT) * (a0 / (a0 - aT))) ** 2
            betas = 1.0 - torch.cat([ab[0:1], ab[1:] / ab[:-1]])
        self.prediction_type, self.timestep_spacing = prediction_type, timestep_spacing
        self.alphas_cumprod = torch.cumprod(1.0 - betas, dim=0)
        self.final_alpha_cumprod = torch.tensor(1.0) if set_alpha_to_one else self.alphas_cumprod[0]
        self.num_train_timesteps = num_train_timesteps
        self.steps_offset = steps_offset
        self.num_inference_steps = None
        self.timesteps = None

    def set_timesteps(self, n: int):
        self.num_inference_steps = n
        if self.timestep_spacing == "trailing":  # diffusers "trailing" spacing (un-vendored base class: restated, unpinned)
            ts = np.round(np.arange(self.num_train_timesteps, 0, -self.num_train_timesteps / n)).astype(np.int64) - 1
        else:
            ratio = self.num_train_timesteps // n
            ts = (np.arange(0, n) * ratio).round()[::-1].copy().astype(np.int64) + self.steps_offset
        self.timesteps = torch.from_numpy(ts)

    def alphas(self, t: int):
        prev_t = t - self.num_train_timesteps // self.num_inference_steps  # scheduling_ddim.py:198-200
        a_t = self.alphas_cumprod[t]
        a_prev = self.alphas_cumprod[prev_t] if prev_t >= 0 else self.final_alpha_cumprod  # :203-208
        return float(a_t), float(a_prev)

    def step(self, model_output: Tensor, t: int, sample: Tensor) -> Tensor:
        """scheduling_ddim.py:198-264 with eta = 0, epsilon prediction, no clipping."""
        a_t, a_prev = self.alphas(int(t))
        beta_t = 1 - a_t
        if self.prediction_type == "v_prediction":                  # :224-231
            x0 = a_t ** 0.5 * sample - beta_t ** 0.5 * model_output
            eps = a_t ** 0.5 * model_output + beta_t ** 0.5 * sample
        else:
            x0 = (sample - beta_t ** 0.5 * model_output) / a_t ** 0.5  # :214-218
            eps = model_output
        direction = (1 - a_prev) ** 0.5 * eps                       # :257-259 (std_dev_t = 0)
        return a_prev ** 0.5 * x0 + direction                       # :262-264


class EulerOracle:
    """musev/schedulers/scheduling_euler_discrete.py:47-167 (the step override) on diffusers' EulerDiscreteScheduler base
    (un-vendored; set_timesteps / sigmas / scale_model_input / init_noise_sigma restated from upstream v0.24 -- unpinned).
    s_churn = 0 as the reference pipeline calls it: gamma = 0, the drawn noise never enters the sample."""

    def __init__(self, num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, timestep_spacing="linspace", steps_offset=0):
        betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
        self.alphas_cumprod = torch.cumprod(1.0 - betas, dim=0)
        self.train_sigmas = (((1 - self.alphas_cumprod) / self.alphas_cumprod) ** 0.5).numpy()
        self.num_train_timesteps, self.timestep_spacing, self.steps_offset = num_train_timesteps, timestep_spacing, steps_offset
        self.sigmas = None
        self.timesteps = None

    def set_timesteps(self, n: int):
        N = self.num_train_timesteps
        if self.timestep_spacing == "linspace":
            ts = np.linspace(0, N - 1, n, dtype=np.float32)[::-1].copy()
        elif self.timestep_spacing == "leading":
            ts = (np.arange(0, n) * (N // n)).round()[::-1].copy().astype(np.float32) + self.steps_offset
        else:
            ts = (np.arange(N, 0, -N / n)).round().copy().astype(np.float32) - 1
        sig = np.interp(ts, np.arange(0, len(self.train_sigmas)), self.train_sigmas)
        self.sigmas = torch.from_numpy(np.concatenate([sig, [0.0]]).astype(np.float32))
        self.timesteps = torch.from_numpy(ts)

    @property
    def init_noise_sigma(self):
        m = float(self.sigmas.max())
        return m if self.timestep_spacing in ("linspace", "trailing") else (m ** 2 + 1) ** 0.5

    def scale_model_input(self, sample: Tensor, i: int) -> Tensor:
        sigma = self.sigmas[i]
        return sample / ((sigma ** 2 + 1) ** 0.5)

    def step(self, model_output: Tensor, i: int, sample: Tensor) -> Tensor:
        """scheduling_euler_discrete.py:110-167 with gamma = 0, epsilon prediction."""
        sigma = self.sigmas[i]
        sigma_hat = sigma * (0.0 + 1)                       # :133
        pred_original_sample = sample - sigma_hat * model_output   # :146-147
        derivative = (sample - pred_original_sample) / sigma_hat   # :158
        dt = self.sigmas[i + 1] - sigma_hat                        # :160
        return sample + derivative * dt                            # :162


# ---- loop utilities: guidance-scale schedule and initial noise ---------------------------------------------------------
def guidance_schedule(start, num: int, stop=None, method: str = "linear", n_fix_start: int = 3) -> List[float]:
    """musev/utils/timesteps_util.py:5-61 restated (the pipeline calls it with start = guidance_scale,
    stop = guidance_scale_end, num = len(timesteps): pipeline_controlnet.py:1718-1723)."""
    if stop is None or start == stop:                                   # :12-13
        return [start] * num
    if method == "linear":                                              # :30-38
        return [float(v) for v in np.linspace(start, stop, num)]
    if method == "two_stage":                                           # :41-45
        return [start] * (num // 2) + [stop] * (num - num // 2)
    if method == "three_stage":                                         # :55-61 (integer midpoint)
        k = num // 3
        return [start] * k + [(start + stop) // 2] * k + [stop] * (num - 2 * k)
    if method == "fix_two_stage":                                       # :48-52
        return [start] * n_fix_start + [stop] * (num - n_fix_start)
    raise ValueError(method)                                            # :23-26


def fusion_noise(shape, generator, w_ind_noise: float = 0.5, initial_common_noise: Optional[Tensor] = None) -> Tensor:
    """musev/utils/noise_util.py:31-83 (fp32, CPU): the common [b, c, 1, h, w] noise is drawn first, then the individual
    [b, c, t, h, w] noise; a list of generators draws item by item (:69-82)."""
    if isinstance(generator, list):
        return torch.cat([fusion_noise((1, *shape[1:]), g, w_ind_noise, initial_common_noise) for g in generator], dim=0)
    b, c, t, h, w = shape
    common = initial_common_noise if initial_common_noise is not None else torch.randn((b, c, 1, h, w), generator=generator)
    ind = torch.randn(tuple(shape), generator=generator)
    s = torch.tensor(w_ind_noise)
    return torch.sqrt(1 - s) * common + torch.sqrt(s) * ind               # :67-68


# ---- the loop --------------------------------------------------------------------------------------------
def condition_indices(n_cond: int, video_length: int, vision_condition_latent_index: Optional[Sequence[int]] = None):
    """prepare_condition_latents_and_index (pipeline_controlnet.py:966-1040), the index part: -1 -> n_cond + video_length - 1
    (:995-1003), default arange(n_cond) (:1010-1016); latent_index = the remaining positions in ascending order (:1017-1029).
    Pinned by tests/golden/reference_condition_index.json (the reference's own function executed)."""
    if not n_cond:
        return None, None
    total = n_cond + video_length
    if vision_condition_latent_index is None:
        vis = list(range(n_cond))
    else:
        vis = [int(i) if int(i) != -1 else total - 1 for i in vision_condition_latent_index]
    lat = sorted(set(range(total)) - set(vis))
    return torch.tensor(vis, dtype=torch.long), torch.tensor(lat, dtype=torch.long)


def concat_with_index(data1: Tensor, data1_index: Tensor, data2: Tensor, data2_index: Tensor) -> Tensor:
    """data_util.py:242-268 (batch_concat_two_tensor_with_index, dim = 2): a zero tensor of len1 + len2 frames, data1 copied to its
    positions, then data2 to its own (a later copy wins; a position out of range raises IndexError)."""
    full = torch.zeros((data1.shape[0], data1.shape[1], data1.shape[2] + data2.shape[2], *data1.shape[3:]), dtype=data1.dtype)
    full.index_copy_(2, data1_index, data1)
    full.index_copy_(2, data2_index, data2)
    return full


def denoise_loop(unet_fn: Callable[..., Tensor], latents: Tensor, prompt_embeds: Tensor, *, num_inference_steps: int,
                 guidance_scale: float, condition_latents: Optional[Tensor] = None, context_frames: int = 12,
                 context_overlap: int = 4, context_stride: int = 1, context_schedule: str = "uniform",
                 context_batch_size: int = 1, motion_speed: float = 8.0, unet_kwargs: Optional[dict] = None,
                 record: Optional[list] = None, max_steps: Optional[int] = None, scheduler: str = "ddim",
                 scheduler_kwargs: Optional[dict] = None, guidance_scale_end: Optional[float] = None,
                 guidance_scale_method: str = "linear", controlnet_fn: Optional[Callable[..., tuple]] = None,
                 control_image: Optional[Tensor] = None, controlnet_conditioning_scale: float = 1.0,
                 control_guidance_start: float = 0.0, control_guidance_end: float = 1.0, guess_mode: bool = False,
                 record_latents: Optional[list] = None, start_step: int = 0,
                 vision_condition_latent_index: Optional[Sequence[int]] = None) -> Tensor:
    """pipeline_controlnet.py:1832-2156.  ``vision_condition_latent_index`` (prepare_condition_latents_and_index, :966-1040): where
    the condition frames sit among the n_cond + T output frames, -1 = the last one (:995-1003); None = in front.  The loop scatters
    the condition latents into every WINDOW's input at these GLOBAL positions and then the window's frames at n_cond.. (:1914-1946,
    a later index_copy_ wins; a position outside the window's n_cond + win slots raises IndexError as torch does): with [0, -1] and
    one window the tail condition frame is overwritten by the last generated frame and slot 1 stays zero, while the UNet is still
    told that slots 0 and n_cond + T - 1 are the condition frames -- the reference's literal behaviour, restated as it is.  ``start_step`` (test helper: resume of an interrupted golden run; DDIM at eta 0 carries no
    state between steps): ``latents`` are the latents AFTER step ``start_step``, the first ``start_step`` schedule entries are skipped.  ``record`` / ``record_latents`` (test helpers): per-step guided noise prediction / latents.  ``max_steps`` (test helper, not in the reference): stop after the first
    max_steps entries of the num_inference_steps-long schedule.  latents [1, c, T, h, w] (generated frames only); condition_latents
    [1, c, n_cond, h, w] or None; prompt_embeds [2, 77, d] = [uncond, cond].  unet_fn(sample, t, ehs, sample_index=,
    vision_conditon_frames_sample_index=, sample_frame_rate=, **unet_kwargs) -> eps [2, c, n_cond + win, h, w].
    Returns the final latents with the condition frames re-inserted in front (:2149-2156)."""
    do_cfg = guidance_scale > 1.0
    unet_kwargs = unet_kwargs or {}
    euler = scheduler == "euler"
    sched = EulerOracle(**(scheduler_kwargs or {})) if euler else DDIMOracle()
    sched.set_timesteps(num_inference_steps)
    n_cond = 0 if condition_latents is None else condition_latents.shape[2]
    T = latents.shape[2]
    vis_idx, latent_index = condition_indices(n_cond, T, vision_condition_latent_index)
    gscales = guidance_schedule(guidance_scale, num_inference_steps, guidance_scale_end, guidance_scale_method)  # :1718-1723
    global_context = prepare_global_context(context_schedule, num_inference_steps, T, context_frames, context_stride,
                                            context_overlap, context_batch_size)
    n_t = len(sched.timesteps)
    keep = [1.0 - float(i / n_t < control_guidance_start or (i + 1) / n_t > control_guidance_end) for i in range(n_t)]  # :1700-1710
    # controlnet_fn(sample[(b t), c, h, w], t, text[(b t), L, D], cond[(b t), 3, H, W], conditioning_scale, guess_mode)
    #   -> (down residuals, mid residual); control_image [1, 3, n_cond + T, H, W], duplicated for CFG unless guess mode (:476-477)
    for i, t in enumerate(sched.timesteps):
        if max_steps is not None and i >= max_steps:
            break
        if i < start_step:
            continue
        noise_pred = torch.zeros((latents.shape[0] * (2 if do_cfg else 1), *latents.shape[1:]), dtype=latents.dtype)
        counter = torch.zeros((1, 1, T, 1, 1), dtype=latents.dtype)
        for context in global_context:
            latents_c = torch.cat([latents[:, :, c] for c in context])                       # :1902
            x = latents_c.repeat(2 if do_cfg else 1, 1, 1, 1, 1)                              # :1908-1910
            if euler:
                x = sched.scale_model_input(x, i)                                             # :1911 (identity for DDIM)
            sub_idx = None
            if latent_index is not None:
                win = len(context[0])
                sub_idx = torch.arange(win, dtype=torch.long) + n_cond                       # :1914-1920
            if condition_latents is not None:
                cond = torch.cat([condition_latents] * 2) if do_cfg else latents             # :1922-1926
                x = concat_with_index(cond, vis_idx, x, sub_idx)                                # :1939-1946
            extra = {}
            if controlnet_fn is not None:
                cond_scale = controlnet_conditioning_scale * keep[i]                          # :1235
                cctx = [vis_idx.tolist() + [ci + n_cond for ci in c] for c in context]        # :1953-1961
                ctrl = torch.cat([control_image[:, :, c] for c in cctx])                      # :1977-1979
                if guess_mode and do_cfg:                                                     # :1218-1226: conditional half only
                    cin, ctext = x[x.shape[0] // 2:], prompt_embeds.chunk(2)[1]
                else:
                    cin, ctext = x, prompt_embeds
                    ctrl = ctrl.repeat(2 if do_cfg else 1, 1, 1, 1, 1)                        # :476-477
                frames = cin.permute(0, 2, 1, 3, 4).reshape(-1, cin.shape[1], *cin.shape[3:])  # "b c t h w -> (b t) c h w" (:1236)
                text = ctext.repeat_interleave(cin.shape[2], dim=0)                           # :1242-1246
                cimg = ctrl.permute(0, 2, 1, 3, 4).reshape(-1, ctrl.shape[1], *ctrl.shape[3:])
                down, mid = controlnet_fn(frames, t, text, cimg, cond_scale, guess_mode)      # :1251-1260
                if guess_mode and do_cfg:                                                     # :1276-1288
                    down = [torch.cat([torch.zeros_like(d), d]) for d in down]
                    mid = torch.cat([torch.zeros_like(mid), mid])
                extra = dict(down_block_additional_residuals=down, mid_block_additional_residual=mid)
            eps = unet_fn(x, t, prompt_embeds, sample_index=sub_idx, vision_conditon_frames_sample_index=vis_idx,
                          sample_frame_rate=motion_speed, **unet_kwargs, **extra)             # :2045-2067
            if condition_latents is not None:
                eps = eps.index_select(2, sub_idx)                                            # :2068-2071
            for j, c in enumerate(context):
                noise_pred[:, :, c] = noise_pred[:, :, c] + eps                              # :2076-2078
                counter[:, :, c] = counter[:, :, c] + 1
        noise_pred = noise_pred / counter                                                     # :2079
        if do_cfg:
            u, tx = noise_pred.chunk(2)
            noise_pred = u + gscales[i] * (tx - u)                                            # :2101-2105
        if record is not None:
            record.append(noise_pred.clone())
        latents = sched.step(noise_pred, i if euler else int(t), latents)                     # :2112-2117
        if record_latents is not None:
            record_latents.append(latents.clone())
    if condition_latents is not None:
        latents = concat_with_index(condition_latents, vis_idx, latents, latent_index)        # :2149-2156
    return latents


def multi_shot_loop(unet_fn: Callable[..., Tensor], noises: List[Tensor], prompt_embeds: Tensor, condition_latents: Optional[Tensor],
                    n_vision_condition: int = 1, fix_condition_images: bool = False, **loop_kwargs) -> Tensor:
    """musev/pipelines/pipeline_controlnet_predictor.py:643-745 (``run_pipe_text2video``'s shot loop) over ``denoise_loop``: shot i
    starts from ``noises[i]``; from shot 1 on the condition latents are the last ``n_vision_condition`` frames of the previous
    shot's output (``out_latents_batch[:, :, -n_vision_condition:]``, :656-659) and the first ``n_vision_condition`` output frames
    (``result_overlap``, :662) are dropped before the shots are concatenated along t."""
    parts = []
    cond = condition_latents
    for i, noise in enumerate(noises):
        out = denoise_loop(unet_fn, noise, prompt_embeds, condition_latents=cond, **loop_kwargs)
        overlap = 0 if i == 0 else (n_vision_condition if cond is not None else 0)
        parts.append(out[:, :, overlap:])
        if cond is not None and n_vision_condition > 0 and not fix_condition_images:
            cond = out[:, :, -n_vision_condition:]
    return torch.cat(parts, dim=2)
