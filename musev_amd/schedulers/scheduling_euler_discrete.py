"""EulerDiscreteScheduler as the reference's predictor installs it by default
(musev/pipelines/pipeline_controlnet_predictor.py:258-261: ``EulerDiscreteScheduler.from_config(pipeline.scheduler.config)``;
step override musev/schedulers/scheduling_euler_discrete.py:47-167 on top of diffusers' EulerDiscreteScheduler).

Host side: betas / alphas_cumprod / sigma table / timestep table exactly as diffusers builds them (numpy fp32/fp64
steps restated from upstream v0.24 -- the base class is NOT vendored in the reference).  Device side: with
``s_churn = 0`` (the only value the reference pipeline passes, so gamma = 0 and the noise drawn at
scheduling_euler_discrete.py:120-131 never enters the sample) one step is

    x0 = x - sigma * eps ;  derivative = (x - x0) / sigma = eps ;  x_prev = x + eps * (sigma_next - sigma)

which the parallel-denoise loop runs fused with the window average and CFG in ``mv_cfg_affine_step`` (cx = 1,
ce = sigma_next - sigma); ``scale_model_input`` (x / sqrt(sigma^2 + 1)) is applied to the latents a window is gathered
from.  Scope: epsilon prediction, linear sigma interpolation, no Karras sigmas, s_churn = 0."""
from __future__ import annotations

from types import SimpleNamespace
from typing import Optional

import numpy as np
import torch

from .. import ops


class EulerDiscreteScheduler:
    order = 1

    def __init__(self, num_train_timesteps: int = 1000, beta_start: float = 0.00085, beta_end: float = 0.012,
                 beta_schedule: str = "scaled_linear", prediction_type: str = "epsilon", interpolation_type: str = "linear",
                 use_karras_sigmas: bool = False, timestep_spacing: str = "linspace", steps_offset: int = 0):
        if beta_schedule == "scaled_linear":
            betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
        elif beta_schedule == "linear":
            betas = torch.linspace(beta_start, beta_end, num_train_timesteps, dtype=torch.float32)
        else:
            raise NotImplementedError(f"{beta_schedule} is not implemented for {self.__class__}")
        if prediction_type != "epsilon":
            raise NotImplementedError("only epsilon prediction (SD-1.5 config)")
        if interpolation_type != "linear" or use_karras_sigmas:
            raise NotImplementedError("only linear sigma interpolation without Karras sigmas")
        if timestep_spacing not in ("linspace", "leading", "trailing"):
            raise ValueError(f"{timestep_spacing} is not supported. Please make sure to choose one of 'linspace', 'leading' or 'trailing'.")
        self.config = SimpleNamespace(num_train_timesteps=num_train_timesteps, beta_start=beta_start, beta_end=beta_end,
                                      beta_schedule=beta_schedule, prediction_type=prediction_type,
                                      interpolation_type=interpolation_type, use_karras_sigmas=use_karras_sigmas,
                                      timestep_spacing=timestep_spacing, steps_offset=steps_offset)
        self.alphas_cumprod = torch.cumprod(1.0 - betas, dim=0)
        sigmas = (((1 - self.alphas_cumprod) / self.alphas_cumprod) ** 0.5).numpy()
        self._train_sigmas = sigmas
        self.sigmas = torch.from_numpy(np.concatenate([sigmas[::-1], [0.0]]).astype(np.float32))
        self.timesteps = torch.from_numpy(np.linspace(0, num_train_timesteps - 1, num_train_timesteps, dtype=float)[::-1].copy())
        self.num_inference_steps: Optional[int] = None

    # ---- diffusers interface -------------------------------------------------------------------------------------
    @classmethod
    def from_config(cls, config, **overrides) -> "EulerDiscreteScheduler":
        """``EulerDiscreteScheduler.from_config(pipeline.scheduler.config)`` (pipeline_controlnet_predictor.py:258-261): ``config`` is
        another scheduler's config (a mapping or an object with attributes); keys this constructor does not know are ignored,
        as diffusers' ConfigMixin does."""
        import inspect
        names = [n for n in inspect.signature(cls.__init__).parameters if n != "self"]
        get = (lambda k: config[k]) if isinstance(config, dict) else (lambda k: getattr(config, k))
        has = (lambda k: k in config) if isinstance(config, dict) else (lambda k: hasattr(config, k))
        kw = {n: get(n) for n in names if has(n)}
        kw.update({k: v for k, v in overrides.items() if k in names})
        return cls(**kw)

    def step(self, model_output: torch.Tensor, timestep, sample: torch.Tensor, generator=None, noise_type: str = "random",
             w_ind_noise: float = 0.5, return_dict: bool = False):
        """One Euler step with s_churn = 0 on plain tensors (musev/schedulers/scheduling_euler_discrete.py:47-167): the per-step
        noise is drawn with ``model_output``'s dtype and shape (:120-131) -- it only advances ``generator`` -- then
        ``prev = sample + model_output * (sigma_next - sigma)`` in fp32, cast back to ``model_output``'s dtype.  The loop itself uses
        the fused ``loop_update``; this is the drop-in for callers that step the scheduler themselves."""
        i = self._index_of(timestep)
        self.consume_step_noise(model_output.shape, model_output.dtype, model_output.device, generator, noise_type, w_ind_noise)
        sigma, sigma_next = float(self.sigmas[i]), float(self.sigmas[i + 1])
        prev = (sample.float() + model_output.float() * (sigma_next - sigma)).to(model_output.dtype)
        pred_original = (sample.float() - sigma * model_output.float()).to(model_output.dtype)
        if return_dict:
            return SimpleNamespace(prev_sample=prev, pred_original_sample=pred_original)
        return (prev, pred_original)

    @property
    def init_noise_sigma(self) -> float:
        max_sigma = float(self.sigmas.max())
        if self.config.timestep_spacing in ("linspace", "trailing"):
            return max_sigma
        return (max_sigma ** 2 + 1) ** 0.5

    def add_noise(self, original: torch.Tensor, noise: torch.Tensor, step_index: int) -> torch.Tensor:
        """x0 + sigma_t noise at schedule entry ``step_index`` (diffusers EulerDiscreteScheduler.add_noise; the img2img start of the
        pipeline, pipeline_controlnet.py:414,423)"""
        return original + float(self.sigmas[int(step_index)]) * noise

    def set_timesteps(self, num_inference_steps: int, device=None) -> None:
        n_train = self.config.num_train_timesteps
        self.num_inference_steps = num_inference_steps
        sp = self.config.timestep_spacing
        if sp == "linspace":
            timesteps = np.linspace(0, n_train - 1, num_inference_steps, dtype=np.float32)[::-1].copy()
        elif sp == "leading":
            ratio = n_train // num_inference_steps
            timesteps = (np.arange(0, num_inference_steps) * ratio).round()[::-1].copy().astype(np.float32)
            timesteps += self.config.steps_offset
        else:  # trailing
            ratio = n_train / num_inference_steps
            timesteps = (np.arange(n_train, 0, -ratio)).round().copy().astype(np.float32)
            timesteps -= 1
        sigmas = np.interp(timesteps, np.arange(0, len(self._train_sigmas)), self._train_sigmas)
        sigmas = np.concatenate([sigmas, [0.0]]).astype(np.float32)
        self.sigmas = torch.from_numpy(sigmas)
        self.timesteps = torch.from_numpy(timesteps)
        if device is not None:
            self.timesteps = self.timesteps.to(device)

    def scale_model_input(self, sample: torch.Tensor, timestep=None, step_index: Optional[int] = None) -> torch.Tensor:
        if step_index is None:
            step_index = self._index_of(timestep)
        return sample * self.input_scale(step_index)

    def _index_of(self, timestep) -> int:
        ts = self.timesteps.detach().cpu().numpy()
        idx = np.nonzero(ts == np.float32(float(timestep)))[0]
        if len(idx) == 0:
            raise ValueError(f"timestep {timestep} is not on the schedule")
        return int(idx[0])  # diffusers takes the first match (the second only for img2img restarts)

    # ---- interface of musev_amd.pipelines.parallel_denoise ----------------------------------------------------------
    def input_scale(self, step_index: int) -> float:
        """scale_model_input factor 1 / sqrt(sigma_i^2 + 1)"""
        sigma = float(self.sigmas[step_index])
        return 1.0 / (sigma ** 2 + 1) ** 0.5

    def consume_step_noise(self, shape, dtype: torch.dtype, device, generator, noise_type: str = "random",
                           w_ind_noise: float = 0.5) -> None:
        """The reference's step draws a noise tensor of the model-output shape on EVERY step (:120-131) and only uses it
        when s_churn > 0 -- which the pipeline never passes.  The draw still advances the caller's generator, so a seeded
        multi-shot run (the predictor hands the same generator to the next shot) only reproduces if it is consumed
        here too.  No-op without a generator."""
        if generator is None:
            return
        from ..utils import noise_util
        if noise_type == "video_fusion":
            noise_util.video_fusion_noise(shape=tuple(shape), dtype=dtype, device=device, w_ind_noise=w_ind_noise, generator=generator)
        elif noise_type == "random":
            noise_util.random_noise(shape=tuple(shape), dtype=dtype, device=device, generator=generator)
        else:
            raise ValueError(f"noise_type must be 'random' or 'video_fusion', got {noise_type}")

    def loop_update(self, latents: torch.Tensor, eps_acc: torch.Tensor, counter: torch.Tensor, guidance: float,
                    step_index: int, timestep) -> None:
        """fused average / CFG / Euler step on the loop state (latents fp32 [C, T, HW], in place)"""
        sigma, sigma_next = float(self.sigmas[step_index]), float(self.sigmas[step_index + 1])
        ops.cfg_affine_step(latents, eps_acc, counter, guidance, 1.0, sigma_next - sigma)
