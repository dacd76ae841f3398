"""DDIMScheduler with the interface the reference pipeline uses (musev/schedulers/scheduling_ddim.py:136-302 on top of
diffusers' DDIMScheduler): ``set_timesteps``, ``timesteps``, ``scale_model_input`` (identity), ``init_noise_sigma``,
``step(...).prev_sample``.  Constants (betas, alphas_cumprod, timestep table) are host-side; the elementwise update
itself runs in the fused HIP kernel ``mv_cfg_ddim_step`` (no torch elementwise fallback).

Scope: epsilon prediction (the SD-1.5 configuration BASELINE.json's metric is quoted on) and v-prediction with the zero-terminal-SNR
beta rescale and "trailing" timestep spacing (the predictor's `enable_zero_snr` scheduler, pipeline_controlnet_predictor.py:270-282);
eta = 0, no clipping / thresholding.  "sample" prediction and eta > 0 raise NotImplementedError."""
from __future__ import annotations

from dataclasses import dataclass
from types import SimpleNamespace
from typing import Optional, Tuple, Union

import numpy as np
import torch

from .. import ops


def rescale_zero_terminal_snr(betas: torch.Tensor) -> torch.Tensor:
    """diffusers' rescale_zero_terminal_snr (Lin et al., "Common Diffusion Noise Schedules and Sample Steps are Flawed", alg. 1):
    shift and scale sqrt(alpha_bar) so that the last timestep has zero SNR and the first keeps its value"""
    alphas_bar_sqrt = torch.cumprod(1.0 - betas, dim=0).sqrt()
    a0, aT = alphas_bar_sqrt[0].clone(), alphas_bar_sqrt[-1].clone()
    alphas_bar_sqrt = (alphas_bar_sqrt - aT) * (a0 / (a0 - aT))
    alphas_bar = alphas_bar_sqrt ** 2
    alphas = torch.cat([alphas_bar[0:1], alphas_bar[1:] / alphas_bar[:-1]])
    return 1.0 - alphas


@dataclass
class DDIMSchedulerOutput:
    prev_sample: torch.Tensor
    pred_original_sample: Optional[torch.Tensor] = None


class DDIMScheduler:
    order = 1

    def __init__(self, num_train_timesteps: int = 1000, beta_start: float = 0.00085, beta_end: float = 0.012,
                 beta_schedule: str = "scaled_linear", clip_sample: bool = False, set_alpha_to_one: Optional[bool] = None,
                 steps_offset: int = 1, prediction_type: str = "epsilon", timestep_spacing: str = "leading",
                 rescale_betas_zero_snr: bool = False):
        # set_alpha_to_one = None resolves to what the reference ends up with for the same keywords: False for the SD-1.5
        # scheduler_config.json the MuseV checkpoints ship (leading spacing, no beta rescale -- the value that file sets), True
        # -- diffusers' constructor default, which the reference's DDIMScheduler(...) call relies on -- for the zero-SNR / trailing
        # configuration (trailing spacing steps to prev_timestep < 0 on the last step, where final_alpha_cumprod is used)
        if set_alpha_to_one is None:
            set_alpha_to_one = bool(rescale_betas_zero_snr or timestep_spacing == "trailing")
        # constants are built with torch on the host exactly like diffusers does (fp32 linspace / cumprod), so that
        # alpha_bar_t matches the reference bit for bit
        if beta_schedule == "scaled_linear":
            betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
        elif beta_schedule == "linear":
            betas = torch.linspace(beta_start, beta_end, num_train_timesteps, dtype=torch.float32)
        else:
            raise NotImplementedError(f"{beta_schedule} is not implemented for {self.__class__}")
        if prediction_type not in ("epsilon", "v_prediction") or clip_sample:
            raise NotImplementedError("epsilon / v_prediction without clipping")
        if timestep_spacing not in ("leading", "trailing"):
            raise NotImplementedError("'leading' (SD-1.5 config) or 'trailing' (zero-SNR config) timestep spacing")
        if rescale_betas_zero_snr:
            betas = rescale_zero_terminal_snr(betas)
        self.config = SimpleNamespace(num_train_timesteps=num_train_timesteps, beta_start=beta_start, beta_end=beta_end,
                                      beta_schedule=beta_schedule, clip_sample=clip_sample, set_alpha_to_one=set_alpha_to_one,
                                      steps_offset=steps_offset, prediction_type=prediction_type,
                                      timestep_spacing=timestep_spacing, rescale_betas_zero_snr=rescale_betas_zero_snr)
        self.alphas_cumprod = torch.cumprod(1.0 - betas, dim=0)
        self.final_alpha_cumprod = torch.tensor(1.0) if set_alpha_to_one else self.alphas_cumprod[0]
        self.init_noise_sigma = 1.0
        self.num_inference_steps: Optional[int] = None
        self.timesteps = torch.from_numpy(np.arange(0, num_train_timesteps)[::-1].copy().astype(np.int64))

    def scale_model_input(self, sample: torch.Tensor, timestep=None) -> torch.Tensor:
        return sample

    def set_timesteps(self, num_inference_steps: int, device=None) -> None:
        if num_inference_steps > self.config.num_train_timesteps:
            raise ValueError("`num_inference_steps` cannot be larger than `num_train_timesteps`")
        self.num_inference_steps = num_inference_steps
        if self.config.timestep_spacing == "trailing":
            ratio = self.config.num_train_timesteps / num_inference_steps
            ts = np.round(np.arange(self.config.num_train_timesteps, 0, -ratio)).astype(np.int64) - 1
        else:
            ratio = self.config.num_train_timesteps // num_inference_steps
            ts = (np.arange(0, num_inference_steps) * ratio).round()[::-1].copy().astype(np.int64) + self.config.steps_offset
        self.timesteps = torch.from_numpy(ts).to(device) if device is not None else torch.from_numpy(ts)

    def alphas_for(self, timestep: int) -> Tuple[float, float]:
        """(alpha_prod_t, alpha_prod_t_prev) of reference :198-208"""
        if self.num_inference_steps is None:
            raise ValueError("Number of inference steps is 'None', you need to run 'set_timesteps' after creating the scheduler")
        prev = timestep - self.config.num_train_timesteps // self.num_inference_steps
        a_t = float(self.alphas_cumprod[timestep])
        a_prev = float(self.alphas_cumprod[prev]) if prev >= 0 else float(self.final_alpha_cumprod)
        return a_t, a_prev

    # ---- interface of musev_amd.pipelines.parallel_denoise ----------------------------------------------------------
    def input_scale(self, step_index: int) -> float:
        return 1.0  # scale_model_input is the identity for DDIM

    def consume_step_noise(self, shape, dtype, device, generator, noise_type: str = "random", w_ind_noise: float = 0.5) -> None:
        """DDIM draws step noise only when eta > 0 (scheduling_ddim.py:266-295); this loop runs eta = 0: nothing to consume"""
        return

    def add_noise(self, original: torch.Tensor, noise: torch.Tensor, step_index: int) -> torch.Tensor:
        """sqrt(alpha_bar_t) x0 + sqrt(1 - alpha_bar_t) noise at the timestep of schedule entry ``step_index`` (diffusers
        DDIMScheduler.add_noise; the img2img start of the pipeline, pipeline_controlnet.py:414,423)"""
        a_t = float(self.alphas_cumprod[int(self.timesteps[int(step_index)])])
        return (a_t ** 0.5) * original + ((1.0 - a_t) ** 0.5) * noise

    def loop_update(self, latents: torch.Tensor, eps_acc: torch.Tensor, counter: torch.Tensor, guidance: float,
                    step_index: int, timestep) -> None:
        """fused average / CFG / DDIM step on the loop state (latents fp32 [C, T, HW], in place)"""
        a_t, a_prev = self.alphas_for(int(timestep))
        if self.config.prediction_type == "v_prediction":
            cx, ce = self.v_coefficients(a_t, a_prev)
            ops.cfg_affine_step(latents, eps_acc, counter, guidance, cx, ce)
        else:
            ops.cfg_ddim_step(latents, eps_acc, counter, guidance, a_t, a_prev)

    @staticmethod
    def v_coefficients(a_t: float, a_prev: float) -> Tuple[float, float]:
        """x_prev = cx x + ce v for v-prediction at eta = 0 (scheduling_ddim.py:224-231,257-264): x0 = sqrt(a_t) x - sqrt(1 - a_t) v,
        eps = sqrt(a_t) v + sqrt(1 - a_t) x, x_prev = sqrt(a_prev) x0 + sqrt(1 - a_prev) eps -- an affine map of (x, v), so the
        fused CFG + affine step kernel of the Euler scheduler serves it"""
        sa, sb = a_t ** 0.5, (1.0 - a_t) ** 0.5
        pa, pb = a_prev ** 0.5, (1.0 - a_prev) ** 0.5
        return pa * sa + pb * sb, pb * sa - pa * sb

    def step(self, model_output: torch.Tensor, timestep: int, sample: torch.Tensor, eta: float = 0.0,
             use_clipped_model_output: bool = False, generator=None, variance_noise=None, return_dict: bool = True,
             w_ind_noise: float = 0.5, noise_type: str = "random") -> Union[DDIMSchedulerOutput, Tuple]:
        if eta != 0.0 or use_clipped_model_output or variance_noise is not None:
            raise NotImplementedError("eta > 0 / clipped model output are outside the DDIM scope of this build")
        if not sample.is_cuda:
            raise RuntimeError("DDIMScheduler.step runs on the GPU only")
        a_t, a_prev = self.alphas_for(int(timestep))
        shape = sample.shape
        x = sample.detach().to(torch.float32).reshape(1, -1, 1, 1).clone().view(-1, 1, 1)   # [C=n, T=1, HW=1] view of all elements
        eps = model_output.detach().to(torch.float32).reshape(1, -1, 1, 1).contiguous()
        ones = torch.ones(1, dtype=torch.float32, device=sample.device)
        if self.config.prediction_type == "v_prediction":
            cx, ce = self.v_coefficients(a_t, a_prev)
            ops.cfg_affine_step(x, eps, ones, 0.0, cx, ce)
        else:
            ops.cfg_ddim_step(x, eps, ones, 0.0, a_t, a_prev)
        prev = x.view(shape).to(sample.dtype)
        if not return_dict:
            return (prev,)
        return DDIMSchedulerOutput(prev_sample=prev)
