"""Visual-conditioned parallel denoising: the hot loop of MusevControlNetPipeline.__call__
(reference musev/pipelines/pipeline_controlnet.py:1832-2156) on HIP kernels, optionally sharded over the GPUs of one
node.

Per denoise step the reference (:1847-2117)
  zeroes an accumulator, loops over the windows of ``prepare_global_context`` (gather frames, duplicate for CFG,
  prepend the vision-condition latents, UNet, drop the condition frames, scatter-add the prediction and a coverage
  counter), divides, applies classifier-free guidance and calls ``scheduler.step``.
Here
  * the window input is built directly in the UNet's channels-last layout by one kernel (mv_window_gather),
  * every (window, CFG half) pair is an independent *unit* of work: a rank owns a contiguous slice of the unit list,
    runs the UNet on its units (both halves of a window batched when it owns both) and contributes the predictions
    to the step's exchange (RCCL all-gather of the fp32 predictions, <= 786 KB per unit, one collective per exchange slot issued
    as soon as the slot is filled so that it runs under the next window's forward) -- there is no other collective,
  * every rank then performs the identical, order-fixed accumulation / average / CFG / DDIM update
    (mv_window_units_reduce -- one table-driven gather-reduce launch -- + mv_cfg_ddim_step), so the replicated latents stay
    bit-identical across ranks.
Latents are kept in fp32 [C, T, HW]; the reference keeps them in the model dtype (fp16 on GPU)."""
from __future__ import annotations

import os
from dataclasses import dataclass
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import torch

from .. import ops
from ..models.runtime import PrefixMemo
from ..schedulers.scheduling_ddim import DDIMScheduler  # default; EulerDiscreteScheduler offers the same loop interface
from ..utils.timesteps_util import generate_parameters_with_timesteps
from .context import prepare_global_context


@dataclass
class Unit:
    window: int
    half: int  # 0 = unconditional, 1 = text-conditioned (prompt_embeds order, text_emb_util.py:428)


def shard_units(n_windows: int, halves: int, world: int) -> List[List[Unit]]:
    """contiguous slices of the (window-major) unit list; sizes differ by at most one"""
    units = [Unit(w, hf) for w in range(n_windows) for hf in range(halves)]
    per, rem = divmod(len(units), world)
    out, s = [], 0
    for r in range(world):
        n = per + (1 if r < rem else 0)
        out.append(units[s:s + n])
        s += n
    return out


def group_units(units: Sequence[Unit]) -> List[Tuple[int, List[int]]]:
    """[(window, [halves...])]: halves of the same window are batched into one UNet call"""
    out: List[Tuple[int, List[int]]] = []
    for u in units:
        if out and out[-1][0] == u.window:
            out[-1][1].append(u.half)
        else:
            out.append((u.window, [u.half]))
    return out


class ParallelDenoiser:
    _device_check = True  # tests of the sharding logic (gloo, CPU, fake kernels) switch this off
    always_exchange = False  # diagnostic: take the multi-rank exchange path in a 1-rank group as well (see __call__)
    time_exchange = False    # diagnostic (bench.py): HIP event pairs around [wait for the slots' all-gathers + table reduce] of every step
    exchange_events: list = []

    def __init__(self, unet, scheduler: Optional[DDIMScheduler] = None, *, context_frames: int = 12,
                 context_overlap: int = 4, context_stride: int = 1, context_schedule: str = "uniform",
                 context_batch_size: int = 1, use_graphs: bool = True, share_cfg_prefix: bool = True, odd_unit_lane: bool = True):
        self.unet = unet
        self.exchange_events = []
        # hipGraph capture of the per-window UNet forward (~1 500 kernel launches): replayed once per window and step,
        # so the host only issues the loop glue.  A failed capture raises (MUSEV_NO_GRAPH=1 / use_graphs=False = eager on purpose).
        self.use_graphs = use_graphs and os.environ.get("MUSEV_NO_GRAPH", "0") != "1"  # env knob for per-kernel PMC profiling
        self._graphs: Dict[tuple, "_GraphedForward"] = {}
        # the two CFG halves of a window as two batch-1 forwards on two HIP streams (+2.7 % frames/s at config 2,
        # profiles/r01j): MUSEV_HALF_STREAMS=0 restores the single batch-2 forward (per-kernel profiling)
        self.half_streams = os.environ.get("MUSEV_HALF_STREAMS", "1") == "1"
        # the half-independent front of the network once for both CFG halves of a window (see _unet_rows)
        self.share_cfg_prefix = share_cfg_prefix
        # A rank with an ODD number of units (24 units over 8 GPUs = 3) owns one two-half window and one lone half.  Run one after
        # the other the lone batch-1 forward has the GPU to itself at ~0.65 of a pair's time; run CONCURRENTLY with the neighbouring
        # pair (its graph replayed on a third stream) three half-forwards share the GPU like a pair and a half.  Used from the second
        # executed step of a call on (every signature captured by then), with graphs, without a ControlNet (whose per-length static
        # control-frame buffer is shared by the groups).  odd_unit_lane=False keeps the groups one after the other.
        self.odd_unit_lane = odd_unit_lane
        self._t_bufs: Dict[str, torch.Tensor] = {}
        self._warm: Dict[tuple, bool] = {}  # signatures whose lazily built caches (packed weights, K/V projections) are filled
        self.scheduler = scheduler or DDIMScheduler()
        self.context_frames, self.context_overlap = context_frames, context_overlap
        self.context_stride, self.context_schedule = context_stride, context_schedule
        if context_batch_size != 1:
            # with vision-condition latents the reference itself only supports one window per batch (Appendix B.7)
            raise NotImplementedError("context_batch_size != 1")

    def windows(self, time_size: int, num_inference_steps: int) -> List[List[int]]:
        gc = prepare_global_context(self.context_schedule, num_inference_steps, time_size, self.context_frames,
                                    self.context_stride, self.context_overlap, 1)
        return [c[0] for c in gc]

    def _unet_dtype(self) -> torch.dtype:
        return _first_param_dtype(self.unet)

    @torch.no_grad()
    def __call__(self, latents: torch.Tensor, prompt_embeds: torch.Tensor, *, num_inference_steps: int = 20,
                 guidance_scale: float = 7.5, condition_latents: Optional[torch.Tensor] = None, motion_speed: float = 8.0,
                 unet_kwargs: Optional[dict] = None, group=None, callback: Optional[Callable] = None,
                 reinsert_condition: bool = True, max_steps: Optional[int] = None, guidance_scale_end: Optional[float] = None,
                 guidance_scale_method: str = "linear", generator=None, noise_type: str = "random",
                 w_ind_noise: float = 0.5, controlnet=None, control_image: Optional[torch.Tensor] = None,
                 controlnet_conditioning_scale: float = 1.0, control_guidance_start: float = 0.0,
                 control_guidance_end: float = 1.0, guess_mode: bool = False, start_step: int = 0,
                 vision_condition_latent_index: Optional[Sequence[int]] = None) -> torch.Tensor:
        """latents [1, c, T, h, w] (frames to generate, any float dtype, on the GPU); prompt_embeds [2, L, D] =
        [negative, positive] (or [1, L, D] when guidance_scale <= 1); condition_latents [1, c, n_cond, h, w] or None.
        ``group``: torch.distributed process group to shard the units over (None = this process alone).
        ``guidance_scale_end`` / ``guidance_scale_method``: per-step guidance schedule from guidance_scale to
        guidance_scale_end ("linear", "two_stage", "three_stage", "fix_two_stage"; reference :1718-1723, :2103).
        ``generator`` (+ ``noise_type``, ``w_ind_noise``): the caller's torch.Generator; the reference's Euler step draws
        (and, at s_churn = 0, discards) one noise tensor per step from it, which is reproduced so that whatever the caller
        draws AFTER the loop matches a seeded reference run.
        ``controlnet`` (+ ``control_image`` [1, 3, n_cond + T, 8h, 8w], ``controlnet_conditioning_scale``,
        ``control_guidance_start`` / ``_end``, ``guess_mode``): the per-window ControlNet call of the reference
        (get_controlnet_emb, :1202-1291; window gather of the control frames :1947-1976): its residuals are added to the
        UNet's skips / mid block.  The control frames of a window are the condition frames followed by the window's frames.
        ``max_steps``: run only the first max_steps steps of the num_inference_steps-long schedule (smoke / bench helper).
        ``start_step``: enter the schedule at this step with ``latents`` being the latents of that step (what the reference's
        ``get_timesteps(strength)`` does for img2img starts, :1613-1622; the parity tests use it to start a step from recorded latents).
        ``vision_condition_latent_index``: where the condition frames sit among the n_cond + T output frames, -1 = the last one
        (prepare_condition_latents_and_index, :966-1040; the CLI's ``condition_images_index``); None = in front.  The reference's
        literal behaviour is reproduced (tests/golden/reference_condition_index.json): the final re-insert (:2149-2156) uses these
        positions, but every WINDOW's input is built by writing the condition latents at the same GLOBAL positions into a zero tensor
        of n_cond + win frames and the window's frames at n_cond.. afterwards (:1914-1946) -- a position outside a window's
        n_cond + win slots raises IndexError as torch's index_copy_ does (any run with more than one window and a -1), a tail
        condition frame is overwritten by the window's last frame and the slot it would have had in front stays zero, while the
        UNet is told that the given positions hold the condition frames (temb zeroing, reference-only attention).
        Returns fp32 latents [1, c, n_cond + T, h, w] (condition frames re-inserted at their positions, in front by default)."""
        if latents.ndim != 5 or latents.shape[0] != 1:
            raise ValueError("latents must be [1, c, T, h, w]")
        if self._device_check and not latents.is_cuda:
            raise RuntimeError("ParallelDenoiser runs on the GPU only")
        dev = latents.device
        _, c, T, h, w = latents.shape
        hw = h * w
        do_cfg = guidance_scale > 1.0
        halves = 2 if do_cfg else 1
        if prompt_embeds.shape[0] != halves:
            raise ValueError(f"prompt_embeds batch must be {halves} for guidance_scale={guidance_scale}")
        if condition_latents is not None and not do_cfg:
            # the reference substitutes `latents` for the condition in this branch (:1922-1926), which only works by
            # accident; parity is defined for the CFG path
            raise NotImplementedError("vision-condition latents without classifier-free guidance")
        unet_kwargs = dict(unet_kwargs or {})
        n_cond = 0 if condition_latents is None else condition_latents.shape[2]
        lat = latents.detach().to(torch.float32).reshape(c, T, hw).contiguous().clone()
        cond = None if condition_latents is None else condition_latents.detach().to(torch.float32).reshape(c, n_cond, hw).contiguous()

        sched = self.scheduler
        sched.set_timesteps(num_inference_steps)
        timesteps = [float(t) for t in sched.timesteps.tolist()]  # integral for DDIM, fractional for Euler ("linspace")
        guidance = [float(g) for g in generate_parameters_with_timesteps(start=guidance_scale, stop=guidance_scale_end,
                                                                        num=len(timesteps), method=guidance_scale_method)]
        wins = self.windows(T, num_inference_steps)
        # windows may differ in length: `uniform_v2` (the CLI default, scripts/inference/text2video.py:499-505) ends with a
        # short window whenever (T - overlap) % (window - overlap) != 0, and the reference runs every window as its own UNet
        # call of whatever length (pipeline_controlnet.py:1900-1946).  Exchange slots are sized by the longest window.
        win_len = max(len(wd) for wd in wins)
        for wd in wins:
            if len(set(wd)) != len(wd):
                # `uniform` with context_stride > 1 can wrap a window onto a frame it already holds; the reference's
                # index-assign then keeps one of the two predictions while the coverage counter counts the frame once --
                # an ill-defined case (and a write race for a parallel scatter), refused instead of silently differing
                raise NotImplementedError(f"window {wd} visits a frame twice")
        idx_dev = [torch.tensor(wd, dtype=torch.int32, device=dev) for wd in wins]
        counter = torch.zeros(T, dtype=torch.float32, device=dev)
        for wd in wins:  # coverage count is the same at every step: computed once (:2078)
            counter[torch.tensor(wd, device=dev, dtype=torch.long)] += 1.0
        if bool((counter == 0).any()):
            raise RuntimeError("window schedule leaves frames uncovered")

        world, rank = 1, 0
        if group is not None:
            world = torch.distributed.get_world_size(group)
            rank = torch.distributed.get_rank(group)
        shards = shard_units(len(wins), halves, world)
        my_groups = group_units(shards[rank])
        max_units = max(len(s) for s in shards)
        unit_elems = win_len * hw * c
        # the exchange path (send slots -> all-gather -> table-driven reduce); `always_exchange` runs it in a 1-rank group too, which
        # lets a 1-GPU box drive RCCL through exactly the calls of a multi-GPU run (tests/test_pipeline_gpu.py)
        exchange = world > 1 or (group is not None and self.always_exchange)
        send = torch.zeros((max_units, win_len * hw, c), dtype=torch.float32, device=dev) if exchange else None
        recv = torch.empty((max_units, world, win_len * hw, c), dtype=torch.float32, device=dev) if exchange else None  # [slot][rank]
        cover = None
        if exchange:
            # (slot, position-in-window) pairs covering every (half, frame), in unit order (= window order: the shards are
            # contiguous slices of the window-major unit list): ONE gather-reduce launch per step replaces world x max_units
            # scatter-adds, and its summation order is the same on every rank (bit-identical replicas)
            pairs = [[[] for _ in range(T)] for _ in range(halves)]
            for r in range(world):
                for k, u in enumerate(shards[r]):
                    for j, f in enumerate(wins[u.window]):
                        pairs[u.half][f].append((k * world + r, j))
            maxc = max(len(e) for hp in pairs for e in hp)
            tab = torch.full((halves, T, maxc, 2), -1, dtype=torch.int32)
            for hf in range(halves):
                for f in range(T):
                    for q, (slot_, j) in enumerate(pairs[hf][f]):
                        tab[hf, f, q, 0], tab[hf, f, q, 1] = slot_, j
            cover = tab.to(dev)

        # vision_condition_latent_index / latent_index as prepare_condition_latents_and_index builds them (:995-1029)
        vis_idx, latent_index, cond_slot = None, None, None
        if n_cond:
            total_frames = n_cond + T
            vis_idx = (list(range(n_cond)) if vision_condition_latent_index is None
                       else [int(i) if int(i) != -1 else total_frames - 1 for i in vision_condition_latent_index])
            if len(vis_idx) != n_cond:
                raise IndexError(f"vision_condition_latent_index names {len(vis_idx)} positions for {n_cond} condition frames")
            if len(set(vis_idx)) != n_cond:
                raise ValueError("vision_condition_latent_index visits a position twice (index_copy_ is undefined there)")
            for wd in wins:
                if min(vis_idx) < 0 or max(vis_idx) >= n_cond + len(wd):
                    # index_copy_ into the window's n_cond + win slots (data_util.py:457): the reference fails here as well
                    raise IndexError(f"vision-condition position {max(vis_idx)} is outside a window's {n_cond + len(wd)} input frames "
                                     "(the reference scatters GLOBAL positions into every window's input, pipeline_controlnet.py:1939-1946)")
            latent_index = sorted(set(range(total_frames)) - set(vis_idx))
            if vis_idx != list(range(n_cond)):
                cond_slot = torch.tensor(vis_idx, dtype=torch.int32, device=dev)
        # sub_latent_index_c = arange(len(window)) + n_cond (:1914-1920), one tensor per distinct window length
        sub_idx_by_len = {n: ((torch.arange(n, dtype=torch.long, device=dev) + n_cond) if n_cond else None)
                          for n in sorted({len(wd) for wd in wins})}
        # static timestep buffer (graph input): one per device for the lifetime of this object, so that the captures of one
        # call are reused by the next (multi-shot generation replays the same graphs)
        t_dev = self._t_bufs.get(str(dev))
        if t_dev is None:
            t_dev = self._t_bufs[str(dev)] = torch.zeros(1, dtype=torch.float32, device=dev)
        embeds = prompt_embeds.to(dev)
        eps_acc = torch.empty((halves, c, T, hw), dtype=torch.float32, device=dev)

        # ---- ControlNet conditioning (optional) ----
        cn_keep, ctrl_frames, ctrl_bufs, text_rep_by_len = None, None, None, None
        if controlnet is not None:
            if control_image is None or control_image.ndim != 5 or control_image.shape[0] != 1 or control_image.shape[2] != n_cond + T:
                raise ValueError("control_image must be [1, c, n_cond + T, H, W] (condition frames first)")
            n_steps = len(timesteps)
            # controlnet_keep (:1700-1710 of the reference, the diffusers recipe): 0 outside [start, end] of the schedule
            cn_keep = [1.0 - float(i / n_steps < control_guidance_start or (i + 1) / n_steps > control_guidance_end)
                       for i in range(n_steps)]
            frames_all = control_image[0].to(dev).permute(1, 0, 2, 3)  # [n_cond + T, c, H, W]
            ctrl_frames = []
            for wd in wins:  # controlnet_context = condition indices + (window indices + n_cond)   (:1953-1961)
                sel = torch.tensor((vis_idx or []) + [i + n_cond for i in wd], dtype=torch.long, device=dev)
                ctrl_frames.append(frames_all.index_select(0, sel).to(torch.float16).contiguous())
            # static buffers (one per window length): the captured graphs read the window's frames from here
            ctrl_bufs = {n: torch.empty_like(next(f for f, wd in zip(ctrl_frames, wins) if len(wd) == n)) for n in sub_idx_by_len}
            # align_repeat_tensor_single_dim(prompt_embeds, (b t)) (:1242-1246): one row of text per frame, per CFG half
            text_rep_by_len = {n: embeds.repeat_interleave(n_cond + n, dim=0).contiguous() for n in sub_idx_by_len}

        steps_done = 0  # steps executed by THIS call (the first one captures / warms every signature: groups one after the other)
        for step, t in enumerate(timesteps):
            if max_steps is not None and step >= max_steps:
                break
            if step < start_step:
                continue
            if not exchange:
                eps_acc.zero_()  # (the multi-rank gather-reduce overwrites it)
            t_dev.fill_(float(t))
            # scheduler.scale_model_input (:1911): identity for DDIM, 1/sqrt(sigma^2+1) for Euler; the vision-condition
            # latents are concatenated AFTER the scaling in the reference (:1922-1946) and stay unscaled
            in_scale = sched.input_scale(step)
            lat_in = lat if in_scale == 1.0 else lat * in_scale
            slot = 0
            works = []
            cn_on = False
            if controlnet is not None:
                cond_scale = float(controlnet_conditioning_scale) * cn_keep[step]   # :1229-1236
                cn_on = cond_scale != 0.0  # a zero scale makes every residual zero: adding them is a no-op, skip the network

            def run_group(wi, hs):
                wl = len(wins[wi])
                tw = n_cond + wl
                # (hi_lo: the fp32 latents as two fp16 halves -- conv_in sees them unrounded, ops.CARRY)
                x = ops.window_gather(lat_in, cond, idx_dev[wi], n_cond, len(hs), hi_lo=ops.CARRY and controlnet is None and 18 * c <= 128,
                                      **({} if cond_slot is None else {"cond_slot": cond_slot}))
                cn_ = None
                if controlnet is not None and cn_on:
                    ctrl_bufs[wl].copy_(ctrl_frames[wi])
                    cn_ = (controlnet, ctrl_bufs[wl], text_rep_by_len[wl], cond_scale, bool(guess_mode))
                # (window_gather builds every owned half from the same latents: the halves' inputs are identical)
                return self._unet_rows(x, tuple(hs), halves, tw, h, w, t_dev, embeds, sub_idx_by_len[wl], vis_idx, motion_speed,
                                       unet_kwargs, cn_, same_input=True)

            def consume(wi, hs, eps):
                nonlocal slot
                wl = len(wins[wi])
                tw = n_cond + wl
                if not exchange:
                    for k, hf in enumerate(hs):
                        ops.window_scatter_add(eps[k * tw * hw:(k + 1) * tw * hw], idx_dev[wi], n_cond, 1, hf, eps_acc, counter, False)
                else:
                    for k, hf in enumerate(hs):
                        send[slot, :wl * hw].copy_(eps[(k * tw + n_cond) * hw:(k + 1) * tw * hw])
                        # one all-gather per exchange slot, issued as soon as the slot is filled: RCCL runs it on the process
                        # group's own stream, under the next group's forward; only the last slot's exchange is exposed
                        works.append(torch.distributed.all_gather_into_tensor(recv[slot].view(-1), send[slot].view(-1), group=group,
                                                                              async_op=True))
                        slot += 1

            lanes_ok = (self.odd_unit_lane and self.use_graphs and lat.is_cuda and controlnet is None and halves == 2 and
                        steps_done > 0 and hasattr(torch.cuda, "CUDAGraph"))
            gi = 0
            while gi < len(my_groups):
                wi, hs = my_groups[gi]
                nxt = my_groups[gi + 1] if gi + 1 < len(my_groups) else None
                if lanes_ok and nxt is not None and {len(hs), len(nxt[1])} == {1, 2}:
                    # a pair and a lone half next to each other: the lone forward on the lane stream, concurrently with the pair
                    lone, pair = ((wi, hs), nxt) if len(hs) == 1 else (nxt, (wi, hs))
                    main = torch.cuda.current_stream()
                    lane = self._lane_stream(dev)
                    lane.wait_stream(main)
                    with torch.cuda.stream(lane):
                        eps_lone = run_group(*lone)
                    eps_pair = run_group(*pair)
                    main.wait_stream(lane)
                    eps_lone.record_stream(main)
                    for g_, e_ in (((wi, hs), eps_lone if len(hs) == 1 else eps_pair), (nxt, eps_pair if len(hs) == 1 else eps_lone)):
                        consume(g_[0], g_[1], e_)
                    gi += 2
                    continue
                consume(wi, hs, run_group(wi, hs))
                gi += 1
            steps_done += 1
            if exchange:
                for k in range(slot, max_units):  # a rank with fewer units still takes part in every slot's collective
                    works.append(torch.distributed.all_gather_into_tensor(recv[k].view(-1), send[k].view(-1), group=group, async_op=True))
                ev = None
                if self.time_exchange and lat.is_cuda:  # bench.py --gpus N: the exposed part of the exchange, per step
                    ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                    ev[0].record()
                for wk in works:
                    wk.wait()  # stream dependency for RCCL (no host block)
                works.clear()
                ops.window_units_reduce(recv.view(max_units * world, win_len * hw, c), cover, eps_acc)  # table order -> bit-identical replicas
                if ev is not None:
                    ev[1].record()
                    self.exchange_events.append(ev)
            sched.loop_update(lat, eps_acc, counter, guidance[step], step, t)
            # the reference draws this (unused) tensor with the model output's dtype = the UNet's (:120-131), whatever the caller's latents are
            sched.consume_step_noise((1, c, T, h, w), self._unet_dtype(), dev, generator, noise_type, w_ind_noise)
            if callback is not None:
                callback(step, t, lat)

        out = lat.view(1, c, T, h, w)
        if cond is not None and reinsert_condition:
            if cond_slot is None:
                out = torch.cat([cond.view(1, c, n_cond, h, w), out], dim=2)
            else:  # batch_concat_two_tensor_with_index(condition_latents, vision_condition_latent_index, latents, latent_index) (:2149-2156)
                full = torch.zeros((1, c, n_cond + T, h, w), dtype=out.dtype, device=dev)
                full.index_copy_(2, torch.tensor(vis_idx, dtype=torch.long, device=dev), cond.view(1, c, n_cond, h, w))
                full.index_copy_(2, torch.tensor(latent_index, dtype=torch.long, device=dev), out)
                out = full
        return out

    def _unet_rows(self, x, hs: tuple, halves: int, tw: int, h: int, w: int, t_dev, embeds, sub_idx, vis_idx, motion_speed,
                   unet_kwargs: dict, cn=None, same_input: bool = False) -> torch.Tensor:
        """one UNet forward on window rows (preceded by the ControlNet forward on the same rows when ``cn`` is set);
        hipGraph-replayed when the call signature was captured before"""
        ehs = embeds[hs[0]:hs[-1] + 1] if len(hs) == 2 else embeds[hs[0]:hs[0] + 1]
        kw = {k: (self._slice_half(v, list(hs), halves) if k in _PER_HALF_KWARGS else v) for k, v in unet_kwargs.items()}

        def one(inp, nb, e, k, first_half=None, memo=None):
            first_half = hs[0] if first_half is None else first_half
            if memo is not None:
                k = dict(k, prefix_memo=memo)
            if cn is not None:
                net, ctrl, text_rep, scale, guess = cn
                # guess mode: the ControlNet sees only the text-conditioned half, the unconditional half gets no residuals
                # (:1218-1226, 1276-1288); this branch is only entered with one half per call
                if not (guess and halves == 2 and first_half == 0):
                    frames = inp.view(nb * tw, h, w, inp.shape[1]).permute(0, 3, 1, 2)   # rows -> (b t) c h w view (:1236-1238)
                    text = text_rep[first_half * tw:(first_half + nb) * tw]
                    down, mid = net(frames, t_dev, text, ctrl if nb == 1 else ctrl.repeat(nb, 1, 1, 1),
                                    conditioning_scale=scale, guess_mode=guess, return_dict=False)
                    k = dict(k, down_block_additional_residuals=down, mid_block_additional_residual=mid)
            return self.unet.forward_rows(inp, nb, tw, h, w, t_dev, e, sample_index=sub_idx,
                                          vision_conditon_frames_sample_index=vis_idx, sample_frame_rate=motion_speed, **k)

        split_halves = len(hs) == 2 and cn is not None and cn[4]  # guess mode needs the halves apart
        # identity of everything a forward of this call reads besides x
        sig = (hs, tw, h, w, float(motion_speed), t_dev.data_ptr(), embeds.data_ptr(), tuple(vis_idx or ()),
               tuple(sorted((k, _ident(v)) for k, v in kw.items())), tuple(x.shape),
               None if cn is None else (id(cn[0]), cn[1].data_ptr(), cn[2].data_ptr(), cn[3], cn[4]))

        def eager(inp):
            if split_halves and not (self.half_streams and inp.is_cuda):
                rows = inp.shape[0] // 2
                kws = [{k: (self._slice_half(v, [i], halves) if k in _PER_HALF_KWARGS else v) for k, v in unet_kwargs.items()} for i in hs]
                return torch.cat([one(inp[:rows], 1, embeds[hs[0]:hs[0] + 1], kws[0], hs[0]),
                                  one(inp[rows:], 1, embeds[hs[1]:hs[1] + 1], kws[1], hs[1])], dim=0)
            if not (self.half_streams and len(hs) == 2 and inp.is_cuda):
                return one(inp, len(hs), ehs, kw)
            # The CFG halves only meet in the loop glue: run them as two batch-1 forwards on two HIP streams, so the
            # prologue / epilogue / tail phases of one half's kernels overlap the other half's MFMA phases.
            rows = inp.shape[0] // 2
            main = torch.cuda.current_stream()
            side = self._side_stream(inp.device)
            kws = [{k: (self._slice_half(v, [i], halves) if k in _PER_HALF_KWARGS else v) for k, v in unet_kwargs.items()} for i in hs]
            # The forward builds shared caches lazily (packed conv / QKV / GEGLU weights, batched embedding projections, K/V
            # of the prompt): the FIRST forward of a signature -- and the first after any parameter / pack-epoch change --
            # runs both halves on ONE stream, so no stream ever reads a cache another stream is still filling.
            wkey = (sig, _pack_epoch(), _param_epoch(self.unet), None if cn is None else _param_epoch(cn[0]))
            # Shared CFG prefix (models/runtime.PrefixMemo): the loop hands both halves the same latents and timestep, so everything
            # in front of the first text cross-attention is computed by the first half's forward and replayed by the second's
            # (bit-identical to computing it twice).  Not with a ControlNet in the call (its residual inputs are per half).
            memo = PrefixMemo() if (self.share_cfg_prefix and cn is None and same_input) else None
            if wkey not in self._warm:
                if len(self._warm) >= 64:
                    self._warm.clear()
                e0 = one(inp[:rows], 1, embeds[hs[0]:hs[0] + 1], kws[0], hs[0], memo)
                e1 = one(inp[rows:], 1, embeds[hs[1]:hs[1] + 1], kws[1], hs[1], None if memo is None else memo.replay())
                self._warm[wkey] = True
                return torch.cat([e0, e1], dim=0)
            if memo is None:
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    e1 = one(inp[rows:], 1, embeds[hs[1]:hs[1] + 1], kws[1], hs[1])
                e0 = one(inp[:rows], 1, embeds[hs[0]:hs[0] + 1], kws[0], hs[0])
            else:
                # the first half runs on the main stream and marks the end of the shared prefix with an event; the side stream's
                # forward starts there, on the recorded tensors
                split_ev = torch.cuda.Event()
                memo.on_split = lambda: split_ev.record(main)
                e0 = one(inp[:rows], 1, embeds[hs[0]:hs[0] + 1], kws[0], hs[0], memo)
                if not memo.closed:
                    # a model that ignores `prefix_memo` (any object with the forward_rows interface): nothing was shared, the
                    # second half runs as a plain forward behind the first
                    side.wait_stream(main)
                    with torch.cuda.stream(side):
                        e1 = one(inp[rows:], 1, embeds[hs[1]:hs[1] + 1], kws[1], hs[1])
                else:
                    side.wait_event(split_ev)
                    with torch.cuda.stream(side):
                        for v in memo.store.values():
                            for t_ in (v if isinstance(v, (tuple, list)) else (v,)):
                                if torch.is_tensor(t_):
                                    t_.record_stream(side)
                        e1 = one(inp[rows:], 1, embeds[hs[1]:hs[1] + 1], kws[1], hs[1], memo.replay())
            main.wait_stream(side)
            e1.record_stream(main)
            return torch.cat([e0, e1], dim=0)

        if not (self.use_graphs and x.is_cuda and hasattr(torch.cuda, "CUDAGraph")):
            return eager(x)
        # a captured graph is only valid for the exact tensors it was recorded with: key on their identities, on the model's
        # parameter epoch and on the process-wide pack epoch (a .to() / load_state_dict of ANY HipModule re-packs weights)
        key = sig + (_pack_epoch(), _param_epoch(self.unet), None if cn is None else _param_epoch(cn[0]))
        gf = self._graphs.pop(key, None)
        if gf is None:
            while len(self._graphs) >= 8:  # least recently used first: stale captures pin their activation pools
                self._graphs.pop(next(iter(self._graphs)))
            gf = _GraphedForward(eager, x)  # its eager warm-up call is the single-stream cache fill (see eager())
            self._graphs[key] = gf
            return gf.first_result
        self._graphs[key] = gf  # re-insert: dict order = recency
        return gf(x)

    def graph_replays(self) -> int:
        """hipGraph replays issued so far by this object's captured forwards (bench.py asserts the timed steps were replays)"""
        return sum(g.replays for g in self._graphs.values())

    # The side / lane streams are PROCESS-WIDE, one of each per device: HIP maps streams onto a handful of hardware queues
    # round-robin (4 by default), so a process that creates a fresh stream per denoiser object ends up with a "side" stream that
    # shares its queue with the main stream -- the CFG halves then run one after the other, silently (measured, round 6: a
    # two-stream replay on the 4th stream created in the process took the serial 49.4 ms instead of 38.8).
    _shared_streams: Dict[tuple, "torch.cuda.Stream"] = {}

    @classmethod
    def _shared_stream(cls, kind: str, dev):
        key = (kind, str(dev))
        st = cls._shared_streams.get(key)
        if st is None:
            st = cls._shared_streams[key] = torch.cuda.Stream(device=dev)
        return st

    def _lane_stream(self, dev):
        return self._shared_stream("lane", dev)

    def _side_stream(self, dev):
        return self._shared_stream("side", dev)

    @staticmethod
    def _slice_half(v, hs: List[int], halves: int):
        """conditioning tensors batched over the CFG halves are sliced to the owned halves: [uncond, cond] on dim 0, or -- the
        ControlNet residuals, [(b t), C, h, w] (unet_3d_condition.py:1146-1156) -- b-major blocks of rows on dim 0"""
        if v is None or halves == 1 or len(hs) == halves:
            return v
        if torch.is_tensor(v):
            per = v.shape[0] // halves
            if per * halves != v.shape[0]:
                raise ValueError(f"a per-half tensor of {v.shape[0]} rows cannot be split into {halves} CFG halves")
            return v[hs[0] * per:(hs[0] + 1) * per]
        return [ParallelDenoiser._slice_half(e, hs, halves) for e in v]


_PER_HALF_KWARGS = ("down_block_refer_embs", "mid_block_refer_emb", "vision_clip_emb", "down_block_additional_residuals",
                    "mid_block_additional_residual", "pose_guider_emb", "ip_adapter_face_emb", "refer_self_attn_emb")


def _ident(v) -> tuple:
    if torch.is_tensor(v):
        return (v.data_ptr(), tuple(v.shape), v._version)
    if isinstance(v, (list, tuple)):
        return tuple(_ident(e) for e in v)
    return (repr(v),)


def _first_param_dtype(module) -> torch.dtype:
    params = getattr(module, "parameters", None)
    if callable(params):
        for prm in params():
            return prm.dtype
    return torch.float16  # (the UNet of this package computes in fp16)


def _pack_epoch() -> int:
    from ..models.layers import _PACK_EPOCH
    return _PACK_EPOCH[0]


def _param_epoch(unet) -> int:
    """changes when any parameter of the model is modified in place (LoRA merge, new checkpoint): captured graphs hold
    pointers into the packed weight copies, which are rebuilt on such a change"""
    fn = getattr(unet, "param_epoch", None)
    return int(fn()) if callable(fn) else 0


class _GraphedForward:
    """Warm-up (eager: fills the packed-weight / conditioning caches and the allocator), then capture of fn(static_in)
    into a hipGraph through torch.cuda.CUDAGraph.  Every kernel of libmusev_hip is launched on torch's current stream,
    which is the capture stream inside torch.cuda.graph()."""

    def __init__(self, fn, example: torch.Tensor):
        self.static_in = example.clone()
        self.first_result = fn(self.static_in)  # eager warm-up; also the result of this first call
        self.graph = None
        self.static_out = None
        self.fn = fn
        self.replays = 0
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        try:
            # thread_local: other host threads (the RCCL watchdog of torch.distributed polls events) must not abort a capture
            with torch.cuda.graph(g, capture_error_mode="thread_local"):
                self.static_out = fn(self.static_in)
        except Exception as ex:
            # a failed capture is an ERROR, not a silent switch to eager launches: the timed path is the graph replay, and a run
            # that quietly lost it would report numbers of a different code path.  MUSEV_NO_GRAPH=1 (or use_graphs=False) is the
            # explicit way to run eager.
            torch.cuda.synchronize()
            raise RuntimeError(f"musev_amd: hipGraph capture of the window forward failed ({ex!r}); set MUSEV_NO_GRAPH=1 to run "
                               "eager launches instead") from ex
        self.graph = g

    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        self.static_in.copy_(x)
        self.graph.replay()
        self.replays += 1
        return self.static_out
