"""Test doubles for the loop-glue kernels (torch on CPU) and a closed-form fake UNet, so that the ORCHESTRATION of
musev_amd.pipelines.parallel_denoise (unit sharding, all-gather layout, fixed-order accumulation, DDIM bookkeeping) can
be exercised under gloo without a GPU.  Test infrastructure only: the product path never imports this module."""
from __future__ import annotations

import math

import torch


def window_gather(latents, cond, idx, n_cond, copies, hi_lo=False, cond_slot=None):
    frames = latents[:, idx.long()]
    if n_cond and cond_slot is None:
        frames = torch.cat([cond, frames], dim=1)
    elif n_cond:  # condition frames at their slots first, the window's frames at n_cond.. afterwards (data_util.py:242-268)
        full = torch.zeros((latents.shape[0], n_cond + idx.numel(), latents.shape[2]), dtype=latents.dtype)
        full.index_copy_(1, cond_slot.long(), cond)
        full[:, n_cond:] = frames
        frames = full
    rows = frames.permute(1, 2, 0).reshape(-1, latents.shape[0])
    rows = torch.cat([rows] * copies, dim=0)
    hi = rows.to(torch.float16)
    if hi_lo:  # rows of 2 C columns [hi | lo] (mv_window_gather(hi_lo = 1))
        return torch.cat([hi, (rows.float() - hi.float()).to(torch.float16)], dim=1)
    return hi


def window_scatter_add(eps_win, idx, n_cond, halves, half_offset, eps_acc, counter, add_counter):
    _, c, t_total, hw = eps_acc.shape
    win = idx.numel()
    e = eps_win.float().reshape(halves, n_cond + win, hw, c)[:, n_cond:].permute(0, 3, 1, 2)
    eps_acc[half_offset:half_offset + halves, :, idx.long()] += e
    if add_counter:
        counter[idx.long()] += 1


def window_units_reduce(units, table, eps_acc):
    halves, c, t_total, hw = eps_acc.shape
    for h in range(halves):
        for f in range(t_total):
            acc = torch.zeros(hw, c)
            for slot, j in table[h, f].tolist():
                if slot < 0:
                    break
                acc = acc + units[slot, j * hw:(j + 1) * hw]
            eps_acc[h, :, f] = acc.t()


def cfg_ddim_step(latents, eps_acc, counter, guidance, a_t, a_prev):
    eps = eps_acc / counter[None, None, :, None]
    e = eps[0] + guidance * (eps[1] - eps[0]) if eps.shape[0] == 2 else eps[0]
    x0 = (latents - math.sqrt(1 - a_t) * e) / math.sqrt(a_t)
    latents.copy_(math.sqrt(a_prev) * x0 + math.sqrt(1 - a_prev) * e)


def cfg_affine_step(latents, eps_acc, counter, guidance, cx, ce):
    eps = eps_acc / counter[None, None, :, None]
    e = eps[0] + guidance * (eps[1] - eps[0]) if eps.shape[0] == 2 else eps[0]
    latents.copy_(cx * latents + ce * e)


class FakeUNet:
    """eps = tanh(0.5 x) * (1 + 0.1 * mean(text)) + 0.01 * (timestep / 1000) + 0.05 * frame_position + 0.02 * [the frame is one the
    caller names a vision-condition frame] -- depends on the input frames, the CFG half's prompt, the timestep, the window-local
    frame position and the condition-frame index list, like the real network."""

    in_channels = 4

    def forward_rows(self, x, b, t, h, w, timestep, ehs, **kw):
        if x.shape[1] == 2 * self.in_channels:
            x = x[:, :self.in_channels]  # [hi | lo] rows: this double works on the fp16 half, like `nchw` below
        c = x.shape[1]
        v = x.float().reshape(b, t, h * w, c)
        s = 1.0 + 0.1 * ehs.float().mean(dim=(1, 2)).reshape(b, 1, 1, 1)
        pos = torch.arange(t, dtype=torch.float32).reshape(1, t, 1, 1)
        out = torch.tanh(0.5 * v) * s + 0.01 * float(timestep.reshape(-1)[0]) / 1000.0 + 0.05 * pos
        vis = kw.get("vision_conditon_frames_sample_index")
        if vis is not None:
            mark = torch.zeros(t)
            mark[[int(i) for i in (vis.tolist() if torch.is_tensor(vis) else vis)]] = 1.0
            out = out + 0.02 * mark.reshape(1, t, 1, 1)
        return out.reshape(b * t * h * w, c)  # fp32 rows, like UNet3DConditionModel.forward_rows

    def nchw(self, x, t, ehs, **kw):
        """the same function on the reference layout [b, c, t, h, w] (for the oracle loop)"""
        b, c, tt, h, w = x.shape
        rows = x.permute(0, 2, 3, 4, 1).reshape(-1, c).to(torch.float16)
        y = self.forward_rows(rows, b, tt, h, w, torch.as_tensor(float(t)), ehs, **kw)
        return y.float().reshape(b, tt, h, w, c).permute(0, 4, 1, 2, 3)
