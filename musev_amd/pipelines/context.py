"""Sliding-window ("context") schedulers of the parallel denoise loop -- same names, arguments and results as the
reference's musev/pipelines/context.py (:12-149).  Pure host-side integer logic."""
from __future__ import annotations

import math
from typing import Callable, Iterator, List, Optional


def ordered_halving(val: int) -> float:
    """fraction whose binary digits are the 64-bit reversal of ``val`` (context.py:12-17)"""
    rev = 0
    for _ in range(64):
        rev = (rev << 1) | (val & 1)
        val >>= 1
    return rev / float(1 << 64)


def uniform(step: int = ..., num_steps: Optional[int] = None, num_frames: int = ..., context_size: Optional[int] = None,
            context_stride: int = 3, context_overlap: int = 4, closed_loop: bool = True) -> Iterator[List[int]]:
    """Windows of ``context_size`` frames at strides 1, 2, 4, ... starting every (size*stride - overlap) frames;
    indices wrap modulo ``num_frames`` (context.py:21-48)."""
    if num_frames <= context_size:
        yield list(range(num_frames))
        return
    n_strides = min(context_stride, int(math.ceil(math.log2(num_frames / context_size))) + 1)
    frac = ordered_halving(step)
    for level in range(n_strides):
        cstep = 1 << level
        pad = int(round(num_frames * frac))
        start = int(frac * cstep) + pad
        stop = num_frames + pad + (0 if closed_loop else -context_overlap)
        for j in range(start, stop, context_size * cstep - context_overlap):
            yield [e % num_frames for e in range(j, j + context_size * cstep, cstep)]


def generate_sample_idxs(total: int, window_size: int, step: int, sample_rate: int = 1, drop_last: bool = False) -> List[List[int]]:
    """stand-in for mmcm.utils.itertools_util.generate_sample_idxs (un-vendored dependency of context.py:60-66):
    consecutive windows range(s, min(s + window, total)) for s = 0, step, 2*step, ..."""
    out, s = [], 0
    while s < total:
        idx = list(range(s, min(s + window_size * sample_rate, total), sample_rate))
        if len(idx) < window_size and drop_last:
            break
        out.append(idx)
        s += step
    return out


def uniform_v2(step: int = ..., num_steps: Optional[int] = None, num_frames: int = ..., context_size: Optional[int] = None,
               context_stride: int = 3, context_overlap: int = 4, closed_loop: bool = True) -> List[List[int]]:
    return generate_sample_idxs(total=num_frames, window_size=context_size, step=context_size - context_overlap,
                                sample_rate=1, drop_last=False)


def get_context_scheduler(name: str) -> Callable:
    if name == "uniform":
        return uniform
    if name == "uniform_v2":
        return uniform_v2
    raise ValueError(f"Unknown context_overlap policy {name}")


def get_total_steps(scheduler, timesteps: List[int], num_steps: Optional[int] = None, num_frames: int = ...,
                    context_size: Optional[int] = None, context_stride: int = 3, context_overlap: int = 4,
                    closed_loop: bool = True) -> int:
    return sum(len(list(scheduler(i, num_steps, num_frames, context_size, context_stride, context_overlap)))
               for i in range(len(timesteps)))


def drop_last_repeat_context(contexts: List[List[int]]) -> List[List[int]]:
    """drop a trailing window that ends on the same frame as its predecessor (context.py:105-117)"""
    if len(contexts) >= 2 and contexts[-1][-1] == contexts[-2][-1]:
        return contexts[:-1]
    return contexts


def prepare_global_context(context_schedule: str, num_inference_steps: int, time_size: int, context_frames: int,
                           context_stride: int, context_overlap: int, context_batch_size: int) -> List[List[List[int]]]:
    """window list for the whole denoise loop, evaluated once at step 0 and grouped into batches (context.py:120-149)"""
    sched = get_context_scheduler(context_schedule)
    queue = list(sched(step=0, num_steps=num_inference_steps, num_frames=time_size, context_size=context_frames,
                       context_stride=context_stride, context_overlap=context_overlap))
    queue = drop_last_repeat_context(queue)
    nb = math.ceil(len(queue) / context_batch_size)
    return [queue[i * context_batch_size:(i + 1) * context_batch_size] for i in range(nb)]
