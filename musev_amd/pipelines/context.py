"""Window ("context") schedules of the parallel denoise loop.

The public names, keyword arguments and results are those of the reference's musev/pipelines/context.py (:12-149) so that
callers can switch imports; the implementation is a small table-driven restatement: a schedule is a function
``(n_frames, window, overlap, ...) -> list of frame-index lists`` registered in ``_SCHEDULES``, evaluated once per call of
the loop (the reference evaluates it at step 0 only, :132) and cut into batches.  Host-side integer logic only."""
from __future__ import annotations

from math import ceil, log2
from typing import Callable, Dict, Iterator, List, Optional

__all__ = ["ordered_halving", "uniform", "uniform_v2", "generate_sample_idxs", "get_context_scheduler", "get_total_steps",
           "drop_last_repeat_context", "prepare_global_context"]

Windows = List[List[int]]


def ordered_halving(val: int) -> float:
    """van-der-Corput style fraction: the 64 binary digits of ``val`` mirrored behind the binary point (:12-17)"""
    bits = format(val & ((1 << 64) - 1), "064b")
    return int(bits[::-1], 2) / 2.0 ** 64


def _strided_windows(n_frames: int, window: int, overlap: int, max_levels: int, phase: float, wrap_tail: bool) -> Iterator[List[int]]:
    # level l samples every 2**l-th frame; a level's windows start every (window * 2**l - overlap) frames from a phase offset
    # and wrap around the clip (:33-48)
    levels = min(max_levels, ceil(log2(n_frames / window)) + 1)
    shift = int(round(n_frames * phase))
    for lvl in range(levels):
        hop = 1 << lvl
        first = int(phase * hop) + shift
        last = n_frames + shift - (0 if wrap_tail else overlap)
        for s in range(first, last, window * hop - overlap):
            yield [(s + k * hop) % n_frames for k in range(window)]


def uniform(step: int = ..., num_steps: Optional[int] = None, num_frames: int = ..., context_size: Optional[int] = None,
            context_stride: int = 3, context_overlap: int = 4, closed_loop: bool = True) -> Iterator[List[int]]:
    """the AnimateDiff-style schedule: one window if the clip fits, else strided wrapping windows (:21-48)"""
    if num_frames <= context_size:
        return iter([list(range(num_frames))])
    return _strided_windows(num_frames, context_size, context_overlap, context_stride, ordered_halving(step), closed_loop)


def generate_sample_idxs(total: int, window_size: int, step: int, sample_rate: int = 1, drop_last: bool = False) -> Windows:
    """stand-in for mmcm.utils.itertools_util.generate_sample_idxs (un-vendored; called at :60-66): windows
    range(s, min(s + window_size * sample_rate, total), sample_rate) for s = 0, step, 2 step, ...; a short last window
    is kept unless ``drop_last``"""
    found: Windows = []
    for s in range(0, total, step):
        frames = list(range(s, min(s + window_size * sample_rate, total), sample_rate))
        if drop_last and len(frames) < window_size:
            break
        found.append(frames)
    return found


def uniform_v2(step: int = ..., num_steps: Optional[int] = None, num_frames: int = ..., context_size: Optional[int] = None,
               context_stride: int = 3, context_overlap: int = 4, closed_loop: bool = True) -> Windows:
    """non-wrapping consecutive windows (:51-66)"""
    return generate_sample_idxs(num_frames, context_size, context_size - context_overlap)


_SCHEDULES: Dict[str, Callable] = {"uniform": uniform, "uniform_v2": uniform_v2}


def get_context_scheduler(name: str) -> Callable:
    if name not in _SCHEDULES:
        raise ValueError(f"Unknown context_overlap policy {name}")
    return _SCHEDULES[name]


def get_total_steps(scheduler, timesteps: List[int], num_steps: Optional[int] = None, num_frames: int = ...,
                    context_size: Optional[int] = None, context_stride: int = 3, context_overlap: int = 4,
                    closed_loop: bool = True) -> int:
    """number of window forwards of a whole run when the schedule is re-evaluated at every step (:85-102)"""
    total = 0
    for i, _ in enumerate(timesteps):
        total += sum(1 for _ in scheduler(i, num_steps, num_frames, context_size, context_stride, context_overlap))
    return total


def drop_last_repeat_context(contexts: Windows) -> Windows:
    """a trailing window that ends on the same frame as the one before it adds nothing (:105-117)"""
    redundant = len(contexts) >= 2 and contexts[-1][-1] == contexts[-2][-1]
    return contexts[:-1] if redundant else contexts


def prepare_global_context(context_schedule: str, num_inference_steps: int, time_size: int, context_frames: int,
                           context_stride: int, context_overlap: int, context_batch_size: int) -> List[Windows]:
    """the loop's window list, evaluated once (step 0) and cut into batches of ``context_batch_size`` (:120-149)"""
    make = get_context_scheduler(context_schedule)
    wins = drop_last_repeat_context(list(make(step=0, num_steps=num_inference_steps, num_frames=time_size, context_size=context_frames,
                                              context_stride=context_stride, context_overlap=context_overlap)))
    return [wins[i:i + context_batch_size] for i in range(0, len(wins), context_batch_size)]
