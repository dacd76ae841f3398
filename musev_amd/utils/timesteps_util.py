"""Per-step parameter schedules of the denoise loop -- the guidance-scale schedule of MusevControlNetPipeline.__call__
(reference musev/utils/timesteps_util.py:5-61; used at pipeline_controlnet.py:1718-1723, consumed per step at :2103).
Host-side, no tensors: the value of step i is passed to the fused CFG + scheduler kernel as a scalar argument."""
from __future__ import annotations

from typing import Callable, Dict, List

import numpy as np

__all__ = ["generate_parameters_with_timesteps"]


def _linear(start, stop, num, n_fix_start):
    return list(np.linspace(start=start, stop=stop, num=num))          # numpy floats, as the reference returns them


def _two_stage(start, stop, num, n_fix_start):
    head = num // 2
    return [start] * head + [stop] * (num - head)


def _fix_two_stage(start, stop, num, n_fix_start):
    return [start] * n_fix_start + [stop] * (num - n_fix_start)


def _three_stage(start, stop, num, n_fix_start):
    third = num // 3
    return [start] * third + [(start + stop) // 2] * third + [stop] * (num - 2 * third)   # integer midpoint, as upstream


_METHODS: Dict[str, Callable] = {"linear": _linear, "two_stage": _two_stage, "three_stage": _three_stage,
                                 "fix_two_stage": _fix_two_stage}


def generate_parameters_with_timesteps(start, num: int, stop=None, method: str = "linear", n_fix_start: int = 3) -> List[float]:
    """``num`` values going from ``start`` to ``stop``: constant when ``stop`` is None or equal to ``start``; otherwise
    "linear" (linspace), "two_stage" (first half start, rest stop), "three_stage" (thirds: start, floor-midpoint, stop) or
    "fix_two_stage" (``n_fix_start`` steps of start, rest stop).  Unknown methods raise ValueError."""
    if stop is None or start == stop:
        return [start] * num
    try:
        fn = _METHODS[method]
    except KeyError:
        raise ValueError(f"now only support linear, two_stage, three_stage, fix_two_stage, but given {method}") from None
    return fn(start, stop, num, n_fix_start)
