"""musev_amd: the MI355X-native MuseV parallel-denoise hot path.

Importing the package sets ONE process default, and only if the user has not set it: GPU_MAX_HW_QUEUES=8.  The HIP runtime maps a
process's streams onto that many hardware queues round-robin (4 by default) and two streams that share a queue run one after the other,
silently.  The denoise step runs the CFG halves on two streams, a rank of an 8-GPU config-4 run adds the lone half's "lane" stream, and a
captured graph's parallel branch and RCCL bring streams of their own: with 4 queues the lane stream landed on an occupied queue and
bought nothing (round 6, tools/gpu_odd_unit_lane.py: 83.4 -> 83.7 ms per rank-step), with 8 it is worth -13 % (80.4 -> 69.6 ms;
profiles/r06y_odd_unit_lane_q8.log); the two-stream step itself is indifferent (profiles/r06y_ab_hwq.log).  The runtime reads the
variable when it initialises, i.e. at the process's first HIP call: import musev_amd (or set the variable) before touching the GPU."""
import os as _os

_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
