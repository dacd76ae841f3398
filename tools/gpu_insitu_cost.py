#!/usr/bin/env python3
"""What a class of launches costs WHERE IT RUNS (two HIP streams, hipGraph replay) -- not its isolated duration:

    python tools/gpu_insitu_cost.py gpurun_out/<tag>_insitu_cost.json [--rounds 2] [--workload config2]

Every C-ABI entry of the list below is, in its own child process, issued TWICE per call (same arguments; the results stay valid:
every entry is idempotent except an accumulating attention segment, whose second issue only moves values), the denoise step is
timed exactly as bench.py times it (the driver's step count, graphs on), and the difference to the unmodified step -- measured in the
same call, interleaved -- is the marginal cost of that entry's launches in the step.  rocprofv3's per-kernel durations are the
kernels' own (the profiler serialises the two streams); this table says how much of each the second stream hides.

    child:  python tools/gpu_insitu_cost.py --child <entry[:class]> [bench args]      (prints bench.py's JSON line)
"""
from __future__ import annotations

import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

ENTRIES = [
    "mv_gemm_f16:linear", "mv_gemm_f16:conv3x3", "mv_gemm_f16:tconv3", "mv_ffn_geglu_f16", "mv_temporal_attn_block_f16",
    "mv_attention_f16:d40self", "mv_attention_f16:short", "mv_attention_f16:rest", "mv_temporal_attention_f16",
    "mv_groupnorm_cs_f16", "mv_groupnorm_cs_fold_linear_f16", "mv_groupnorm_f16", "mv_layernorm_f16", "mv_add_f16", "mv_xattn_block_f16",
]
BENCH_ARGS = ["--steps", "20", "--warmup", "5", "--no-cpu-baseline", "--no-roofline", "--no-config4"]


def _selector(cls: str):
    """which calls of the entry are doubled: by the descriptor behind the first argument (ctypes byref keeps it as ._obj)"""
    if cls in ("linear", "conv3x3", "tconv3"):
        mode = {"linear": 0, "conv3x3": 1, "tconv3": 2}[cls]
        return lambda a: a[0]._obj.mode == mode
    if cls == "d40self":
        return lambda a: a[0]._obj.d == 40 and a[0]._obj.seg[0].len > 1024
    if cls == "short":   # <= 1024 keys in the first segment: the text cross-attention of every level + the self-attention of levels 1-3
        return lambda a: a[0]._obj.seg[0].len <= 1024 and a[0]._obj.lq > 64
    if cls == "rest":
        return lambda a: not (a[0]._obj.d == 40 and a[0]._obj.seg[0].len > 1024) and not (a[0]._obj.seg[0].len <= 1024 and a[0]._obj.lq > 64)
    return lambda a: True


def child(entry: str, bench_args):
    from musev_amd import _lib
    lib = _lib.load()
    if entry != "none":
        name, _, cls = entry.partition(":")
        orig, sel = getattr(lib, name), _selector(cls)

        def twice(*a):
            rc = orig(*a)
            return orig(*a) if (rc == 0 and sel(a)) else rc

        setattr(lib, name, twice)   # (an instance attribute of the CDLL: musev_amd.ops resolves the entry through it)
    import bench
    sys.argv = ["bench.py"] + list(bench_args)
    bench.main()


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "--child":
        return child(sys.argv[2], sys.argv[3:])
    out = sys.argv[1]
    rounds = int(sys.argv[sys.argv.index("--rounds") + 1]) if "--rounds" in sys.argv else 2
    extra = ["--workload", sys.argv[sys.argv.index("--workload") + 1]] if "--workload" in sys.argv else []

    def run(entry):
        p = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", entry] + BENCH_ARGS + extra, capture_output=True, text=True, timeout=600)
        lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
        if p.returncode != 0 or not lines:
            return None
        return json.loads(lines[-1])["ms_per_step"]

    res = {"base": [], "entries": {e: [] for e in ENTRIES}}
    for _ in range(rounds):
        res["base"].append(run("none"))
        for i, e in enumerate(ENTRIES):
            res["entries"][e].append(run(e))
            if i % 4 == 3:
                res["base"].append(run("none"))
    base = [b for b in res["base"] if b]
    b0 = sum(base) / len(base)
    print(f"unmodified step: {b0:.3f} ms (min {min(base):.3f}, max {max(base):.3f}, {len(base)} legs)")
    tot = 0.0
    table = {}
    for e in ENTRIES:
        v = [x for x in res["entries"][e] if x]
        if not v:
            print(f"{e:36s} failed")
            continue
        c = sum(v) / len(v) - b0
        tot += c
        table[e] = c
        print(f"{e:36s} +{c:6.3f} ms in the step   ({', '.join(f'{x:.3f}' for x in v)})")
    print(f"sum of the marginal costs: {tot:.3f} ms of {b0:.3f}")
    with open(out, "w") as f:
        json.dump({"base_ms": base, "legs": res["entries"], "marginal_ms": table, "sum_ms": tot}, f, indent=1)


if __name__ == "__main__":
    main()
