#!/usr/bin/env python3
"""A timeline of the graph-replayed two-stream denoise step, taken on the device (run on the MI355X):

    python tools/gpu_timeline.py gpurun_out/<tag>_timeline.json [bench args]

rocprofv3 serialises the two HIP streams under tracing (profiles/r06b_trace_overlap.json), so its per-kernel durations say nothing about
what runs BESIDE what in the step.  Here every C-ABI launch of the step is followed, on its own stream, by a one-lane kernel that stores the
100 MHz wall clock into a slot (tools/timeline_ts.hip); the stamps are captured into the hipGraph with the launches, and the last replay
of bench.py's timed region leaves one time per launch: its END on the device (its start is the previous stamp of the same stream: the
streams are in-order).  The stamps cost 0.8 us per launch in the step (+ 1.0 ms on the config-2 step, 2 %): the table is a picture of
the schedule, not a timing of the product.

Output: per stream the launches of the last replay in order (entry, class, end time, duration in the step), and a summary per class:
time in the step against the sum of the partner stream's classes that ran beside it."""
from __future__ import annotations

import ctypes as C
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def launch_entries():
    """the C-ABI entries that launch on a stream: their last parameter is `void* stream` (include/musev_hip.h)"""
    text = open(os.path.join(ROOT, "include", "musev_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(m.group(1) for m in re.finditer(r"\bint\s+(mv_\w+)\s*\(([^;]*?)void\*\s*stream\s*\)\s*;", text, flags=re.S)))


def classify(name, a):
    try:
        if name == "mv_gemm_f16":
            d = a[0]._obj
            return f"gemm:{('linear', 'conv3x3', 'tconv3')[d.mode]}", f"M{d.M} N{d.N} K{d.K}{' geglu' if d.geglu else ''}"
        if name == "mv_attention_f16":
            d = a[0]._obj
            kind = "attn:self_l0" if (d.d == 40 and d.seg[0].len > 1024) else "attn:other"
            return kind, f"d{d.d} nb{d.nb} lq{d.lq} nseg{d.nseg} len0 {d.seg[0].len}"
        if name in ("mv_ffn_geglu_f16", "mv_temporal_attn_block_f16", "mv_xattn_block_f16"):
            return {"mv_ffn_geglu_f16": "ffn_fused", "mv_temporal_attn_block_f16": "tsa_fused", "mv_xattn_block_f16": "xab_fused"}[name], ""
        if name.startswith("mv_groupnorm"):
            return "groupnorm", name[len("mv_groupnorm"):]
        if name == "mv_layernorm_f16":
            return "layernorm", ""
        if name == "mv_temporal_attention_f16":
            return "attn:temporal", ""
    except Exception:
        pass
    return "other", name


def main():
    out_path = sys.argv[1]
    bench_args = sys.argv[2:] or ["--steps", "6", "--warmup", "3", "--no-cpu-baseline", "--no-roofline", "--no-config4"]
    import torch
    from musev_amd import _lib
    lib = _lib.load()
    so = os.path.join(ROOT, "tools", "scratch", "libmvts.so")
    if not os.path.exists(so):   # the stamp kernel (tool code, not part of libmusev_hip): built on first use
        import subprocess
        os.makedirs(os.path.dirname(so), exist_ok=True)
        subprocess.run([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", "-o", so,
                        os.path.join(ROOT, "tools", "timeline_ts.hip")], check=True)
    ts = C.CDLL(so)
    ts.mvts_record.restype = C.c_int
    ts.mvts_record.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    nslots = 1 << 20
    slots = torch.zeros(nslots, dtype=torch.int64, device="cuda")
    log = []   # (slot, entry, class, detail, stream)

    def wrap(name, orig):
        def f(*a):
            rc = orig(*a)
            if rc == 0 and len(log) < nslots:
                st = a[-1]
                st = st.value if hasattr(st, "value") else st
                cls, det = classify(name, a)
                ts.mvts_record(slots.data_ptr(), len(log), st)
                log.append((len(log), name, cls, det, int(st or 0)))
            return rc
        return f

    for name in launch_entries():
        setattr(lib, name, wrap(name, getattr(lib, name)))
    import bench
    sys.argv = ["bench.py"] + list(bench_args)
    bench.main()
    torch.cuda.synchronize()
    t = slots.cpu().tolist()
    tmax = max(t)
    # the last replay: the stamps of the final ~ 1.3 x step window (100 MHz ticks) -- the captured launches' slots were overwritten by it;
    # eager launches of that step (the loop glue) are in it too
    recent = [(i, t[i]) for i in range(len(log)) if t[i] > 0]
    recent.sort(key=lambda p: p[1])
    # window: walk back from the end while the gap between consecutive stamps stays below 2 ms
    cut = len(recent) - 1
    while cut > 0 and recent[cut][1] - recent[cut - 1][1] < 200000:
        cut -= 1
    # (the stamps of earlier replays were overwritten, so everything behind the last big gap is the last step + the few eager stamps around it)
    last = recent[cut:]
    # keep one step: the stamps within 80 ms of the end
    last = [p for p in last if tmax - p[1] < 8000000]
    t0 = min(p[1] for p in last)
    streams = {}
    for i, tick in last:
        _, name, cls, det, st = log[i]
        streams.setdefault(st, []).append(dict(slot=i, entry=name, cls=cls, detail=det, end_us=(tick - t0) / 100.0))
    for st, ev in streams.items():
        ev.sort(key=lambda e: e["end_us"])
        prev = None
        for e in ev:
            e["dur_us"] = None if prev is None else e["end_us"] - prev
            prev = e["end_us"]
    # per class: time in the step; and what the OTHER streams were running during it
    big = sorted(streams, key=lambda s: -len(streams[s]))
    summary = {}
    for st in big:
        for e in streams[st]:
            if e["dur_us"] is None:
                continue
            s = summary.setdefault(e["cls"], dict(launches=0, us=0.0, beside={}))
            s["launches"] += 1
            s["us"] += e["dur_us"]
            a0, a1 = e["end_us"] - e["dur_us"], e["end_us"]
            for so in big:
                if so == st:
                    continue
                for o in streams[so]:
                    if o["dur_us"] is None:
                        continue
                    b0, b1 = o["end_us"] - o["dur_us"], o["end_us"]
                    ov = min(a1, b1) - max(a0, b0)
                    if ov > 0:
                        s["beside"][o["cls"]] = s["beside"].get(o["cls"], 0.0) + ov
    span = max(e["end_us"] for ev in streams.values() for e in ev)
    res = dict(span_us=span, streams={str(k): v for k, v in streams.items()}, summary=summary, note=__doc__.split("\n\n")[1])
    with open(out_path, "w") as f:
        json.dump(res, f)
    print(f"timeline: {sum(len(v) for v in streams.values())} stamps on {len(streams)} streams over {span / 1e3:.2f} ms")
    for st in big:
        ev = streams[st]
        print(f"  stream {st:#x}: {len(ev)} launches, {ev[0]['end_us'] / 1e3:.2f} .. {ev[-1]['end_us'] / 1e3:.2f} ms")
    print(f"{'class':<16} {'launches':>8} {'ms in step':>11}   beside (ms of the other streams' classes running at the same time)")
    for cls, s in sorted(summary.items(), key=lambda kv: -kv[1]["us"]):
        bes = ", ".join(f"{k} {v / 1e3:.2f}" for k, v in sorted(s["beside"].items(), key=lambda kv: -kv[1])[:6])
        print(f"{cls:<16} {s['launches']:>8} {s['us'] / 1e3:>11.2f}   {bes}")


if __name__ == "__main__":
    main()
