#!/usr/bin/env python3
"""Static report of the gfx950 code the compiler emits for the hot kernels (no GPU needed: hipcc cross-compiles):

    python tools/isa_report.py [--out profiles/<tag>_isa_report.json] [--filter xattn]

Per kernel: VGPRs (arch + accumulator), SGPRs, scratch bytes (spills), static LDS, the waves per SIMD the register budget allows, and
the instruction mix of its HOTTEST LOOP (among the backward branches whose body holds matrix instructions, the one with the highest
MFMA density -- the K loop / key-tile loop, not the outer loop that also spans the epilogue; kernels without MFMAs: the longest
loop): MFMA, transcendental (v_exp / v_rcp / ...), conversions, other VALU, LDS, global / buffer memory, scratch (spill traffic INSIDE
that loop), scalar, waits.  The mix counts every instruction in the loop's address range, i.e. also blocks the common path branches
around (attn3: the rescale and the partial-tile masking): an upper bound per iteration.  What it is for: the counters of
profiles/*_sq_counters.log say which pipe is busy; this says why (e.g. attn3<40>: VALU / MFMA instruction ratio), and it is the only
evidence available for a kernel form written while no GPU is at hand (spills, register budget, a loop the compiler bloated)."""
from __future__ import annotations

import argparse
import collections
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "musev_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=fast", "-S", "--cuda-device-only"]   # csrc/build.sh
SOURCES = ("gemm", "attention", "norm", "ffn", "elementwise")
FILT = next((p for p in ("/opt/rocm/lib/llvm/bin/llvm-cxxfilt", "/usr/bin/c++filt") if os.path.exists(p)), "")


def classify(op: str) -> str:
    if op.startswith("v_mfma") or op.startswith("v_smfmac"):
        return "mfma"
    if op.startswith(("v_exp", "v_log", "v_rcp", "v_rsq", "v_sqrt", "v_sin", "v_cos")):
        return "transcendental"
    if op.startswith("v_cvt"):
        return "convert"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith("scratch_"):
        return "scratch"
    if op.startswith(("global_", "buffer_", "flat_")):
        return "vmem"
    if op.startswith("s_waitcnt") or op.startswith("s_barrier"):
        return "wait"
    if op.startswith("s_"):
        return "salu"
    return "other"


def mix(lines):
    c = collections.Counter()
    for ln in lines:
        ln = ln.strip()
        if not ln or ln[0] in ";." or ln.endswith(":"):
            continue
        c[classify(ln.split()[0])] += 1
    return dict(c)


def hottest_loop(body):
    labels = {}
    for i, ln in enumerate(body):
        m = re.match(r"^(\.LBB\d+_\d+):", ln)
        if m:
            labels[m.group(1)] = i
    best = None
    for i, ln in enumerate(body):
        m = re.match(r"\s*s_cbranch_\w+ (\.LBB\d+_\d+)", ln) or re.match(r"\s*s_branch (\.LBB\d+_\d+)", ln)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            st = mix(body[labels[m.group(1)]:i])
            n = sum(st.values())
            nm = st.get("mfma", 0)
            key = (1, nm / max(n, 1)) if nm >= 4 else (0, n)
            if best is None or key > best[0]:
                best = (key, st, i - labels[m.group(1)])
    return None if best is None else dict(lines=best[2], **best[1])


def demangle(names):
    if not FILT:
        return {n: n for n in names}
    out = subprocess.run([FILT], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
    return dict(zip(names, out))


def report(src: str, work: str):
    asm = os.path.join(work, f"{src}.s")
    r = subprocess.run([HIPCC] + FLAGS + ["-I", CSRC, "-I", os.path.join(ROOT, "include"), os.path.join(CSRC, f"{src}.hip"), "-o", asm],
                       capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(r.stderr[-2000:])
    s = open(asm).read()
    rows = []
    for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", s, re.S):
        name, meta = m.group(1), m.group(2)

        def g(key, default=0):
            mm = re.search(key + r"\s+(\S+)", meta)
            return int(mm.group(1)) if mm else default
        i = s.index("\n" + name + ":")
        body = s[i:s.index("s_endpgm", i)].split("\n")
        vg = g(r"\.amdhsa_next_free_vgpr")
        rows.append(dict(symbol=name, file=f"{src}.hip", vgprs=vg, accum_offset=g(r"\.amdhsa_accum_offset"), sgprs=g(r"\.amdhsa_next_free_sgpr"),
                         scratch_bytes=g(r"\.amdhsa_private_segment_fixed_size"), static_lds_bytes=g(r"\.amdhsa_group_segment_fixed_size"),
                         waves_per_simd_by_vgprs=max(1, min(8, 512 // max(8, (vg + 7) // 8 * 8))), whole_kernel=mix(body), hottest_loop=hottest_loop(body)))
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    ap.add_argument("--filter", default=None, help="only kernels whose demangled name contains this")
    ap.add_argument("--sources", default=",".join(SOURCES))
    args = ap.parse_args()
    rows = []
    with tempfile.TemporaryDirectory() as td:
        for src in args.sources.split(","):
            rows += report(src, td)
    names = demangle([r["symbol"] for r in rows])
    for r in rows:
        r["kernel"] = re.sub(r"\(anonymous namespace\)::", "", names[r["symbol"]]).split("(")[0].replace("void ", "")
    if args.filter:
        rows = [r for r in rows if args.filter in r["kernel"]]
    rows.sort(key=lambda r: (r["file"], r["kernel"]))
    for r in rows:
        hl = r["hottest_loop"] or {}
        vm = (hl.get("valu", 0) + hl.get("transcendental", 0) + hl.get("convert", 0)) / hl["mfma"] if hl.get("mfma") else float("nan")
        print(f'{r["file"]:16s} {r["kernel"][:58]:58s} vgpr {r["vgprs"]:3d} scratch {r["scratch_bytes"]:4d} waves/SIMD {r["waves_per_simd_by_vgprs"]} | loop: '
              f'mfma {hl.get("mfma", 0):3d} valu {hl.get("valu", 0):4d} trans {hl.get("transcendental", 0):3d} cvt {hl.get("convert", 0):3d} lds {hl.get("lds", 0):3d} '
              f'vmem {hl.get("vmem", 0):3d} scratch {hl.get("scratch", 0):2d} salu {hl.get("salu", 0):4d} wait {hl.get("wait", 0):3d}  V/M {vm:5.1f}')
    spills = [r["kernel"] for r in rows if r["scratch_bytes"]]
    print(f"{len(rows)} kernels; with scratch (spills): {spills if spills else 'none'}")
    if args.out:
        with open(args.out, "w") as f:
            json.dump({"tool": "tools/isa_report.py", "flags": FLAGS, "kernels": rows}, f, indent=1)
    return rows


if __name__ == "__main__":
    sys.exit(0 if main() is not None else 1)
