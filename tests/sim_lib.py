"""Builds the SIMULATED C-ABI library: the unmodified .hip sources of musev_amd/csrc compiled for x86 against the stand-in
tests/cpu_sim/hip/hip_runtime.h (thread-per-lane execution, MFMA / LDS-DMA / buffer-descriptor models -- see that header), and
lets a test route musev_amd.ops through it.  TEST INFRASTRUCTURE ONLY: the product never imports this module, and the
routing is a pytest monkeypatch of the loaded-library handle -- there is no switch in the product that selects it."""
from __future__ import annotations

import ctypes as C
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SIM = os.path.join(ROOT, "tests", "cpu_sim")
CSRC = os.path.join(ROOT, "musev_amd", "csrc")
CLANG = "/opt/rocm/lib/llvm/bin/clang++"
SOURCES = ("lib", "gemm", "norm", "attention", "elementwise", "ffn", "tsa", "xab")


def transform(text: str) -> str:
    """the only edits: absolute include paths, GCN inline `s_waitcnt` strings -> simulator calls, dynamic LDS declarations ->
    pointers into the simulator's block buffer (static `__shared__` arrays become function-local statics via the header)"""
    text = text.replace('#include "common.h"', '#include "%s"' % os.path.join(CSRC, "common.h"))
    text = text.replace('#include "gemm_tuned.h"', '#include "%s"' % os.path.join(CSRC, "gemm_tuned.h"))
    text = re.sub(r'asm volatile\("s_waitcnt vmcnt\((\d+)\)" ::: "memory"\)', r"sim_waitcnt_vm(\1)", text)
    text = text.replace('asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")', "((void)0)")  # LDS reads are synchronous in the simulator
    text = re.sub(r"extern __shared__ __attribute__\(\(aligned\(16\)\)\) (\w+) (\w+)\[\];", r"\1* \2 = reinterpret_cast<\1*>(sim_smem_buf);", text)
    assert "extern __shared__" not in text
    return text


def build(work, extra_flags=()) -> str:
    srcs = []
    for name in SOURCES:
        dst = os.path.join(str(work), f"{name}_sim.cpp")
        with open(dst, "w") as f:
            f.write(transform(open(os.path.join(CSRC, f"{name}.hip")).read()))
        srcs.append(dst)
    so = os.path.join(str(work), "libmusev_hip_sim.so")
    r = subprocess.run([CLANG, "-O1", "-std=c++17", "-pthread", "-fPIC", "-shared", "-w", "-I", SIM, "-o", so] + list(extra_flags) + srcs,
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-4000:]
    return so


def install(monkeypatch, so_path: str, defer: int = 1):
    """musev_amd.ops -> ctypes -> the simulated library, on CPU tensors, for the duration of one test"""
    from musev_amd import _lib, ops
    lib = C.CDLL(so_path)
    for name, (res, args) in _lib.SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    assert lib.mv_abi_version() == _lib.MV_ABI_VERSION
    monkeypatch.setattr(_lib, "_lib", lib)
    monkeypatch.setattr(ops, "_on_gpu", lambda t: True)     # CPU tensors are this library's "device memory"
    monkeypatch.setattr(ops, "_stream", lambda: None)
    monkeypatch.setenv("SIM_DEFER", str(defer))
    return lib
