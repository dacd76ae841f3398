"""CPU (-m "not gpu"): simple HIP kernels that have not run on a GPU yet are ALSO compiled for the host and executed
thread-per-thread (tests/cpu_sim/hip_cpu_sim.h: a block = 256 real threads, __syncthreads = barrier) on small shapes, against
the same torch references the GPU cases use.  The kernel text is extracted from the .hip source, so an indexing / predicate /
staging mistake in the code that ships shows up here; what this cannot show is anything GPU-specific (alignment faults, LDS
size, occupancy) -- that stays with `-m gpu`."""
import math
import os
import re
import shutil
import subprocess

import numpy as np
import pytest
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SIM = os.path.join(ROOT, "tests", "cpu_sim")
CLANG = "/opt/rocm/lib/llvm/bin/clang++"


def _extract(src_path: str, tag: str) -> str:
    src = open(src_path).read()
    m = re.search(r"// \[cpu-sim:begin %s\][^\n]*\n(.*?)// \[cpu-sim:end %s\]" % (tag, tag), src, re.S)
    assert m, f"markers for {tag} not found in {src_path}"
    return m.group(1)


@pytest.fixture(scope="module")
def conv_direct_bin(tmp_path_factory):
    if not os.path.exists(CLANG):
        pytest.skip("ROCm host clang not available")
    work = tmp_path_factory.mktemp("cpu_sim")
    body = _extract(os.path.join(ROOT, "musev_amd", "csrc", "elementwise.hip"), "conv3x3_direct")
    # dynamic LDS ("extern __shared__ ... dsw[]") becomes the per-block heap buffer of the simulator
    body, n = re.subn(r"extern __shared__ __attribute__\(\(aligned\(16\)\)\) half_t dsw\[\];[^\n]*", "dsw_DECL;", body)
    assert n == 1
    for fn in ("hip_cpu_sim.h", "conv3x3_direct_main.cpp"):
        shutil.copy(os.path.join(SIM, fn), work / fn)
    (work / "conv3x3_direct_extract.inc").write_text(body)
    exe = work / "conv3x3_direct"
    subprocess.run([CLANG, "-O1", "-std=c++17", "-pthread", "-o", str(exe), str(work / "conv3x3_direct_main.cpp")], check=True,
                   cwd=work, capture_output=True)
    return work, exe


@pytest.mark.parametrize("cin,cout,h,w,stride,act", [(3, 16, 10, 13, 1, 1), (16, 32, 10, 13, 2, 1), (16, 16, 9, 11, 2, 0),
                                                      (24, 40, 6, 5, 1, 1), (96, 8, 5, 4, 2, 1)])
def test_conv3x3_direct_kernel_on_the_host(conv_direct_bin, cin, cout, h, w, stride, act):
    work, exe = conv_direct_bin
    n = 2
    g = torch.Generator().manual_seed(cin * 100 + cout)
    x = torch.randn(n, cin, h, w, generator=g).half()
    wt = (torch.randn(cout, cin, 3, 3, generator=g) / math.sqrt(9 * cin)).half()
    b = torch.randn(cout, generator=g).half()
    ref = F.conv2d(x.float(), wt.float(), b.float(), stride=stride, padding=1)
    if act:
        ref = F.silu(ref)
    ref = ref.permute(0, 2, 3, 1).reshape(-1, cout)
    xl = x.permute(0, 2, 3, 1).reshape(-1, cin).contiguous()
    wp = wt.reshape(cout, cin, 9).permute(0, 2, 1).reshape(cout, 9 * cin).contiguous()      # tap-major, channel-minor
    with open(work / "in.bin", "wb") as f:
        f.write(xl.numpy().tobytes() + wp.numpy().tobytes() + b.numpy().tobytes())
    subprocess.run([str(exe), str(work / "in.bin"), str(work / "out.bin"), str(cin), str(cout), str(n), str(h), str(w), str(stride), str(act)],
                   check=True, timeout=300)
    got = torch.from_numpy(np.fromfile(work / "out.bin", dtype=np.float16).reshape(ref.shape)).float()
    err = (got - ref).abs()
    assert torch.isfinite(got).all()
    assert (err <= 3e-3 + 2e-3 * ref.abs()).all(), f"max err {err.max().item()}"
