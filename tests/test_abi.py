"""CPU (-m "not gpu"): the C-ABI library loads and exports every symbol include/musev_hip.h declares; argument
validation returns MV_ERR_INVALID with a message (no kernel is launched, so this runs without a GPU)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    hdr = open(os.path.join(ROOT, "include", "musev_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(mv_[a-z0-9_]+)\s*\(", hdr)))


def test_library_exports_every_declared_symbol():
    from musev_amd import _lib
    lib = _lib.load()
    names = _declared()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/musev_hip.h but not exported"
        assert n in _lib.SIGNATURES, f"{n} has no ctypes signature"
    assert set(_lib.SIGNATURES) == set(names)
    assert lib.mv_abi_version() == _lib.MV_ABI_VERSION


def test_argument_validation_without_gpu():
    from musev_amd import _lib
    lib = _lib.load()
    assert lib.mv_gemm_f16(None, None) == -1
    assert b"null descriptor" in lib.mv_last_error()
    d = _lib.GemmDesc()
    d.a, d.w, d.c = 16, 16, 16
    d.M, d.N, d.K, d.c1, d.lda, d.ldc = 8, 6, 64, 64, 64, 8     # N % 4 != 0
    assert lib.mv_gemm_f16(C.byref(d), None) == -1
    assert b"N % 4" in lib.mv_last_error()
    a = _lib.AttnDesc()
    a.q, a.out, a.nseg, a.d = 16, 16, 1, 64                     # unsupported head dim
    assert lib.mv_attention_f16(C.byref(a), None) == -1
    assert b"head dim" in lib.mv_last_error()
    assert lib.mv_layernorm_f16(16, 8, 16, 8, 4, 12, 16, 16, 1e-5, None) == -1   # C % 8 != 0
    assert lib.mv_temporal_attention_f16(16, 16, 16, 8, 8, 8, 16, 8, 1, 40, 4, 8, 40, 0.1, None) == -1  # T > 32


def test_ops_refuse_cpu_tensors():
    import torch
    from musev_amd import ops
    with pytest.raises(ValueError):
        ops.gemm(torch.zeros(8, 64, dtype=torch.float16), torch.zeros(8, 64, dtype=torch.float16))


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from musev_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.MuseVHipError):
        _lib.load()


# ---- workgroup -> tile order of the implicit-GEMM kernel (host evaluation of the kernel's own inline map) -----------------
def _tile_order(tiles_m, tiles_n, group):
    import numpy as np
    from musev_amd import _lib
    lib = _lib.load()
    n = tiles_m * tiles_n
    tm = np.full(n, -1, dtype=np.int32)
    tn = np.full(n, -1, dtype=np.int32)
    assert lib.mv_gemm_tile_order(tiles_m, tiles_n, group, tm.ctypes.data, tn.ctypes.data) == 0
    return tm, tn


# (tiles_m, tiles_n) of every GEMM shape class of config 2 (M = 106496 / 26624 / 6656 / 1664 rows over 128- and 256-row
# tiles; N = 320 ... 10240 over 160- / 128- / 256- / 320-wide tiles) plus ragged and degenerate grids
_GRIDS = [(832, 2), (832, 6), (832, 20), (416, 10), (208, 4), (208, 12), (208, 40), (104, 20), (52, 8), (52, 24), (52, 80),
          (26, 40), (13, 8), (13, 80), (7, 9), (1, 1), (1, 37), (37, 1), (9, 9), (8, 9), (17, 64), (3, 100)]


@pytest.mark.parametrize("group", [0, 8, 5])
def test_tile_order_is_a_bijection(group):
    """every output tile is produced by exactly one workgroup, for any grid and group size (so results cannot depend on
    the order); group 0 and grids at most `group` n-tiles wide keep the plain m-major order"""
    import numpy as np
    for tiles_m, tiles_n in _GRIDS:
        tm, tn = _tile_order(tiles_m, tiles_n, group)
        assert tm.min() == 0 and tm.max() == tiles_m - 1 and tn.min() == 0 and tn.max() == tiles_n - 1
        flat = tm.astype(np.int64) * tiles_n + tn
        assert len(np.unique(flat)) == tiles_m * tiles_n, (tiles_m, tiles_n, group)
        if group <= 1 or tiles_n <= group:
            tm0, tn0 = _tile_order(tiles_m, tiles_n, 0)
            assert (tm == tm0).all() and (tn == tn0).all()


def test_weight_stationary_order_gives_every_xcd_its_own_n_tiles():
    """group -1 = mv_gemm_desc.tile_order 1 on the small-M levels: a bijection, and the workgroups of one XCD (ids = xcd mod 8) cover
    ~tiles_n / 8 n-tiles x all m-tiles -- its L2 streams 1/8 of the weight matrix (m-major: every XCD touches every n-tile)"""
    import numpy as np
    for tiles_m, tiles_n in [(7, 8), (13, 8), (26, 10), (7, 80), (13, 24), (3, 100), (1, 37), (52, 8)]:
        tm, tn = _tile_order(tiles_m, tiles_n, -1)
        flat = tm.astype(np.int64) * tiles_n + tn
        assert len(np.unique(flat)) == tiles_m * tiles_n
        tm0, tn0 = _tile_order(tiles_m, tiles_n, 8)
        ids = np.arange(tiles_m * tiles_n)
        for xcd in range(8):
            mine = ids % 8 == xcd
            if not mine.any():
                continue
            n_ws, n_mm = len(np.unique(tn[mine])), len(np.unique(tn0[mine]))
            assert n_ws <= -(-tiles_n // 8) + 1, (tiles_m, tiles_n, xcd, n_ws)
            assert n_ws <= n_mm


def test_tile_order_window_locality():
    """what an XCD's L2 must fetch for the ~64 workgroups resident on it: (distinct m-tiles + distinct n-tiles) of 64
    consecutive workgroups of ONE XCD (workgroup ids = xcd mod 8).  Wide grids: ~16 tile-slabs grouped vs ~(1 + 64)
    m-major."""
    import numpy as np
    for tiles_m, tiles_n in [(52, 80), (208, 40), (832, 20), (52, 24)]:
        worst, mean = {}, {}
        for group in (0, 8):
            tm, tn = _tile_order(tiles_m, tiles_n, group)
            costs = []
            for xcd in range(8):
                ids = np.arange(xcd, tiles_m * tiles_n, 8)
                for s in range(0, len(ids) - 63, 16):
                    w = ids[s:s + 64]
                    costs.append(len(np.unique(tm[w])) + len(np.unique(tn[w])))
            worst[group], mean[group] = max(costs), float(np.mean(costs))
        assert worst[8] <= 32, (tiles_m, tiles_n, worst)        # 8 x 8 window = 16, up to ~2x across a group boundary
        assert worst[8] <= worst[0] and mean[8] < 0.9 * mean[0], (tiles_m, tiles_n, worst, mean)


def test_package_import_sets_the_hardware_queue_default_without_overriding_the_user():
    """musev_amd/__init__.py: GPU_MAX_HW_QUEUES=8 unless the user set it (streams that share a hardware queue serialise)"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = "import os, musev_amd; print(os.environ['GPU_MAX_HW_QUEUES'])"
    env = {k: v for k, v in os.environ.items() if k != "GPU_MAX_HW_QUEUES"}
    assert subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, check=True).stdout.strip() == "8"
    env["GPU_MAX_HW_QUEUES"] = "4"
    assert subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, check=True).stdout.strip() == "4"


def test_timeline_tool_stamps_every_launching_entry():
    """tools/gpu_timeline.py wraps the C-ABI entries whose last parameter is the stream (parsed from include/musev_hip.h): every one of
    them must be an exported symbol, and every entry musev_amd.ops launches through must be among them -- a launch the tool does not
    stamp would be missing from the step's timeline"""
    import importlib.util
    import re
    spec = importlib.util.spec_from_file_location("gpu_timeline", os.path.join(ROOT, "tools", "gpu_timeline.py"))
    tl = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(tl)
    entries = tl.launch_entries()
    from musev_amd import _lib
    lib = _lib.load()
    for name in entries:
        assert hasattr(lib, name), name
    src = open(os.path.join(ROOT, "musev_amd", "ops.py")).read()
    used = set(re.findall(r"\.(mv_\w+)\(", src))
    host_only = {"mv_attention_resident_ok", "mv_gemm_choice", "mv_gemm_stats_layout", "mv_gemm_workspace_bytes", "mv_groupnorm_default_nsplit"}
    assert used - host_only <= set(entries), sorted(used - host_only - set(entries))   # (host-side queries launch nothing)
    assert set(entries) <= used, sorted(set(entries) - used)
