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
