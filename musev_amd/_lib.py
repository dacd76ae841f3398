"""ctypes binding of libmusev_hip.so (the C ABI declared in include/musev_hip.h).

The product path has no CPU fallback: if the library is missing or an entry point fails, this module raises.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# MUSEV_HIP_LIBRARY: an alternative build of the library (developer A/B runs: another tile table, an experiment build)
LIB_PATH = os.environ.get("MUSEV_HIP_LIBRARY") or os.path.join(_HERE, "csrc", "libmusev_hip.so")

MV_GEMM_LINEAR, MV_GEMM_CONV3X3, MV_GEMM_TCONV3 = 0, 1, 2
MV_ACT_NONE, MV_ACT_SILU = 0, 1
MV_ATTN_MAX_SEG = 4
MV_ABI_VERSION = 13


class MuseVHipError(RuntimeError):
    pass


class GemmDesc(C.Structure):
    _fields_ = [
        ("a", C.c_void_p), ("a2", C.c_void_p), ("w", C.c_void_p), ("c", C.c_void_p),
        ("bias", C.c_void_p), ("rowbias", C.c_void_p), ("residual", C.c_void_p), ("alpha", C.c_void_p),
        ("M", C.c_int64),
        ("N", C.c_int32), ("K", C.c_int32),
        ("lda", C.c_int32), ("lda2", C.c_int32), ("ldc", C.c_int32), ("ldr", C.c_int32), ("ldrb", C.c_int32),
        ("c1", C.c_int32), ("c2", C.c_int32),
        ("mode", C.c_int32), ("stride", C.c_int32), ("upsample", C.c_int32),
        ("hin", C.c_int32), ("win", C.c_int32), ("hout", C.c_int32), ("wout", C.c_int32),
        ("t", C.c_int32), ("hw", C.c_int32),
        ("rows_per_group", C.c_int32), ("act", C.c_int32), ("geglu", C.c_int32),
        ("cfg", C.c_int32), ("splitk", C.c_int32), ("workspace", C.c_void_p), ("workspace_bytes", C.c_int64),
        ("ln_colsum", C.c_void_p), ("ln_colbias", C.c_void_p), ("ln_eps", C.c_float), ("w_group_rows", C.c_int32),
        ("colstats", C.c_void_p), ("colstats_floats", C.c_int64),
        ("residual_lo", C.c_void_p), ("c_lo", C.c_void_p),
        ("tile_order", C.c_int32),
        ("rowbias_lo", C.c_void_p),
    ]


class FfnDesc(C.Structure):
    _fields_ = [
        ("x", C.c_void_p), ("ln_gamma", C.c_void_p), ("ln_beta", C.c_void_p), ("w1", C.c_void_p), ("bias1", C.c_void_p),
        ("w2", C.c_void_p), ("bias2", C.c_void_p), ("residual", C.c_void_p), ("out", C.c_void_p),
        ("M", C.c_int64), ("C", C.c_int32), ("H", C.c_int32), ("ldx", C.c_int32), ("ldr", C.c_int32), ("ldo", C.c_int32),
        ("ln_eps", C.c_float), ("flags", C.c_int32),
    ]


class TsaDesc(C.Structure):
    _fields_ = [
        ("x", C.c_void_p), ("ln_gamma", C.c_void_p), ("ln_beta", C.c_void_p), ("wqkv", C.c_void_p), ("wo", C.c_void_p),
        ("bias_o", C.c_void_p), ("out", C.c_void_p),
        ("B", C.c_int64), ("T", C.c_int32), ("HW", C.c_int32), ("C", C.c_int32), ("heads", C.c_int32), ("d", C.c_int32),
        ("ldx", C.c_int32), ("ldo", C.c_int32), ("ln_eps", C.c_float), ("scale", C.c_float), ("flags", C.c_int32),
    ]


class XabDesc(C.Structure):
    _fields_ = [
        ("x", C.c_void_p), ("ln_gamma", C.c_void_p), ("ln_beta", C.c_void_p), ("wq", C.c_void_p), ("k", C.c_void_p), ("v", C.c_void_p),
        ("wo", C.c_void_p), ("bias_o", C.c_void_p), ("out", C.c_void_p),
        ("M", C.c_int64), ("rows_per_kvb", C.c_int32), ("len", C.c_int32), ("C", C.c_int32), ("heads", C.c_int32), ("d", C.c_int32),
        ("ldk", C.c_int32), ("ldv", C.c_int32), ("ldx", C.c_int32), ("ldo", C.c_int32), ("ln_eps", C.c_float), ("scale", C.c_float),
        ("flags", C.c_int32),
    ]


class AttnSeg(C.Structure):
    _fields_ = [
        ("k", C.c_void_p), ("v", C.c_void_p),
        ("ldk", C.c_int32), ("ldv", C.c_int32), ("len", C.c_int32),
        ("div", C.c_int32), ("mul", C.c_int32), ("add", C.c_int32),
        ("new_group", C.c_int32), ("group_scale", C.c_float),
    ]


class AttnDesc(C.Structure):
    _fields_ = [
        ("q", C.c_void_p), ("out", C.c_void_p),
        ("ldq", C.c_int32), ("ldo", C.c_int32),
        ("nb", C.c_int32), ("lq", C.c_int32), ("heads", C.c_int32), ("d", C.c_int32),
        ("scale", C.c_float), ("nseg", C.c_int32),
        ("seg", AttnSeg * MV_ATTN_MAX_SEG),
        ("accumulate", C.c_int32), ("out_scale", C.c_float), ("resident_kv", C.c_int32),
    ]


# name -> (restype, argtypes); every symbol include/musev_hip.h declares
_vp, _i32, _i64, _f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_float
SIGNATURES = {
    "mv_abi_version": (_i32, []),
    "mv_last_error": (C.c_char_p, []),
    "mv_gemm_f16": (_i32, [C.POINTER(GemmDesc), _vp]),
    "mv_ffn_geglu_f16": (_i32, [C.POINTER(FfnDesc), _vp]),
    "mv_temporal_attn_block_f16": (_i32, [C.POINTER(TsaDesc), _vp]),
    "mv_xattn_block_f16": (_i32, [C.POINTER(XabDesc), _vp]),
    "mv_gemm_workspace_bytes": (_i64, [C.POINTER(GemmDesc)]),
    "mv_gemm_choice": (_i32, [C.POINTER(GemmDesc), _vp, _vp]),
    "mv_gemm_weight_stationary": (_i32, [C.POINTER(GemmDesc)]),
    "mv_gemm_stats_layout": (_i32, [C.POINTER(GemmDesc), _vp, _vp]),
    "mv_gemm_num_configs": (_i32, []),
    "mv_gemm_config_desc": (_i32, [_i32, _vp]),
    "mv_gemm_tile_order": (_i32, [_i32, _i32, _i32, _vp, _vp]),
    "mv_groupnorm_f16": (_i32, [_vp, _vp, _i32, _i32, _i32, _i32, _i64, _i64, _i32, _f32, _vp, _vp, _i32, _vp, _i32,
                                _vp, _i32, _vp, _vp, _vp, _vp]),
    "mv_groupnorm_cs_f16": (_i32, [_vp, _vp, _i32, _i32, _i32, _i32, _i64, _i64, _i32, _f32, _vp, _vp, _i32, _vp, _i32,
                            _vp, _i32, _vp, _i32, _i32, _vp, _vp, _vp, _vp]),
    "mv_groupnorm_cs_fold_linear_f16": (_i32, [_vp, _i32, _i32, _i64, _i64, _i32, _f32, _vp, _vp, _vp, _vp, _i32, _vp, _i32, _i32,
                                        _vp, _vp, _vp, _vp, _vp]),
    "mv_groupnorm_partial_floats": (_i64, [_i64, _i32, _i32]),
    "mv_groupnorm_default_nsplit": (_i32, [_i64, _i64, _i32]),
    "mv_layernorm_f16": (_i32, [_vp, _i32, _vp, _i32, _i64, _i32, _vp, _vp, _f32, _vp]),
    "mv_attention_f16": (_i32, [C.POINTER(AttnDesc), _vp]),
    "mv_attention_resident_ok": (_i32, [C.POINTER(AttnDesc)]),
    "mv_temporal_attention_f16": (_i32, [_vp, _vp, _vp, _i32, _i32, _i32, _vp, _i32, _i32, _i32, _i32, _i32, _i32,
                                         _f32, _vp]),
    "mv_geglu_f16": (_i32, [_vp, _i32, _vp, _i32, _i64, _i32, _vp]),
    "mv_conv3x3_cin_small_f16": (_i32, [_vp, _i32, _vp, _vp, _vp, _vp, _i32, _i64, _i32, _i32, _vp]),
    "mv_conv3x3_direct_f16": (_i32, [_vp, _i32, _vp, _vp, _vp, _i32, _i64, _i32, _i32, _i32, _i32, _vp]),
    "mv_im2col3x3_f16": (_i32, [_vp, _i32, _vp, _i32, _i64, _i32, _i32, _vp]),
    "mv_conv3x3_cout_small_f16": (_i32, [_vp, _vp, _i32, _vp, _vp, _vp, _i32, _i32, _i64, _i32, _i32, _vp]),
    "mv_timestep_embedding_f16": (_i32, [_vp, _i32, _i32, _vp, _vp]),
    "mv_silu_f16": (_i32, [_vp, _vp, _i64, _vp]),
    "mv_add_f16": (_i32, [_vp, _vp, _vp, _vp, _vp, _i64, _vp]),
    "mv_upsample_nearest_f16": (_i32, [_vp, _i32, _vp, _i32, _i64, _i32, _i32, _i32, _i32, _i32, _vp]),
    "mv_zero_rows_f16": (_i32, [_vp, _i32, _vp, _i32, _i32, _vp]),
    "mv_bcthw_to_bthwc_f16": (_i32, [_vp, _i32, _vp, _i32, _i32, _i32, _i32, _vp]),
    "mv_bthwc_to_bcthw_f16": (_i32, [_vp, _i32, _vp, _i32, _i32, _i32, _i32, _i32, _vp]),
    "mv_window_gather": (_i32, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp]),
    "mv_window_scatter_add": (_i32, [_vp, _i32, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _i32, _vp]),
    "mv_softmax_rows_f16": (_i32, [_vp, _i64, _i64, _i32, _vp]),
    "mv_window_units_reduce": (_i32, [_vp, _i64, _vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp]),
    "mv_cfg_ddim_step": (_i32, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _f32, _f32, _f32, _vp]),
    "mv_cfg_affine_step": (_i32, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _f32, _f32, _f32, _vp]),
    "mv_pack_conv_weight_f16": (_i32, [_vp, _i32, _vp, _i32, _i32, _i32, _vp]),
    "mv_probe_tr16": (_i32, [_vp, _vp, _vp]),
}

_lib = None


def load() -> C.CDLL:
    """Load libmusev_hip.so (once) and bind every declared symbol.  Raises MuseVHipError when it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MuseVHipError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(musev_amd has no CPU fallback)")
    # torch first: its wheel bundles its own HIP runtime (libamdhip64, same soname as /opt/rocm's).  Whichever copy the process loads
    # first serves both; with libmusev_hip.so loaded ahead of torch the system runtime came first and torch's device enumeration
    # then failed on the GPU box ("no ROCm-capable device is detected" at the first launch, profiles/r04y_smoke.log: build() and
    # smoke() in one process).  Tensors, streams and graphs are torch's: its runtime is the one that must be resident.
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing: the ABI is incomplete
        fn.restype = res
        fn.argtypes = args
    if lib.mv_abi_version() != MV_ABI_VERSION:
        raise MuseVHipError(f"ABI mismatch: library {lib.mv_abi_version()} vs binding {MV_ABI_VERSION}")
    _lib = lib
    return lib


def check(status: int, what: str) -> None:
    if status != 0:
        msg = load().mv_last_error().decode("utf-8", "replace")
        raise MuseVHipError(f"{what} failed ({status}): {msg}")
