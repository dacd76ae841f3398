"""CPU (-m "not gpu"): the product's Python wrappers (musev_amd.ops) -> ctypes -> the C ABI -> the REAL kernel sources, compiled
for the host and executed thread-per-lane (tests/sim_lib.py, tests/cpu_sim/hip/hip_runtime.h), against the same torch fp32
reference expressions the kernels are verified against on the MI355X (tests/kernel_cases.py, here with DEV = "cpu" and small
shapes).  Functional only -- see the header for what the simulator does and does not model.

Every case runs in its own interpreter under a hard timeout: a simulated kernel that deadlocks on a barrier sits in C code
and cannot be interrupted from within the process."""
import json
import os
import subprocess
import sys

import pytest

import sim_lib

HERE = os.path.dirname(os.path.abspath(__file__))

CASES = {
    "tr16_probe": "kc.case_tr16_probe()",
    "gemm": "kc.case_gemm(M=200, N=320, K=256)",
    "gemm_two_src": "kc.case_gemm(M=130, N=160, K=448, two_src=True)",
    "gemm_plain": "kc.case_gemm(M=64, N=64, K=64, epilogue=False)",
    "gemm_geglu": "kc.case_gemm_geglu(M=77, C=64)",
    "gemm_ln": "kc.case_gemm_ln(M=150, N=320, K=320, residual=True)",
    "gemm_ln_geglu": "kc.case_gemm_ln(M=77, N=512, K=64, geglu=True, offset=5.0)",
    "gemm_ln_every_tile": "kc._all_ok([kc.case_gemm_ln(M=70, N=256, K=128, seed=24 + c, cfg=c) for c in range(19)])",
    "colstats_conv": "kc.case_colstats_groupnorm(n=2, h=8, w=16, cin=64, c=64)",
    "colstats_conv_two_src_seam": "kc.case_colstats_groupnorm(n=2, h=8, w=16, cin=64, c=64, c2=32)",
    "colstats_tconv": "kc.case_colstats_groupnorm(kind='tconv', n=3, h=8, w=16, c=64)",
    "colstats_linear_ragged": "kc.case_colstats_groupnorm(kind='linear', n=3, h=8, w=8, cin=64, c=96, rows_mul=1)",
    "colstats_linear_fold_of_256": "kc.case_colstats_groupnorm(kind='linear', n=1, h=96, w=96, cin=64, c=64, cfg=1)",  # 576 pairs per group: the 256-thread fold played by one wave
    "colstats_every_tile": "kc._all_ok([kc.case_colstats_groupnorm(n=2, h=8, w=16, cin=64, c=64, seed=340 + c, cfg=c) for c in range(19)])",
    "gn_fold_linear_spatial": "kc.case_gn_fold_linear(n=3, h=8, w=8, cin=64, c=64, n_out=96)",
    "gn_fold_linear_temporal": "kc.case_gn_fold_linear(kind='temporal', n=3, h=8, w=8, cin=64, c=64, n_out=64, seed=910)",
    "gn_fold_linear_every_tile": "kc._all_ok([kc.case_gn_fold_linear(n=2, h=8, w=16, cin=64, c=64, n_out=64, seed=940 + c, cfg=c) for c in range(19)])",
    "carry_linear": "kc.case_carry(kind='linear', n=2, h=8, w=12, cin=64, c=160)",
    "carry_conv": "kc.case_carry(kind='conv', n=2, h=8, w=12, cin=64, c=64)",
    "carry_tconv": "kc.case_carry(kind='tconv', n=3, h=4, w=12, c=64)",
    "carry_ragged": "kc.case_carry(kind='linear', n=1, h=5, w=15, cin=72, c=72)",
    "carry_every_tile": "kc._all_ok([kc.case_carry(kind=('linear', 'conv', 'tconv')[c % 3], n=2, h=4, w=12, cin=64, c=64, seed=640 + c, cfg=c) for c in range(19)])",
    "carry_256x320": "kc._all_ok([kc.case_carry(kind=k, n=3, h=8, w=12, cin=64, c=c, seed=660 + i, cfg=6) for i, (k, c) in enumerate((('linear', 320), ('conv', 64), ('tconv', 64)))])",
    "ffn_fused": "kc.case_ffn_fused(M=200)",
    "ffn_fused_one_block_no_bias": "kc.case_ffn_fused(M=70, seed=720, with_bias=False, offset=3.0)",
    "tail_carry": "kc.case_tail_carry(n=2, h=8, w=8, c=64)",
    "conv3x3": "kc.case_conv3x3(n=2, h=8, w=12, c1=64, cout=64)",
    "conv3x3_two_src": "kc.case_conv3x3(n=2, h=8, w=12, c1=64, c2=64, cout=64)",
    "conv3x3_stride2_odd": "kc.case_conv3x3(n=2, h=7, w=9, c1=64, cout=64, stride=2)",
    "conv3x3_upsample": "kc.case_conv3x3(n=1, h=6, w=8, c1=64, cout=64, upsample=True)",
    "tconv3": "kc.case_tconv3(b=2, t=5, hw=12, c=64)",
    "groupnorm": "kc.case_groupnorm(n=3, rows=50, c1=64)",
    "groupnorm_two_src": "kc.case_groupnorm(n=2, rows=50, c1=64, c2=32, silu=False)",
    "groupnorm_one_launch": "kc.case_groupnorm(n=2, rows=70, c1=256, c2=256)",          # whole-octet groups: gn_small_kernel
    "groupnorm_one_launch_16_waves": "kc.case_groupnorm(n=1, rows=1100, c1=1024, silu=False)",   # 1024-thread blocks, slab in registers
    "groupnorm_one_launch_reread": "kc.case_groupnorm(n=1, rows=2100, c1=1024, silu=False)",     # > 8 rows per thread: second pass re-reads
    "layernorm": "kc.case_layernorm(rows=99, c=64)",
    "attention_self": "kc.case_attention_self(d=40, b=1, t=2, lq=70, cond_idx=1)",
    "attention_self_ref": "kc.case_attention_self_ref(d=40, b=1, t=2, lq=70, lr=40, cond_idx=1)",
    "attention_groups": "kc.case_attention_groups(d=40, nb=4, t=2, lq=70)",
    "gemm_weight_stationary": "kc.case_gemm_weight_stationary(M=200, N=640, K=128)",
    "gemm_weight_stationary_split": "kc.case_gemm_weight_stationary(M=130, N=640, K=1024, splitk=4, seed=885)",
    "tsa_block": "kc.case_tsa_block(b=1, t=13, hw=8, seed=720)",
    "tsa_block_t5_two_items": "kc.case_tsa_block(b=2, t=5, hw=8, seed=730, offset=0.7)",
    "xab_block": "kc.case_xab_block(nkvb=2, rows_per_kvb=128)",
    "xab_block_ragged_5_keys": "kc.case_xab_block(nkvb=2, rows_per_kvb=128, n_keys=5, seed=780, with_bias=False, ragged=37, offset=0.7)",
    "attention_resident": "kc.case_attention_resident(d=40, nb=2, t=2, lq=36)",
    "attention_resident_text_ip": "kc.case_attention_resident(d=40, nb=2, t=2, lq=36, face=False, seed=99)",
    "attention_resident_rows_per_block": "kc._all_ok([kc.case_attention_resident(d=40, nb=2, t=2, lq=70, groups=False, seed=101, rows=r) for r in (16, 48, 512)])",
    "attention_resident_d80": "kc.case_attention_resident(d=80, nb=4, t=2, lq=50, groups=True, seed=91)",
    "attention_resident_text_128": "kc.case_attention_resident(d=40, nb=2, t=1, lq=300, lk=128, groups=False, seed=94)",
    "attention_resident_5_heads": "kc.case_attention_resident(d=40, nb=3, t=3, lq=17, lk=5, groups=False, seed=93, heads=5)",
    "attention_groups_d80_spike": "kc.case_attention_groups(d=80, nb=2, t=2, lq=40, lk=150, spike=True)",
    "attention_cross_ip": "kc.case_attention_cross(d=80, nb=4, t=2, lq=40)",
    "temporal_attention": "kc.case_temporal_attention(b=1, t=13, hw=9, d=40)",
    "temporal_attention_d80_items_per_wave": "kc.case_temporal_attention(b=2, t=13, hw=1100, d=80)",   # 17 600 items: 2 per wave
    "temporal_attention_d160_t4": "kc.case_temporal_attention(b=1, t=4, hw=5, d=160)",
    "temporal_attention_t20_valu": "kc.case_temporal_attention(b=1, t=20, hw=3, d=80)",
    "conv_in_out": "kc.case_conv_in_out()",
    "timestep_embedding": "kc.case_timestep_embedding()",
    "layout_and_misc": "kc.case_layout_and_misc()",
    "upsample_nearest": "kc.case_upsample_nearest()",
    "window_loop": "kc.case_window_loop()",
    "cfg_affine_step": "kc.case_cfg_affine_step()",
}

_RUNNER = """
import json, os, sys
sys.path.insert(0, {root!r}); sys.path.insert(0, {here!r})
import torch
torch.set_num_threads(2)
import sim_lib
class MP:
    def setattr(self, o, n, v): setattr(o, n, v)
    def setenv(self, k, v): os.environ[k] = v
sim_lib.install(MP(), {so!r})
import kernel_cases as kc
kc.DEV = "cpu"
res = {expr}
print("RESULT " + json.dumps({{k: v for k, v in res.items() if isinstance(v, (bool, int, float, str))}}))
"""


@pytest.fixture(scope="module")
def sim_so(tmp_path_factory):
    if not os.path.exists(sim_lib.CLANG):
        pytest.skip("ROCm host clang not available")
    return sim_lib.build(tmp_path_factory.mktemp("sim_lib"))


_DEFAULT = ("tr16_probe", "upsample_nearest", "gemm", "gemm_geglu", "gemm_ln", "gemm_ln_geglu", "colstats_conv_two_src_seam", "colstats_tconv", "gn_fold_linear_spatial", "gn_fold_linear_temporal", "carry_linear", "carry_conv", "carry_tconv", "carry_256x320", "ffn_fused", "tail_carry", "conv3x3_two_src", "tconv3", "groupnorm_two_src", "groupnorm_one_launch", "groupnorm_one_launch_16_waves", "groupnorm_one_launch_reread", "layernorm", "attention_self_ref", "attention_groups", "attention_resident", "attention_resident_text_ip", "attention_resident_5_heads", "gemm_weight_stationary", "tsa_block", "xab_block", "xab_block_ragged_5_keys",
            "temporal_attention", "temporal_attention_d160_t4", "window_loop", "cfg_affine_step")
# ("attention_self" -- two segments -- is covered by "attention_self_ref": the same duplicate-segment path + a third segment)


def _run_case(so, name):
    code = _RUNNER.format(root=sim_lib.ROOT, here=HERE, so=so, expr=CASES[name])
    try:
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
        return r.returncode, r.stdout, r.stderr
    except subprocess.TimeoutExpired as ex:
        return -9, "", f"timed out: {ex}"


@pytest.fixture(scope="module")
def case_results(sim_so, request):
    """every SELECTED case of this module, run once, four child processes at a time (each case is its own interpreter on the simulated
    library, two torch threads: the cases are independent, and one after the other they were half of the CPU suite's time)"""
    from concurrent.futures import ThreadPoolExecutor
    full = bool(os.environ.get("MUSEV_SIM_FULL"))
    names = []
    for it in request.session.items:
        if getattr(it, "originalname", "") == "test_kernel_case_through_the_simulated_library":
            n = it.callspec.params["name"]
            if (full or n in _DEFAULT) and n not in names:
                names.append(n)
    with ThreadPoolExecutor(max_workers=4) as pool:
        return dict(zip(names, pool.map(lambda n: _run_case(sim_so, n), names)))


@pytest.mark.parametrize("name", list(CASES))
def test_kernel_case_through_the_simulated_library(case_results, name):
    if name not in _DEFAULT and not os.environ.get("MUSEV_SIM_FULL"):
        pytest.skip("the default CPU suite runs a representative subset (suite time); MUSEV_SIM_FULL=1 runs every case")
    rc, out, err = case_results[name]
    assert rc == 0, (rc, err[-1500:])
    line = [ln for ln in out.splitlines() if ln.startswith("RESULT ")][-1]
    res = json.loads(line[len("RESULT "):])
    assert res["ok"], res


_POSEGUIDER = """
import os, sys
sys.path.insert(0, {root!r}); sys.path.insert(0, {here!r})
import numpy as np, torch
torch.set_num_threads(2)
import sim_lib
class MP:
    def setattr(self, o, n, v): setattr(o, n, v)
    def setenv(self, k, v): os.environ[k] = v
sim_lib.install(MP(), {so!r})
from golden_cases import POSEGUIDER_CASES, poseguider_case_inputs
from oracle import poseguider as opg
from musev_amd.models.controlnet import PoseGuider
c = POSEGUIDER_CASES["default_b2"]
sd = opg.init_state_dict(opg.param_shapes(c["emb"], c["cond"], c["ch"]), c["weight_seed"])
net = PoseGuider.from_pretrained(sd, conditioning_embedding_channels=c["emb"], conditioning_channels=c["cond"], block_out_channels=c["ch"]).half()
net._device_check = False
got = net(poseguider_case_inputs(c))
want = torch.from_numpy(np.load(os.path.join({here!r}, "golden", "reference_poseguider_default_b2.npz"))["out"])
print("RESULT", float((got.float() - want).abs().max()))
"""


def test_poseguider_module_through_the_simulated_library(sim_so):
    """the whole PoseGuider module (layout kernels + eight mv_conv3x3_direct_f16 launches with fused bias / SiLU) on the
    simulated kernels against the output recorded from the reference's own class"""
    r = subprocess.run([sys.executable, "-c", _POSEGUIDER.format(root=sim_lib.ROOT, here=HERE, so=sim_so)], capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, (r.returncode, r.stderr[-1500:])
    err = float([ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")][-1].split()[1])
    assert err < 1e-2, err


_UNET = """
import os, sys
sys.path.insert(0, {root!r}); sys.path.insert(0, {here!r})
import torch
torch.set_num_threads(2)
import sim_lib
class MP:
    def setattr(self, o, n, v): setattr(o, n, v)
    def setenv(self, k, v): os.environ[k] = v
sim_lib.install(MP(), {so!r})
from oracle import unet3d
from musev_amd.models.unet_loader import load_unet_by_name
from model_cases import ARCHS, make_inputs
flavour = {flavour!r}
over = ARCHS["small" if flavour == "musev" else "small3"]
cfg = unet3d.flavour_config(flavour, **over)
sd = unet3d.init_state_dict(cfg, 3)
x, ehs, kw = make_inputs(cfg, b=1, t=3, h=8, w=8, seed=103, n_cond=1)
ref = unet3d.unet3d_forward(sd, cfg, x, torch.tensor(601), ehs, **kw)
model = load_unet_by_name(flavour, sd_unet_model=sd, dtype=torch.float16, **over)
model._device_check = False
got = model(x, torch.tensor(601), encoder_hidden_states=ehs, return_dict=False, **kw)[0]
print("RESULT", float((got.float() - ref).abs().max()))
"""


@pytest.mark.parametrize("flavour", ["musev", "musev_referencenet"])
def test_unet_forward_through_the_simulated_library(sim_so, flavour):
    """the WHOLE UNet3DConditionModel forward (small architecture, B = 1, T = 3, 8x8 latents) on the simulated kernels against
    the oracle -- 25-30 minutes of simulation per flavour, so only with MUSEV_SIM_MODEL=1 (recorded results: DESIGN.md 7b)"""
    if not os.environ.get("MUSEV_SIM_MODEL"):
        pytest.skip("half an hour of simulation: set MUSEV_SIM_MODEL=1")
    r = subprocess.run([sys.executable, "-c", _UNET.format(root=sim_lib.ROOT, here=HERE, so=sim_so, flavour=flavour)], capture_output=True,
                       text=True, timeout=3 * 3600)
    assert r.returncode == 0, (r.returncode, r.stderr[-1500:])
    err = float([ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")][-1].split()[1])
    assert err < 1e-2, err
