"""-m gpu: whole-network parity, HIP model vs the CPU oracle on identical seeded weights/inputs (|delta|max < 1e-2)."""
import pytest
import torch

from model_cases import MODEL_CASES

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name,fn", MODEL_CASES, ids=[n for n, _ in MODEL_CASES])
def test_unet_parity(name, fn):
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    res = fn()
    assert res["ok"], res


# ---- HIP model vs the outputs of the REFERENCE'S OWN SOURCE (tests/golden/make_reference_goldens.py) -------------------
from golden_cases import UNET_CASES, case_config, case_inputs  # noqa: E402

HIP_GOLDEN_CASES = [n for n, c in UNET_CASES.items() if c["arch"]["block_out_channels"][0] == 320]  # head dims 40 / 80


@pytest.mark.parametrize("name", HIP_GOLDEN_CASES)
def test_unet_matches_reference_golden(name):
    """musev_amd.UNet3DConditionModel (fp16, HIP kernels) against the fp32 output recorded from
    /root/reference/musev/models/unet_3d_condition.py on the same seeded weights and inputs: |delta|max < 1e-2.
    Includes the config-5 inputs (ControlNet residuals + PoseGuider embedding) on a non-square 24x16 latent."""
    import os

    import numpy as np
    from oracle import unet3d
    from musev_amd.models.unet_loader import load_unet_by_name
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    case = UNET_CASES[name]
    cfg = case_config(case)
    sd = unet3d.init_state_dict(cfg, case["weight_seed"])
    x, t, ehs, kw = case_inputs(case, cfg)
    want = torch.from_numpy(np.load(os.path.join(os.path.dirname(__file__), "golden", f"reference_unet_{name}.npz"))["out"])
    model = load_unet_by_name(case["flavour"], sd_unet_model=sd, dtype=torch.float16, **case["arch"]).to("cuda")

    def dev(v):
        if torch.is_tensor(v):
            return v.to("cuda")
        if isinstance(v, (list, tuple)):
            return [dev(u) for u in v]
        return v

    got = model(x.to("cuda"), t.to("cuda"), encoder_hidden_states=ehs.to("cuda"), return_dict=False,
                **{k: dev(v) for k, v in kw.items()})[0]
    torch.cuda.synchronize()
    err = (got.float().cpu() - want).abs().max().item()
    assert torch.isfinite(got).all()
    assert err < 1e-2, f"{name}: |delta|max = {err}"


def _random_full_unet(flavour="musev", seed=3):
    """the 1.42 B-parameter SD-1.5 MuseV architecture with seeded random fp16 weights, built directly on the GPU"""
    from musev_amd.models.layers import bump_pack_epoch
    from musev_amd.models.unet_loader import load_unet_by_name
    dev = torch.device("cuda", 0)
    with torch.device("meta"):
        unet = load_unet_by_name(flavour, dtype=torch.float16)
    unet = unet.to_empty(device=dev)
    gg = torch.Generator(device=dev).manual_seed(seed)
    res_out = ("conv2.weight", "to_out.0.weight", "ff.net.2.weight", "proj_out.weight", "conv4.3.weight")
    with torch.no_grad():
        for name, p in unet.named_parameters():
            if name.endswith("temporal_weight"):
                p.copy_(0.1 + 0.9 * torch.rand(p.shape, generator=gg, device=dev))
            elif p.ndim >= 2:
                p.copy_(torch.randn(p.shape, generator=gg, device=dev) * ((0.3 if name.endswith(res_out) else 1.0) / p[0].numel() ** 0.5))
            elif name.endswith(".weight"):
                p.copy_(1.0 + 0.1 * torch.randn(p.shape, generator=gg, device=dev))
            else:
                p.copy_(0.05 * torch.randn(p.shape, generator=gg, device=dev))
    unet.eval()
    bump_pack_epoch()
    return unet


def test_full_size_batch_independence():
    """BASELINE config-2 size (1.42 B parameters, 64x64 latents, 12 + 1 frames): size-independent property of the network --
    the two CFG halves never interact inside the UNet, so one batch-2 forward must equal the two batch-1 forwards (this is
    also what the two-stream schedule of ParallelDenoiser relies on).  Different batch sizes take different GEMM tile
    shapes and GroupNorm row splits, so the comparison is to 5e-3 (half the parity bound), not bitwise."""
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    unet = _random_full_unet()
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(0)
    t, h, w = 13, 64, 64
    x = torch.randn(2, 4, t, h, w, generator=g).to(dev)
    ehs = torch.randn(2, 77, 768, generator=g).to(dev)
    kw = dict(sample_index=torch.arange(1, t, device=dev), vision_conditon_frames_sample_index=torch.tensor([0], device=dev),
              sample_frame_rate=8, return_dict=False)
    ts = torch.tensor(601, device=dev)
    both = unet(x, ts, encoder_hidden_states=ehs, **kw)[0].float()
    again = unet(x, ts, encoder_hidden_states=ehs, **kw)[0].float()
    assert torch.equal(both, again), "the forward must be deterministic"
    assert torch.isfinite(both).all()
    for i in range(2):
        one = unet(x[i:i + 1], ts, encoder_hidden_states=ehs[i:i + 1], **kw)[0].float()
        err = (one - both[i:i + 1]).abs().max().item()
        assert err < 5e-3, f"CFG half {i}: batch-1 vs batch-2 forward differ by {err}"
    assert (both[0] - both[1]).abs().max().item() > 1e-3, "the halves see different prompts and must differ"
