"""ReferenceNet2D on HIP kernels against the oracle and against the feature maps recorded from the
reference's own ReferenceNet2D.  Marker ``gpu_pending``, NOT ``gpu``: the module was written after the round's GPU budget
was spent and these tests have not run on a GPU yet -- `pytest -m gpu_pending` is the first thing to run next round."""
import os

import numpy as np
import pytest
import torch

from golden_cases import REFNET_CASES, refnet_case_inputs

pytestmark = pytest.mark.gpu_pending


def test_referencenet_matches_reference_golden_and_oracle():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from oracle import referencenet as oref
    from musev_amd.models.referencenet import load_referencenet_by_name
    case = REFNET_CASES["hipw"]
    cfg = oref.referencenet_config(**case["arch"])
    sd = oref.init_state_dict(cfg, case["weight_seed"])
    x, t, ehs = refnet_case_inputs(case, cfg)
    net = load_referencenet_by_name("musev_referencenet", sd, **case["arch"]).to("cuda")
    down, mid, sa = net(x.to("cuda"), t.to("cuda"), encoder_hidden_states=ehs.to("cuda"), num_frames=case["t"], return_ndim=5)
    torch.cuda.synchronize()
    assert sa is None
    gold = np.load(os.path.join(os.path.dirname(__file__), "golden", "reference_referencenet_hipw.npz"))
    with torch.no_grad():
        odown, omid = oref.referencenet_forward(sd, cfg, x, t, ehs, num_frames=case["t"])
    assert len(down) == len(odown)
    for i, d in enumerate(down):
        want = torch.from_numpy(gold[f"down{i}"])
        assert d.shape == want.shape
        assert (d.float().cpu() - want).abs().max().item() < 1e-2, f"down{i} vs reference"
        assert (d.float().cpu() - odown[i]).abs().max().item() < 1e-2, f"down{i} vs oracle"
    assert (mid.float().cpu() - torch.from_numpy(gold["mid"])).abs().max().item() < 1e-2
