import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import model_cases as mc
from oracle import unet3d
from musev_amd.models.unet_loader import load_unet_by_name
arch = mc.ARCHS["small3"]
for fl in ("musev_referencenet",):
    cfg = unet3d.flavour_config(fl, **arch)
    sd = unet3d.init_state_dict(cfg, 3)
    model = load_unet_by_name(fl, sd_unet_model=sd, dtype=torch.float16, **arch).to("cuda")
    x, ehs, kw = mc.make_inputs(cfg, 2, 5, 16, 16, seed=103, n_cond=1)
    col_o = {}
    ref = unet3d.unet3d_forward(sd, cfg, x, torch.tensor(601), ehs, collect=col_o, **kw)
    model._collect = {}
    hkw = {k: mc.to_dev(v, "cuda") for k, v in kw.items()}
    got = model(x.cuda(), torch.tensor(601, device="cuda"), encoder_hidden_states=ehs.cuda(), return_dict=False, **hkw)[0]
    for k in col_o:
        if k in model._collect:
            a, b = model._collect[k], col_o[k]
            print(k, tuple(b.shape), "maxerr", round((a - b).abs().max().item(), 5), "refmax", round(b.abs().max().item(), 3), flush=True)
        else:
            print(k, "missing in hip", flush=True)
