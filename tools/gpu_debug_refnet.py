import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import model_cases as mc
from oracle import unet3d
from musev_amd.models.unet_loader import load_unet_by_name

arch = mc.ARCHS["small3"]
cfg = unet3d.flavour_config("musev_referencenet", **arch)
sd = unet3d.init_state_dict(cfg, 3)
model = load_unet_by_name("musev_referencenet", sd_unet_model=sd, dtype=torch.float16, **arch).to("cuda")

def run(tag, drop=(), n_cond=1, t=5, override=None):
    x, ehs, kw = mc.make_inputs(cfg, 2, t, 16, 16, seed=103, n_cond=n_cond)
    for d in drop:
        kw.pop(d, None)
    if override:
        kw.update(override)
    ref = unet3d.unet3d_forward(sd, cfg, x, torch.tensor(601), ehs, **kw)
    hkw = {k: mc.to_dev(v, "cuda") for k, v in kw.items()}
    got = model(x.cuda(), torch.tensor(601, device="cuda"), encoder_hidden_states=ehs.cuda(), return_dict=False, **hkw)[0]
    err = (got.float().cpu() - ref).abs()
    # per-frame error
    pf = err.amax(dim=(0, 1, 3, 4)).tolist()
    print(tag, "max", round(err.max().item(), 5), "mean", round(err.mean().item(), 6), "per-frame", [round(v, 4) for v in pf], flush=True)

run("all")
run("no_refer", drop=("down_block_refer_embs", "mid_block_refer_emb"))
run("no_clip", drop=("vision_clip_emb",))
run("no_refer_no_clip", drop=("down_block_refer_embs", "mid_block_refer_emb", "vision_clip_emb"))
run("no_cond", n_cond=0, t=4)
run("no_cond_no_refer_no_clip", n_cond=0, t=4, drop=("down_block_refer_embs", "mid_block_refer_emb", "vision_clip_emb"))
run("rate1", override={"sample_frame_rate": 1})
run("ip_scale0", override={"ip_adapter_scale": 0.0})
