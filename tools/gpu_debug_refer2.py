import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from musev_amd import ops
from musev_amd.models.attention_processor import ReferEmbFuseAttention
from musev_amd.models.runtime import Geo
from oracle import unet3d
import kernel_cases as kc

torch.manual_seed(0)
# (1) kernel level: two segments from different tensors / strides / lengths
for (d, T, B, lq, lref) in [(40, 5, 2, 256, 256), (80, 5, 2, 16, 64), (80, 3, 2, 64, 16), (40, 5, 2, 256, 64)]:
    heads = 8; c = heads * d; nb = B * T
    qkv = kc._rand((nb * lq, 3 * c), 1)
    kvr = kc._rand((B * lref, 2 * c), 2)
    q, k, v = qkv[:, :c], qkv[:, c:2*c], qkv[:, 2*c:]
    got = ops.attention(q, [(kvr[:, :c], kvr[:, c:], lref, T, 1, 0), (k, v, lq, 1, 1, 0)], nb, lq, heads, d, d ** -0.5)
    bidx = [n // T for n in range(nb)]
    ks = torch.cat([kvr[:, :c].reshape(B, lref, c)[bidx], k.reshape(nb, lq, c)], 1)
    vs = torch.cat([kvr[:, c:].reshape(B, lref, c)[bidx], v.reshape(nb, lq, c)], 1)
    ref = kc._attn_ref(q.reshape(nb, lq, c), ks, vs, heads, d, d ** -0.5)
    print("kernel 2seg", d, T, lq, lref, "maxerr", (got.float() - ref).abs().max().item(), flush=True)

# (2) module level
for (C, T, B, h, hr) in [(320, 5, 2, 16, 16), (640, 5, 2, 4, 8), (640, 4, 2, 8, 8)]:
    m = ReferEmbFuseAttention(query_dim=C, heads=8, dim_head=C // 8)
    g = torch.Generator().manual_seed(5)
    sd = {}
    for k_, p in m.state_dict().items():
        sd["r." + k_] = torch.randn(p.shape, generator=g) * (0.05 if p.ndim == 2 else 0.1)
    m.load_state_dict({k_[2:]: v for k_, v in sd.items()})
    m = m.half().cuda()
    x = torch.randn(B * T, C, h, h, generator=g)
    ref_emb = torch.randn(B, C, 1, hr, hr, generator=g)
    want = unet3d.refer_emb_fuse_attention(sd, "r", x, ref_emb, 8, T)  # [(b t), c, h, w]
    rows = x.permute(0, 2, 3, 1).reshape(-1, C).half().cuda().contiguous()
    got = m.hip_forward(rows, ref_emb.cuda(), Geo(B, T, h, h))
    got = got.float().cpu().reshape(B * T, h, h, C).permute(0, 3, 1, 2)
    print("module refer", C, T, h, hr, "maxerr", (got - want).abs().max().item(), "refmax", want.abs().max().item(), flush=True)
