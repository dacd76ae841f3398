"""one kernel for a PMC pass: the level-0 reference-only self-attention (nb 13, Lq 4096, Lkv 8192, d 40), 3 launches"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from musev_amd import ops
nb, lq, d, t, heads = 13, 4096, 40, 13, 8
c = heads * d
qkv = torch.randn(nb * lq, 3 * c, device="cuda").half()
q, k, v = qkv[:, :c], qkv[:, c:2 * c], qkv[:, 2 * c:]
for _ in range(3):
    ops.attention(q, [(k, v, lq, 1, 1, 0), (k, v, lq, t, t, 0)], nb, lq, heads, d, d ** -0.5)
torch.cuda.synchronize()
