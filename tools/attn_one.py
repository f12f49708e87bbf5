"""one attention shape, a few launches (for PMC profiling)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unigen_amd import ops
dev = torch.device("cuda:0"); H, dh = 24, 128; D = H * dh; B, L = 4, 4608
g = torch.Generator(device=dev).manual_seed(0)
qkv = torch.randn(B, L, 3 * D, generator=g, device=dev).to(torch.bfloat16)
out = torch.empty(B, L, D, device=dev, dtype=torch.bfloat16); st = (3 * D, L * 3 * D)
for _ in range(5):
    ops.flash_attn(qkv, qkv[0, 0, D:], qkv[0, 0, 2 * D:], out, batches=B, heads=H, dh=dh, Lq=L, Lkv=L, q_strides=st, k_strides=st, v_strides=st, o_strides=(D, L * D))
torch.cuda.synchronize()
