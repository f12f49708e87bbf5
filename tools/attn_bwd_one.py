"""One ug_flash_attn_bwd call at one shape (for rocprofv3 passes): python tools/attn_bwd_one.py [dh] [B] [L]. Switches via the environment."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unigen_amd import ops

dev = torch.device("cuda:0")
dh = int(sys.argv[1]) if len(sys.argv) > 1 else 128
B = int(sys.argv[2]) if len(sys.argv) > 2 else 2
L = int(sys.argv[3]) if len(sys.argv) > 3 else 4608
H = 24
D = H * dh
g = torch.Generator(device=dev).manual_seed(0)
q, k, v, do = (torch.randn(B, L, D, generator=g, device=dev).to(torch.bfloat16) for _ in range(4))
o = torch.empty(B, L, D, device=dev, dtype=torch.bfloat16)
lse = torch.zeros(B, H, (L + 63) // 64 * 64, device=dev, dtype=torch.float32)
st = (D, L * D)
ops.flash_attn(q, k, v, o, batches=B, heads=H, dh=dh, Lq=L, Lkv=L, lse=lse, q_strides=st, k_strides=st, v_strides=st, o_strides=st)
for _ in range(2):
    ops.flash_attn_bwd(q, k, v, o, do, heads=H, lse=lse)
torch.cuda.synchronize()
print("ok")
