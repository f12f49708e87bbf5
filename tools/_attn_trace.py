import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, ctypes
from unigen_amd import ops, lib as L
dev = torch.device("cuda:0")
H, dh = 24, 128; D = H * dh
B, Lq, Lkv = 4, 4608, 4608
g = torch.Generator(device=dev).manual_seed(0)
qkv = torch.randn(B, Lkv, 3 * D, generator=g, device=dev).to(torch.bfloat16)
out = torch.zeros(B, Lq + 1, D, device=dev, dtype=torch.bfloat16)
st = (3 * D, Lkv * 3 * D)
for scale in (None, -1.0):
    for _ in range(20):
        ops.flash_attn(qkv[0], qkv[0, 0, D:], qkv[0, 0, 2 * D:], out, batches=B, heads=H, dh=dh, Lq=Lq, Lkv=Lkv, q_strides=st, k_strides=st, v_strides=st, o_strides=(D, (Lq + 1) * D), **({} if scale is None else {"scale": scale}))
    torch.cuda.synchronize()
    rawall = out[3, Lq].reshape(-1).view(torch.int64).cpu()
    print("softmax", "on" if scale is None else "SKIPPED")
    for grp, off in (("A", 0), ("B", 256)):
        raw = rawall[off:off + 128].view(16, 8)
        r = {n: (raw[:, b] - raw[:, a]).double().mean().item() for n, a, b in [("Y", 0, 1), ("wait1", 1, 2), ("fetch", 2, 3), ("X", 3, 4), ("wait2", 4, 5)]}
        tot = (raw[1:, 0] - raw[:-1, 0]).double().mean().item()
        print("  group", grp, {k: round(v) for k, v in r.items()}, "tile", round(tot))
