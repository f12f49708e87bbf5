"""Micro-benchmark of ug_flash_attn_fwd on the cfg2 joint-sequence shapes (random gaussian data). UG_ATTN_VERSION=1|2."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unigen_amd import ops

dev = torch.device("cuda:0")
H, dh = 24, 128
D = H * dh
SHAPES = [tuple(int(x) for x in a.split(",")) + ("cli",) for a in sys.argv[1:]] or [(4, 4608, 4608, "base joint/single"), (4, 4096, 4608, "control joint (img q)"), (4, 8192, 8192, "shared0"), (4, 8192, 8704, "shared1"),
          (16, 2048, 2048, "2k ref shape")]
g = torch.Generator(device=dev).manual_seed(0)
for B, Lq, Lkv, label in SHAPES:
    assert Lq <= Lkv, "q rows are the last Lq rows of the packed qkv buffer"
    qkv = torch.randn(B, Lkv, 3 * D, generator=g, device=dev).to(torch.bfloat16)
    out = torch.empty(B, Lq, D, device=dev, dtype=torch.bfloat16)
    st = (3 * D, Lkv * 3 * D)
    run = lambda: ops.flash_attn(qkv[0, Lkv - Lq:], qkv[0, 0, D:], qkv[0, 0, 2 * D:], out, batches=B, heads=H, dh=dh, Lq=Lq, Lkv=Lkv,
                                 q_strides=st, k_strides=st, v_strides=st, o_strides=(D, Lq * D))
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    n = 10
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    tf = 4.0 * B * H * Lq * Lkv * dh / ms / 1e9
    print(f"{label:24s} B{B} {Lq}x{Lkv}  {ms:8.3f} ms  {tf:7.1f} TFLOP/s", flush=True)
