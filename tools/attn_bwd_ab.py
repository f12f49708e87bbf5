"""Interleaved A/B of ug_flash_attn_bwd variants in ONE process (UG_ENV_DYNAMIC=1: the library re-reads its switches per call).
Usage: python tools/attn_bwd_ab.py [dh]; env ATTN_BWD_AB="UG_ATTN_BWD_DMA=0;UG_ATTN_BWD_DMA=1" lists the variants (first = reference bits).
Prints median / best TFLOP/s of the ALGORITHMIC backward FLOPs (10 B H Lq Lkv dh) and checks every variant returns the reference's bits."""
import os, sys
os.environ["UG_ENV_DYNAMIC"] = "1"
# the variants exist only in the probe library: python -m unigen_amd.build --probe (tools/probe/README.md)
os.environ.setdefault("UG_LIB_PATH", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "probe", "libunigen_hip_probe.so"))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unigen_amd import ops

dev = torch.device("cuda:0")
dh = int(sys.argv[1]) if len(sys.argv) > 1 else 128
H = 24
D = H * dh
VARIANTS = [dict(kv.split("=") for kv in v.split(",")) for v in os.environ.get("ATTN_BWD_AB", "UG_ATTN_BWD_DMA=0;UG_ATTN_BWD_DMA=1").split(";")]
SHAPES = [(2, 4608, 4608), (1, 8704, 8704), (2, 1000, 1003)] if dh == 128 else [(4, 4429, 4429), (4, 4096, 4096)]
g = torch.Generator(device=dev).manual_seed(0)
for B, Lq, Lkv in SHAPES:
    q = torch.randn(B, Lq, D, generator=g, device=dev).to(torch.bfloat16)
    k = torch.randn(B, Lkv, D, generator=g, device=dev).to(torch.bfloat16)
    v = torch.randn(B, Lkv, D, generator=g, device=dev).to(torch.bfloat16)
    do = torch.randn(B, Lq, D, generator=g, device=dev).to(torch.bfloat16)
    o = torch.empty(B, Lq, D, device=dev, dtype=torch.bfloat16)
    lse = torch.zeros(B, H, (Lq + 63) // 64 * 64, device=dev, dtype=torch.float32)
    ops.flash_attn(q, k, v, o, batches=B, heads=H, dh=dh, Lq=Lq, Lkv=Lkv, lse=lse,
                   q_strides=(D, Lq * D), k_strides=(D, Lkv * D), v_strides=(D, Lkv * D), o_strides=(D, Lq * D))
    outs = [None] * len(VARIANTS)

    def run(i):
        os.environ.update(VARIANTS[i])
        outs[i] = ops.flash_attn_bwd(q, k, v, o, do, heads=H, lse=lse)
    for i in range(len(VARIANTS)):
        run(i); run(i)
    torch.cuda.synchronize()
    for i in range(1, len(VARIANTS)):
        for a, b, nm in zip(outs[i], outs[0], ("dq", "dk", "dv")):
            if os.environ.get("ATTN_BWD_AB_TOL"):     # variants that sum in another order (the pair-scheme dQ kernel vs the DQ mode): relative L2 instead of bits
                e = float((a.float() - b.float()).norm() / b.float().norm())
                assert e <= float(os.environ["ATTN_BWD_AB_TOL"]), f"variant {VARIANTS[i]}: {nm} differs from the reference variant by {e:.3e}"
            else:
                assert torch.equal(a, b), f"variant {VARIANTS[i]}: {nm} differs from the reference variant"
    times = [[] for _ in VARIANTS]
    for rnd in range(7):
        for i in range(len(VARIANTS)):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                run(i)
            e1.record(); e1.synchronize()
            times[i].append(e0.elapsed_time(e1) / 3)
    fl = 10.0 * B * H * Lq * Lkv * dh
    for i, vv in enumerate(VARIANTS):
        t = sorted(times[i])
        print(f"dh{dh} B{B} {Lq}x{Lkv} {vv}: median {fl / t[len(t) // 2] / 1e9:7.1f}  best {fl / t[0] / 1e9:7.1f} TFLOP/s ({t[len(t) // 2]:.3f} ms)", flush=True)
