"""Interleaved A/B of ug_flash_attn_fwd variants in ONE process (cdna guide rule 24): UG_ENV_DYNAMIC=1 makes the library re-read its
tuning switches on every call. Usage: python tools/attn_ab.py [dh] ; prints median / min TFLOP/s per (variant, shape) and checks that
every variant returns the same bits as the default."""
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
VARIANTS = [dict(UG_ATTN_PRIO="1", UG_ATTN_WIDE="0", UG_ATTN_DMA="0"), dict(UG_ATTN_PRIO="0", UG_ATTN_WIDE="0", UG_ATTN_DMA="0"),
            dict(UG_ATTN_PRIO="0", UG_ATTN_WIDE="1", UG_ATTN_DMA="0"), dict(UG_ATTN_PRIO="0", UG_ATTN_WIDE="1", UG_ATTN_DMA="1")]
if os.environ.get("ATTN_AB_VARIANTS"):        # e.g. "0,1,0;0,1,1" = prio,wide,dma per variant
    VARIANTS = [dict(zip(("UG_ATTN_PRIO", "UG_ATTN_WIDE", "UG_ATTN_DMA"), v.split(","))) for v in os.environ["ATTN_AB_VARIANTS"].split(";")]
if os.environ.get("ATTN_AB_ENVS"):            # generic: "K=V,K=V;K=V" = one environment per variant (switches not named keep their defaults)
    VARIANTS = [dict(kv.split("=") for kv in v.split(",") if kv) for v in os.environ["ATTN_AB_ENVS"].split(";")]
    ALLKEYS = sorted({k for v in VARIANTS for k in v})
SHAPES = ([(4, 4608, 4608), (4, 4096, 4608)] if os.environ.get("ATTN_AB_SHORT") else [(4, 4608, 4608), (4, 4096, 4608), (4, 8192, 8704), (4, 1536, 1536), (2, 1000, 1003)]) if dh == 128 else [(16, 4096, 4429), (16, 4096, 4096)]
g = torch.Generator(device=dev).manual_seed(0)
for B, Lq, Lkv in SHAPES:
    W8 = 8 * D if os.environ.get("ATTN_AB_WIDE_ROWS") else 3 * D       # the single block's [q | k | v | attn | mlp] rows
    qkv = torch.randn(B, Lkv, W8, generator=g, device=dev).to(torch.bfloat16)
    st = (W8, Lkv * W8)
    outs = [torch.empty(B, Lq, D, device=dev, dtype=torch.bfloat16) for _ in VARIANTS]

    def run(i):
        if os.environ.get("ATTN_AB_ENVS"):
            for k in ALLKEYS:
                os.environ.pop(k, None)
        os.environ.update(VARIANTS[i])
        ops.flash_attn(qkv[0, Lkv - Lq:], qkv[0, 0, D:], qkv[0, 0, 2 * D:], outs[i], batches=B, heads=H, dh=dh, Lq=Lq, Lkv=Lkv,
                       q_strides=st, k_strides=st, v_strides=st, o_strides=(D, Lq * D))
    for i in range(len(VARIANTS)):
        run(i); run(i)
    torch.cuda.synchronize()
    for i in range(1, len(VARIANTS)):
        if os.environ.get("ATTN_AB_ENVS"):       # different tile sizes associate the online softmax differently: report, do not assert bit equality
            d = float((outs[i].float() - outs[0].float()).norm() / outs[0].float().norm())
            print(f"  variant {VARIANTS[i]} vs variant 0: rel_l2 {d:.2e}, identical bits: {torch.equal(outs[i], outs[0])}")
            assert d <= 5e-3
        else:
            assert torch.equal(outs[i], outs[0]), f"variant {VARIANTS[i]} differs from the default"
    times = [[] for _ in VARIANTS]
    heat = os.environ.get("ATTN_AB_HEAT")          # in-application conditions: every timed attention launch directly follows a large GEMM (hot chip, L2 turned over)
    if heat:
        ga, gw, gc = (torch.randn(16384, 3072, generator=g, device=dev).to(torch.bfloat16), torch.randn(12288, 3072, generator=g, device=dev).to(torch.bfloat16) * 0.02,
                      torch.empty(16384, 12288, device=dev, dtype=torch.bfloat16))
    for rnd in range(7):
        for i in range(len(VARIANTS)):
            if heat:
                tot = 0.0
                for _ in range(4):
                    ops.gemm(ga, gw, None, gc, M=16384)
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(); run(i); e1.record(); e1.synchronize()
                    tot += e0.elapsed_time(e1)
                times[i].append(tot / 4)
                continue
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(4):
                run(i)
            e1.record(); e1.synchronize()
            times[i].append(e0.elapsed_time(e1) / 4)
    fl = 4.0 * B * H * Lq * Lkv * dh
    for i, v in enumerate(VARIANTS):
        t = sorted(times[i])
        print(f"dh{dh} B{B} {Lq}x{Lkv} {v}: median {fl / t[len(t) // 2] / 1e9:7.1f}  best {fl / t[0] / 1e9:7.1f} TFLOP/s" +
              ("   rounds: " + " ".join(f"{fl / x / 1e9:.0f}" for x in times[i]) if os.environ.get("ATTN_AB_ROUNDS") else ""), flush=True)
