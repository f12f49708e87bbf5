"""The cfg2 GEMM launch shapes through torch's F.linear (hipBLASLt on this image) beside ug_gemm_bf16, bias epilogue only on both sides (the fused
epilogues have no single vendor call), interleaved in one process. A yardstick for the tile kernel, not a product path. usage: python tools/vendor_gemm_compare.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PROBE = "probe" in os.environ.get("UG_LIB_PATH", "")
if PROBE:
    os.environ["UG_ENV_DYNAMIC"] = "1"          # the probe library re-reads its switches per call: one-wave-per-SIMD variants beside the product kernel
import torch
import torch.nn.functional as F
from unigen_amd import ops, lib as L

dev, BF = torch.device("cuda:0"), torch.bfloat16
g = torch.Generator(device=dev).manual_seed(0)
rn = lambda *s, sc=1.0: (torch.randn(*s, generator=g, device=dev) * sc).to(BF)
B, NI, T, D = 4, 4096, 512, 3072
SHAPES = [("attn out", B * NI, D, D), ("qkv image", B * NI, 3 * D, D), ("ff up", B * NI, 4 * D, D), ("ff down K=12288", B * NI, D, 4 * D),
          ("single out K=15360", B * (NI + T), D, 5 * D), ("single qkv+mlp", B * (NI + T), 7 * D, D), ("8192^3", 8192, 8192, 8192)]
if os.environ.get("UG_AB_SHAPES") == "sd3":       # cfg5 (UniGenSD3, D = 1536, 16 samples under CFG): K = 1536 is 24 K-tiles per tile
    M5, D5 = 16 * 4096, 1536
    SHAPES = [("sd3 attn out", M5, D5, D5), ("sd3 qkv", M5, 3 * D5, D5), ("sd3 ff up", M5, 4 * D5, D5), ("sd3 ff down K=6144", M5, D5, 4 * D5),
              ("sd3 context rows out", 16 * 333, D5, D5), ("sd3 context qkv", 16 * 333, 3 * D5, D5)]
for label, M, N, K in SHAPES:
    a, w, b = rn(M, K), rn(N, K, sc=0.03), rn(N, sc=0.1)
    out = torch.empty(M, N, device=dev, dtype=BF)
    def ug(**env):
        def f():
            for k in ("UG_GEMM_PWG", "UG_PWG_VAR"):
                os.environ.pop(k, None)
            os.environ.update(env)
            ops.gemm(a, w, b, out, M=M, epilogue=L.EPI_BIAS)
        return f
    runs = [("hipBLASLt (F.linear)", lambda: F.linear(a, w, b)), ("ug_gemm_bf16", ug())]
    if PROBE:
        runs += [("pwg2", ug(UG_GEMM_PWG="4", UG_PWG_VAR="1")), ("pwg2 shared first operand", ug(UG_GEMM_PWG="4", UG_PWG_VAR="17"))]
    for _, f in runs:
        f(); f()
    torch.cuda.synchronize()
    ref = F.linear(a, w, b)
    runs[-1][1]()
    rel = float((out.float() - ref.float()).norm() / ref.float().norm())
    times = [[] for _ in runs]
    for rnd in range(5):
        for i, (_, f) in enumerate(runs):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                f()
            e1.record(); e1.synchronize()
            times[i].append(e0.elapsed_time(e1) / 3)
    med = [sorted(t)[len(t) // 2] for t in times]
    fl = 2.0 * M * N * K
    print(f"{label:22s} {M}x{N}x{K}  " + "  ".join(f"{nm} {fl / m / 1e9:7.1f} TFLOP/s" for (nm, _), m in zip(runs, med)) + f"   vendor / ug {med[0] / med[1]:.3f}   rel diff of the last variant {rel:.1e}", flush=True)
