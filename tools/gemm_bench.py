"""Micro-benchmark of ug_gemm_bf16 on the cfg2 projection shapes (random data). UG_GEMM_FORCE_TILE=128|256 pins the kernel."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unigen_amd import ops, lib as L

dev = torch.device("cuda:0")
SHAPES = [  # (M, N, K, epilogue, label)
    (16384, 9216, 3072, L.EPI_BIAS, "qkv image"),
    (2048, 9216, 3072, L.EPI_BIAS, "qkv text"),
    (16384, 3072, 3072, L.EPI_RES_GATE, "attn out"),
    (16384, 12288, 3072, L.EPI_BIAS_GELU, "ff up"),
    (16384, 3072, 12288, L.EPI_RES_GATE, "ff down"),
    (18432, 9216, 3072, L.EPI_BIAS, "single qkv"),
    (18432, 12288, 3072, L.EPI_BIAS_GELU, "single mlp"),
    (18432, 3072, 15360, L.EPI_RES_GATE, "single out"),
    (18432, 3072, 3072, L.EPI_RES_SCALE, "zero-res"),
    (4096, 4096, 4096, L.EPI_BIAS, "4096^3"),
    (8192, 8192, 8192, L.EPI_BIAS, "8192^3"),
]
g = torch.Generator(device=dev).manual_seed(0)
res = []
for M, N, K, epi, label in SHAPES:
    a = (torch.rand(M, K, generator=g, device=dev) * 2 - 1).to(torch.bfloat16)
    w = (torch.rand(N, K, generator=g, device=dev) * 2 - 1).to(torch.bfloat16)
    b = torch.zeros(N, device=dev, dtype=torch.bfloat16)
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    r = torch.zeros(M, N, device=dev, dtype=torch.bfloat16)
    gate = torch.ones(M // 512 + 1, N, device=dev, dtype=torch.bfloat16)
    kw = dict(M=M, epilogue=epi)
    if epi in (L.EPI_RES_GATE, L.EPI_RES_SCALE):
        kw.update(residual=r)
    if epi == L.EPI_RES_GATE:
        kw.update(gate=gate, gate_ld=N, rows_per_sample=512)
    for _ in range(3):
        ops.gemm(a, w, b, out, **kw)
    torch.cuda.synchronize()
    n = 10
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        ops.gemm(a, w, b, out, **kw)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    tf = 2.0 * M * N * K / ms / 1e9
    res.append(dict(label=label, M=M, N=N, K=K, ms=ms, tflops=tf))
    print(f"{label:12s} {M:6d}x{N:6d}x{K:6d}  {ms:8.3f} ms  {tf:7.1f} TFLOP/s", flush=True)
print("GEMM_BENCH", os.environ.get("UG_GEMM_FORCE_TILE", "auto"), json.dumps(res))
