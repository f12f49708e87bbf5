"""Sweep ug_gemm_bf16 over M (or any list of M,N,K triples on the command line): time, TFLOP/s, tile rounds. Random data, EPI_BIAS."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unigen_amd import ops, lib as L

dev = torch.device("cuda:0")
shapes = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]] or [(m, 9216, 3072) for m in range(14336, 20481, 1024)]
g = torch.Generator(device=dev).manual_seed(0)
for M, N, K in shapes:
    a = (torch.rand(M, K, generator=g, device=dev) * 2 - 1).to(torch.bfloat16)
    w = (torch.rand(N, K, generator=g, device=dev) * 2 - 1).to(torch.bfloat16)
    b = torch.zeros(N, device=dev, dtype=torch.bfloat16)
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    for _ in range(5):
        ops.gemm(a, w, b, out, M=M, epilogue=L.EPI_BIAS)
    torch.cuda.synchronize()
    n = 20
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        ops.gemm(a, w, b, out, M=M, epilogue=L.EPI_BIAS)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    tiles = ((M + 255) // 256) * ((N + 255) // 256)
    rounds = -(-tiles // 256)
    print(f"{M:6d}x{N:6d}x{K:6d}  {ms:8.3f} ms  {2.0 * M * N * K / ms / 1e9:7.1f} TFLOP/s  tiles {tiles:5d} = {tiles / 256:6.3f} rounds -> {ms * 1e3 / rounds:7.1f} us/round", flush=True)
