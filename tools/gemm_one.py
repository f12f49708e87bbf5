"""one GEMM shape, a few launches (for PMC profiling)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unigen_amd import ops
dev = torch.device("cuda:0")
M, N, K = [int(v) for v in os.environ.get("MNK", "18432,12288,3072").split(",")]
g = torch.Generator(device=dev).manual_seed(0)
a = (torch.rand(M, K, generator=g, device=dev) * 2 - 1).bfloat16(); w = (torch.rand(N, K, generator=g, device=dev) * 2 - 1).bfloat16()
b = torch.zeros(N, device=dev).bfloat16(); o = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
for _ in range(6): ops.gemm(a, w, b, o, M=M)
torch.cuda.synchronize()
