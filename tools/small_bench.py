import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unigen_amd import ops
dev = torch.device("cuda:0")
for M, N, K in [(4, 18432, 3072), (4, 9216, 3072), (8, 18432, 3072)]:
    x = torch.randn(M, K, device=dev).bfloat16(); w = torch.randn(N, K, device=dev).bfloat16(); b = torch.zeros(N, device=dev).bfloat16()
    o = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    for _ in range(3): ops.small_linear(x, w, b, o, silu_in=True)
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): ops.small_linear(x, w, b, o, silu_in=True)
    e1.record(); torch.cuda.synchronize(); ms = e0.elapsed_time(e1) / 20
    print(f"small_linear M={M} N={N} K={K}: {ms*1e3:.1f} us  {N*K*2/ms/1e9:.2f} TB/s")
