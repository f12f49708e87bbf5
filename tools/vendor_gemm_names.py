"""Which hipBLASLt kernels torch's F.linear picks for the cfg2 GEMM shapes (run under rocprofv3 --kernel-trace --stats: the kernel names encode the Tensile configuration)."""
import torch, torch.nn.functional as F
dev, BF = torch.device("cuda:0"), torch.bfloat16
B, NI, T, D = 4, 4096, 512, 3072
for M, N, K in [(B * NI, D, D), (B * NI, 4 * D, D), (B * (NI + T), 7 * D, D), (B * (NI + T), D, 5 * D), (8192, 8192, 8192)]:
    a, w, b = torch.randn(M, K, device=dev).to(BF), torch.randn(N, K, device=dev).to(BF), torch.randn(N, device=dev).to(BF)
    for _ in range(4):
        F.linear(a, w, b)
    torch.cuda.synchronize()
