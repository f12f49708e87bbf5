import sys, torch
sys.path.insert(0, '.')
from unigen_amd import ops
dev = torch.device('cuda:0')
for R, C in [(9216, 3072), (9216, 12288), (9216, 9216), (4608, 3072), (1024, 3072), (9216, 15360), (9216, 64)]:
    x = torch.randn(R, C, device=dev).to(torch.bfloat16)
    for _ in range(3): y = ops.transpose(x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): y = ops.transpose(x)
    e1.record(); e1.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    assert torch.equal(y, x.t().contiguous())
    print(f"transpose {R}x{C}: {us:8.1f} us  {2 * R * C * 2 / us / 1e6:7.2f} TB/s (read + write)", flush=True)
    for _ in range(3): s = ops.colsum(x)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(20): s = ops.colsum(x)
    e1.record(); e1.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print(f"colsum    {R}x{C}: {us:8.1f} us  {R * C * 2 / us / 1e6:7.2f} TB/s (read)", flush=True)
