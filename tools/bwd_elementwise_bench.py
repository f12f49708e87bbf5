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


def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for rows, D, rps in [(9216, 3072, 4608), (1024, 3072, 512), (8192, 1536, 4096)]:
    x, dy = (torch.randn(rows, D, device=dev).to(torch.bfloat16) for _ in range(2))
    sc = torch.randn(rows // rps, D, device=dev).to(torch.bfloat16)
    us = timeit(lambda: ops.adaln_modulate_bwd(x, dy, sc, rows_per_sample=rps))
    print(f"adaln_modulate_bwd {rows}x{D}: {us:8.1f} us incl. the partial sums' reduction  ({3 * rows * D * 2 / us / 1e6:5.2f} TB/s of x + dy + dx)", flush=True)
for rows, H, dh in [(9216, 24, 128), (8192, 24, 64)]:
    x, dy = (torch.randn(rows, H * dh, device=dev).to(torch.bfloat16) for _ in range(2))
    w = torch.randn(dh, device=dev).to(torch.bfloat16)
    cos, sin = torch.randn(rows, dh, device=dev), torch.randn(rows, dh, device=dev)
    us = timeit(lambda: ops.qk_rmsnorm_rope_bwd(x, dy, w, cos, sin, rows_per_batch=rows, pos_offset=0, heads=H, dh=dh))
    print(f"qk_rmsnorm_rope_bwd {rows}x{H}x{dh}: {us:8.1f} us incl. the partial sums' reduction  ({3 * rows * H * dh * 2 / us / 1e6:5.2f} TB/s of x + dy + dx)", flush=True)
for n in [9216 * 12288]:
    x, dy = (torch.randn(n, device=dev).to(torch.bfloat16) for _ in range(2))
    us = timeit(lambda: ops.gelu_tanh(x)); print(f"gelu_tanh {n}: {us:8.1f} us ({2 * n * 2 / us / 1e6:5.2f} TB/s)")
    us = timeit(lambda: ops.gelu_tanh_bwd(x, dy)); print(f"gelu_tanh_bwd {n}: {us:8.1f} us ({3 * n * 2 / us / 1e6:5.2f} TB/s)")
