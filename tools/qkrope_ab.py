"""Fused (UG_EPI_QKV_ROPE) vs two-launch q/k RMSNorm + RoPE at the cfg2 projection shapes: element differences and timing."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unigen_amd import ops, lib as L

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
BF = torch.bfloat16
rn = lambda *s: torch.randn(*s, generator=g, device=dev).to(BF)
D, dh, H = 3072, 128, 24


def tables(S):
    ang = torch.rand(S, dh // 2, generator=g, device=dev) * 6.28
    cos, sin = ang.cos().repeat_interleave(2, 1).contiguous(), ang.sin().repeat_interleave(2, 1).contiguous()
    return cos, sin, torch.stack([cos[:, 0::2], sin[:, 0::2]], -1).contiguous()


def timeit(f, n=5):
    f(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


wq, wk = (1 + 0.2 * torch.randn(dh, generator=g, device=dev)).to(BF), (1 + 0.2 * torch.randn(dh, generator=g, device=dev)).to(BF)
K = 3072
# single block
B, Lj = 4, 4608
M = B * Lj
x, w, b = rn(M, K), rn(7 * D, K) * K ** -0.5, rn(7 * D) * 0.1
cos, sin, cs = tables(Lj)
one, two = torch.zeros(M, 8 * D, device=dev, dtype=BF), torch.zeros(M, 8 * D, device=dev, dtype=BF)
kw = dict(M=M, ldc=8 * D, epilogue=L.EPI_BIAS_GELU, gelu_from_n=3 * D, c_shift_from_n=3 * D, c_shift=D)
def f_two():
    ops.gemm(x, w, b, two, **kw)
    ops.qk_rmsnorm_rope(two, batches=B, rows_per_batch=Lj, ld=8 * D, q_off=0, k_off=D, heads=H, dh=dh, wq_b=wq, wk_b=wk, split=0, cos=cos, sin=sin)
def f_one():
    ops.gemm(x, w, b, one, qk_rope=ops.QkRope(wq, wk, cs, Lj, 0, 2 * D), **kw)
t2, t1 = timeit(f_two), timeit(f_one)
diff = (one.float() - two.float()).abs()
print("QKROPE_AB single", json.dumps(dict(us_two=round(t2, 1), us_fused=round(t1, 1), differing=float((diff > 0).float().mean()), max_abs=float(diff.max()),
      rel=float(diff.norm() / two.float().norm()), rest_equal=bool(torch.equal(one[:, 2 * D:], two[:, 2 * D:])))), flush=True)
# double block sample rows
B, Lc, Ls = 4, 512, 4096
Ljn = Lc + Ls
M = B * Ls
x, w, b = rn(M, K), rn(3 * D, K) * K ** -0.5, rn(3 * D) * 0.1
cos, sin, cs = tables(Ljn)
junk = rn(B * Ljn, 3 * D)
one, two = junk.clone(), junk.clone()
def f_two():
    ops.gemm(x, w, b, two[Lc:], M=M, ldc=3 * D, c_map=ops.RowMap(Ls, Ljn))
    ops.qk_rmsnorm_rope(two[Lc:], batches=B, rows_per_batch=Ls, batch_stride_rows=Ljn, pos_offset=Lc, ld=3 * D, q_off=0, k_off=D, heads=H, dh=dh,
                        wq_b=wq, wk_b=wk, split=0, cos=cos, sin=sin)
def f_one():
    ops.gemm(x, w, b, one[Lc:], M=M, ldc=3 * D, c_map=ops.RowMap(Ls, Ljn), qk_rope=ops.QkRope(wq, wk, cs, Ls, Lc, 2 * D))
t2, t1 = timeit(f_two), timeit(f_one)
diff = (one.float() - two.float()).abs()
print("QKROPE_AB double", json.dumps(dict(us_two=round(t2, 1), us_fused=round(t1, 1), differing=float((diff > 0).float().mean()), max_abs=float(diff.max()),
      rel=float(diff.norm() / two.float().norm()))), flush=True)
