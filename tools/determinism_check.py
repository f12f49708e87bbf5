"""Run-to-run determinism of the two dominant kernels on cfg2 shapes: the same launch repeated must give bitwise identical outputs
(the GEMM's split-K tail reduces slabs in a fixed order; the relaxed vmcnt waits and the LDS ring must never race)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unigen_amd import ops, lib as L
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1)
bad = 0
for (M, N, K, epi) in [(16384, 9216, 3072, L.EPI_BIAS), (18432, 3072, 15360, L.EPI_RES_GATE), (18432, 21504, 3072, L.EPI_BIAS_GELU), (2048, 3072, 3072, L.EPI_RES_GATE)]:
    a = (torch.rand(M, K, generator=g, device=dev) * 2 - 1).to(torch.bfloat16)
    w = ((torch.rand(N, K, generator=g, device=dev) * 2 - 1) * K ** -0.5).to(torch.bfloat16)
    b = (torch.rand(N, generator=g, device=dev) * 0.1).to(torch.bfloat16)
    r = torch.rand(M, N, generator=g, device=dev).to(torch.bfloat16)
    gate = torch.rand(M // 512 + 1, N, generator=g, device=dev).to(torch.bfloat16)
    kw = dict(M=M, epilogue=epi)
    if epi == L.EPI_RES_GATE: kw.update(residual=r, gate=gate, gate_ld=N, rows_per_sample=512)
    ref = None
    for it in range(25):
        out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        ops.gemm(a, w, b, out, **kw)
        if ref is None: ref = out
        elif not torch.equal(ref, out): bad += 1; print("GEMM mismatch", M, N, K, epi, it, (ref.float() - out.float()).abs().max().item())
    print("gemm", M, N, K, epi, "ok" if bad == 0 else "MISMATCH", flush=True)
# Round 6: the split-K slabs travel at system scope without fences (UG_GEMM_SLAB_SC). A slice's partial sums read too early - or a stale line of the slab area from
# an earlier launch - would show as a run-to-run difference: 200 repeats of the sliced small-M launches, with ANOTHER sliced launch on other data in between (it
# rewrites the same slab area and tickets), every output compared bit for bit with the first one.
shapes = [(512, 3072, 12288), (1024, 3072, 12288), (1536, 3072, 15360)]
data = []
for (M, N, K) in shapes:
    a = (torch.rand(M, K, generator=g, device=dev) * 2 - 1).to(torch.bfloat16)
    w = ((torch.rand(N, K, generator=g, device=dev) * 2 - 1) * K ** -0.5).to(torch.bfloat16)
    b = (torch.rand(N, generator=g, device=dev) * 0.1).to(torch.bfloat16)
    data.append((M, N, K, a, w, b))
refs = [None] * len(data)
for it in range(200):
    for i, (M, N, K, a, w, b) in enumerate(data):
        out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        ops.gemm(a, w, b, out, M=M)
        if refs[i] is None:
            refs[i] = out
            exact = (a.float() @ w.float().t() + b.float())
            e = float((out.float() - exact).norm() / exact.norm())
            assert e < 4e-3, (M, N, K, e)
        elif not torch.equal(refs[i], out):
            bad += 1; print("split-K mismatch", M, N, K, it, (refs[i].float() - out.float()).abs().max().item())
print("gemm split-K small-M x 200 interleaved", "ok" if bad == 0 else "MISMATCH", flush=True)
H, dh = 24, 128; D = H * dh
for (B, Lq, Lkv) in [(4, 4608, 4608), (4, 4096, 4608)]:
    qkv = torch.randn(B, Lkv, 3 * D, generator=g, device=dev).to(torch.bfloat16)
    st = (3 * D, Lkv * 3 * D)
    ref = None
    for it in range(25):
        out = torch.empty(B, Lq, D, device=dev, dtype=torch.bfloat16)
        ops.flash_attn(qkv[0, Lkv - Lq:], qkv[0, 0, D:], qkv[0, 0, 2 * D:], out, batches=B, heads=H, dh=dh, Lq=Lq, Lkv=Lkv, q_strides=st, k_strides=st, v_strides=st, o_strides=(D, Lq * D))
        if ref is None: ref = out
        elif not torch.equal(ref, out): bad += 1; print("ATTN mismatch", B, Lq, Lkv, it)
    print("attn", B, Lq, Lkv, "ok" if bad == 0 else "MISMATCH", flush=True)
sys.exit(1 if bad else 0)
