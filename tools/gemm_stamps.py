"""Where a 256^2 GEMM tile's time goes: s_memtime stamps of a -DUG_DIAG_STAMPS build of gemm.hip (five points per tile, kept in LDS, flushed
after the last tile of a workgroup). Build:  hipcc -DUG_DIAG_STAMPS ... gemm.hip  linked with the other objects into tools/probe/bin/libunigen_stamps.so
run:    UG_LIB_PATH=tools/probe/bin/libunigen_stamps.so python tools/gemm_stamps.py
Per shape (whole tiles, no split-K tail): medians over workgroups and tiles of
  wait0  = tile top -> first K-tile's operands landed      loop = K loop (and per K-tile)      next = next tile's addresses + ring request
  epi    = epilogue issue                                   gap  = this tile's epilogue end -> next tile's top (= 0: same instruction stream)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unigen_amd import ops, lib as L
from unigen_amd.ops import QkRope

assert "stamps" in L.LIB_PATH, "set UG_LIB_PATH to the -DUG_DIAG_STAMPS build"
dev, BF = torch.device("cuda:0"), torch.bfloat16
g = torch.Generator(device=dev).manual_seed(0)
rn = lambda *s, sc=1.0: (torch.randn(*s, generator=g, device=dev) * sc).to(BF)
B, NI, T, D = 4, 4096, 512, 3072
SHAPES = [("ff up gelu", B * NI, 4 * D, D, L.EPI_BIAS_GELU), ("qkv bias", B * NI, 3 * D, D, L.EPI_BIAS), ("attn out res_gate", B * NI, D, D, L.EPI_RES_GATE),
          ("single qkv+mlp rope|gelu", B * (NI + T), 7 * D, D, "qkrope"), ("zero-res", B * NI, D, D, L.EPI_RES_SCALE), ("8192^3 bias", 8192, 8192, 8192, L.EPI_BIAS),
          ("ff down K=12288 res_gate (3 rounds)", B * NI, D, 4 * D, L.EPI_RES_GATE)]
for label, M, N, K, epi in SHAPES:
    rows = M // B
    a, w, b = rn(M, K), rn(N, K, sc=0.03), rn(N, sc=0.1)
    res, gate = rn(M, N), rn(B, N)
    out = torch.empty(M, N, device=dev, dtype=BF)
    kw = dict(M=M)
    if epi == "qkrope":
        cs = (torch.rand(rows, 64, 2, generator=g, device=dev) * 2 - 1).contiguous()
        kw.update(qk_rope=QkRope(rn(128) + 1, rn(128) + 1, cs, rows, 0, 2 * D, 1e-6, 128), gelu_from_n=3 * D)
    else:
        kw.update(epilogue=epi)
        if epi in (L.EPI_RES_GATE, L.EPI_RES_SCALE):
            kw.update(residual=res, alpha=0.5)
        if epi == L.EPI_RES_GATE:
            kw.update(gate=gate, gate_ld=N, rows_per_sample=rows)
    for _ in range(3):
        ops.gemm(a, w, b, out, **kw)
    ws = ops._gemm_workspace(dev, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); ops.gemm(a, w, b, out, **kw); e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3
    raw = ws[4096:4096 + 256 * 250 * 8].view(torch.int64).view(256, 250).cpu()
    rt_in, rt_out = raw[:, 248].clone(), raw[:, 249].clone()
    raw[:, 245:] = 0
    st = raw.view(256, 50, 5)
    ok = st[:, :, 0] > 0
    ntile = ok.sum(1)
    wg = ntile > 0
    span = torch.stack([(st[i, :ntile[i], 4].max() - st[i, :ntile[i], 0].min()) for i in range(256) if wg[i]]).double()
    tick_us = float(span.median()) / us                       # ticks per microsecond (the launch is ~ one workgroup's span)
    d = lambda x: x.double()
    wait0 = d(st[..., 1] - st[..., 0])[ok]; loop = d(st[..., 2] - st[..., 1])[ok]; nxt = d(st[..., 3] - st[..., 2])[ok]; epi_t = d(st[..., 4] - st[..., 3])[ok]
    gaps = torch.cat([d(st[i, 1:ntile[i], 0] - st[i, :ntile[i] - 1, 4]) for i in range(256) if ntile[i] > 1]) if int(ntile.max()) > 1 else torch.zeros(1).double()
    med = lambda x: float(x.median()) / tick_us
    nk = K // 64
    tot = med(wait0) + med(loop) + med(nxt) + med(epi_t) + med(gaps)
    print(f"{label:38s} {M}x{N}x{K}: launch {us:7.1f} us ({2.0 * M * N * K / us / 1e6:6.1f} TFLOP/s), {int(ntile.max())} tiles per workgroup, {tick_us / 1e3:.2f} GHz | per tile: "
          f"wait0 {med(wait0):5.2f}  loop {med(loop):6.2f} ({med(loop) / nk * 1e3:5.0f} ns per K-tile)  next {med(nxt):4.2f}  epilogue {med(epi_t):5.2f}  gap {med(gaps):4.2f} us"
          f"  -> non-loop share {1 - med(loop) / tot:.1%}; first tile wait0 {float(d(st[:, 0, 1] - st[:, 0, 0])[wg].median()) / tick_us:.2f} us", flush=True)
    # per XCD (workgroup b runs on XCD b & 7): when its workgroups left the kernel on the chip-wide 100 MHz counter, and the clock they ran at
    t0 = int(rt_in[rt_in > 0].min())
    xs = []
    for x in range(8):
        sel = torch.arange(256)[(torch.arange(256) & 7) == x]
        sel = sel[wg[sel]]
        ticks = torch.stack([st[i, :ntile[i], 4].max() - st[i, 0, 0] for i in sel]).double()
        dur = (rt_out[sel] - rt_in[sel]).double() / 100.0
        xs.append(f"{x}: {float((rt_out[sel] - t0).double().median()) / 100.0:7.1f} us @ {float((ticks / dur).median()) / 1e3:.3f} GHz")
    print("      per XCD, median exit time / clock:  " + "   ".join(xs), flush=True)

