"""Where ONE steady K-tile of the 256^2 GEMM spends its time, segment by segment: s_memtime stamps of a -DUG_DIAG_PHASES build of gemm.hip
(lane 0 of wave 0 = group 0 and wave 4 = group 1, K-tile 10 of every workgroup's second tile; 4 stamps per phase: L start, L done - reads / DMA
pieces issued and the counted vmcnt wait passed -, first barrier passed, MFMAs issued; one more behind the last barrier).

build:  python tools/gemm_phase_stamps.py --build          (hipcc -DUG_DIAG_PHASES gemm.hip, linked with the product objects into tools/probe/bin/libunigen_phases.so)
run:    UG_LIB_PATH=tools/probe/bin/libunigen_phases.so python tools/gemm_phase_stamps.py

Per shape: medians over the 256 workgroups, in shader cycles (s_memtime counts the shader clock):
  L     = L start -> L done          what the wave's own load segment costs (12 / 4 / 8 / 0 fragment reads and 0 / 2 / 2 / 4 DMA pieces in phases 0..3)
  wait  = L done -> barrier passed    how long it then waits for the partner group's MFMA segment (large = L has slack; ~0 = L is the critical path)
  M     = barrier passed -> MFMAs issued (16 MFMAs, includes the lgkmcnt waits on this phase's fragments)
  hand  = MFMAs issued -> second barrier passed (the partner's L done and its barrier)
The stamped K-tile is perturbed (each stamp is an s_memtime + lgkmcnt(0) + ds_write), so read ratios between segments, not absolute times."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if "--build" in sys.argv:
    from unigen_amd import build as B
    B.build()
    obj = os.path.join(ROOT, "tools", "probe", "obj")
    os.makedirs(obj, exist_ok=True); os.makedirs(os.path.join(ROOT, "tools", "probe", "bin"), exist_ok=True)
    o = os.path.join(obj, "gemm_phases.o")
    subprocess.run([B.HIPCC, "-DUG_DIAG_PHASES", *B.FLAGS, "-c", os.path.join(B.CSRC, "gemm.hip"), "-o", o], check=True)
    objs = [os.path.join(B.CSRC, s.replace(".hip", ".o")) for s in B.SOURCES if s != "gemm.hip"] + [o]
    out = os.path.join(ROOT, "tools", "probe", "bin", "libunigen_phases.so")
    subprocess.run([B.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out, *objs], check=True)
    print(out)
    sys.exit(0)

import torch
from unigen_amd import ops, lib as L
from unigen_amd.ops import QkRope
assert "phases" in L.LIB_PATH, "set UG_LIB_PATH to the -DUG_DIAG_PHASES build"
dev, BF = torch.device("cuda:0"), torch.bfloat16
g = torch.Generator(device=dev).manual_seed(0)
rn = lambda *s, sc=1.0: (torch.randn(*s, generator=g, device=dev) * sc).to(BF)
Bn, NI, T, D = 4, 4096, 512, 3072
SHAPES = [("ff up gelu", Bn * NI, 4 * D, D, L.EPI_BIAS_GELU), ("single qkv+mlp rope|gelu", Bn * (NI + T), 7 * D, D, "qkrope"),
          ("ff down K=12288 res_gate", Bn * NI, D, 4 * D, L.EPI_RES_GATE), ("8192^3 bias", 8192, 8192, 8192, L.EPI_BIAS)]
for label, M, N, K, epi in SHAPES:
    rows = M // Bn
    a, w, b = rn(M, K), rn(N, K, sc=0.03), rn(N, sc=0.1)
    out = torch.empty(M, N, device=dev, dtype=BF)
    kw = dict(M=M)
    if epi == "qkrope":
        cs = (torch.rand(rows, 64, 2, generator=g, device=dev) * 2 - 1).contiguous()
        kw.update(qk_rope=QkRope(rn(128) + 1, rn(128) + 1, cs, rows, 0, 2 * D, 1e-6, 128), gelu_from_n=3 * D)
    else:
        kw.update(epilogue=epi)
        if epi == L.EPI_RES_GATE:
            kw.update(residual=rn(M, N), alpha=0.5, gate=rn(Bn, N), gate_ld=N, rows_per_sample=rows)
    for _ in range(3):
        ops.gemm(a, w, b, out, **kw)
    ws = ops._gemm_workspace(dev, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); ops.gemm(a, w, b, out, **kw); e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3
    base = 4096 + 256 * 250 * 8
    raw = ws[base:base + 256 * 64 * 8].view(torch.int64).view(256, 2, 32).cpu().double()      # [workgroup][group][stamp]
    tile = ws[4096:4096 + 256 * 250 * 8].view(torch.int64).view(256, 250).cpu().double()
    ok = raw[:, 0, 16] > 0
    st = raw[ok]
    d = lambda i, j: (st[:, :, j] - st[:, :, i])
    med = lambda x: float(x.median())
    print(f"\n{label}  {M}x{N}x{K}: launch {us:.1f} us = {2.0 * M * N * K / us / 1e6:.0f} TFLOP/s; {int(ok.sum())} workgroups stamped; shader cycles")
    names = ["phase 0 (12 reads, 0 DMA)", "phase 1 (4 reads, 2 DMA)", "phase 2 (8 reads, 2 DMA)", "phase 3 (0 reads, 4 DMA)"]
    for grp in (0, 1):
        tot = med(st[:, grp, 16] - st[:, grp, 0])
        print(f"  group {grp} (wave {4 * grp}): whole K-tile {tot:.0f} cycles (16 x 4 MFMAs of 16 cycles = 1024 per wave)")
        for ph in range(4):
            o = 4 * ph
            Ls, Ld, Bp, Mi, Bn2 = (st[:, grp, o + i] for i in range(5))
            print(f"    {names[ph]:28s} L {med(Ld - Ls):6.1f}  wait {med(Bp - Ld):6.1f}  M {med(Mi - Bp):6.1f}  hand {med(Bn2 - Mi):6.1f}")
