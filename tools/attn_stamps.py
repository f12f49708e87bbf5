"""Where an attention-forward workgroup's time goes: s_memtime stamps of a -DUG_ATTN_STAMPS build of attention.hip (kernel entry, end of the
prologue, end of the tile loop, end of the epilogue, per wave group; the CU it ran on), written into the buffer passed as the log-sum-exp output.
Build:  python tools/attn_stamps.py --build     (hipcc -DUG_ATTN_STAMPS attention.hip, linked with the product's other objects into tools/probe/bin/libunigen_attn_stamps.so)
run:    UG_LIB_PATH=tools/probe/bin/libunigen_attn_stamps.so python tools/attn_stamps.py [64|128]
Round 6: per wave group the stagger loop's time split into its four segments, summed over the tiles in the kernel (UG_SEG accumulators): Y = the
softmax segment (group B: incl. the issue of its LDS-DMAs), barrier behind Y, X = the matrix segment P.V + K.Q^T (group B: incl. its DMA wait),
barrier behind X. A wave that waits long at a barrier is waiting for the OTHER group's segment.
Per shape, medians over workgroups (group A = waves 0-3, group B = waves 4-7, one segment behind):
  prologue = entry -> first S^T segment may start      loop = the tile loop      epilogue = loop end -> O stores issued (A: incl. its trailing barrier)
  gap      = on one CU: a workgroup's entry minus the previous workgroup's last epilogue end (dispatch of the next workgroup)
  span     = first entry -> last end on a CU, against the sum of its workgroups' in-kernel times"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if "--build" in sys.argv:
    import subprocess
    from unigen_amd import build as Bd
    Bd.build()                                                      # the product objects the stamp library links against
    out = os.path.join(ROOT, "tools", "probe", "bin"); os.makedirs(out, exist_ok=True)
    obj = os.path.join(out, "attention_stamps.o")
    extra = [a_ for a_ in sys.argv if a_.startswith("-D")]          # e.g. -DUG_ATTN_AIS_64=1: the stamps of a variant
    tag = os.environ.get("UG_STAMPS_TAG", "")
    subprocess.run([Bd.HIPCC, "-DUG_ATTN_STAMPS", *extra, *Bd.FLAGS, *Bd.EXTRA["attention.hip"], "-c", os.path.join(Bd.CSRC, "attention.hip"), "-o", obj], check=True)
    objs = [obj if s_ == "attention.hip" else os.path.join(Bd.CSRC, s_.replace(".hip", ".o")) for s_ in Bd.SOURCES]
    lib = os.path.join(out, f"libunigen_attn_stamps{tag}.so")
    subprocess.run([Bd.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, *objs], check=True)
    print(lib); sys.exit(0)
import numpy as np
import torch
from unigen_amd import ops, lib as L

assert "stamps" in L.LIB_PATH, "set UG_LIB_PATH to the -DUG_ATTN_STAMPS build"
dev, BF = torch.device("cuda:0"), torch.bfloat16
g = torch.Generator(device=dev).manual_seed(0)
dh = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 128
# dh 128: the cfg2 forward's shapes (B = 4); dh 64: cfg5's (UniGenSD3, B = 8 with CFG = 16 samples: joint 4096 + 333, attn2 over the 4096 image tokens)
B, H = (4, 24) if dh == 128 else (16, 24)
D = H * dh
for Lq, Lkv in ([(4608, 4608), (4096, 4608), (8192, 8704)] if dh == 128 else [(4429, 4429), (4096, 4429), (4096, 4096)]):
    qkv = (torch.randn(B, Lkv, 3 * D, generator=g, device=dev)).to(BF)
    out = torch.empty(B, Lq, D, device=dev, dtype=BF)
    QR = int(os.environ.get("UG_STAMPS_QROWS", "256"))          # query rows per workgroup of the build under test (512: the 16-wave form)
    nwg = B * H * ((Lq + QR - 1) // QR)
    lse = torch.zeros(B, H, max(Lq, (nwg * 2 * 24 + B * H - 1) // (B * H)), device=dev, dtype=torch.float32)
    st = (3 * D, Lkv * 3 * D)
    run = lambda: ops.flash_attn(qkv[0, Lkv - Lq:], qkv[0, 0, D:], qkv[0, 0, 2 * D:], out, batches=B, heads=H, dh=dh, Lq=Lq, Lkv=Lkv, q_strides=st, k_strides=st,
                                 v_strides=st, o_strides=(D, Lq * D), lse=lse)
    for _ in range(30):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); run(); e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    raw = lse.view(-1).cpu().numpy().view(np.uint64)[: nwg * 2 * 12].reshape(nwg, 2, 12).astype(np.int64)
    seg = raw[:, :, 8:12]                                   # [wg][group][Y, barrier, X, barrier] cycles summed over the tiles
    st_ = raw[:, :, 0:4]                                    # [wg][group][stamp]
    clk = (raw[:, :, 3] - raw[:, :, 0]) / np.maximum(raw[:, :, 5] - raw[:, :, 4], 1) * 100e6       # Hz, per workgroup
    ghz = float(np.median(clk)) / 1e9
    us = lambda cyc: cyc / (ghz * 1e3)
    hw, xcc = raw[:, 0, 6] & 0xffffffff, (raw[:, 0, 6] >> 32) & 0xf
    cu = (xcc << 8) | (((hw >> 13) & 7) << 5) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 0xf)
    pro, loop, epi = st_[:, :, 1] - st_[:, :, 0], st_[:, :, 2] - st_[:, :, 1], st_[:, :, 3] - st_[:, :, 2]
    tot = st_[:, :, 3].max(1) - st_[:, :, 0].min(1)
    gaps, spans, busy, per_cu = [], [], [], []
    for c in np.unique(cu):
        w = np.nonzero(cu == c)[0]
        w = w[np.argsort(st_[w, 0, 0])]
        per_cu.append(len(w))
        ends = st_[w, :, 3].max(1)
        starts = st_[w, :, 0].min(1)
        gaps += list(starts[1:] - ends[:-1])
        spans.append(ends[-1] - starts[0]); busy.append(int(tot[w].sum()))
    med = lambda a: float(np.median(np.asarray(a)))
    # chip-wide picture on the 100 MHz real-time counter (one counter for all XCDs): when each CU enters its first / leaves its last workgroup
    rt0, rt1 = raw[:, :, 4].min(1), raw[:, :, 5].max(1)
    first = np.array([rt0[cu == c].min() for c in np.unique(cu)]) - rt0.min()
    last = np.array([rt1[cu == c].max() for c in np.unique(cu)]) - rt0.min()
    ntile = (Lkv + 63) // 64
    print(f"{Lq} x {Lkv}: launch {ms * 1e3:8.1f} us ({4.0 * B * H * Lq * Lkv * dh / ms / 1e9:7.1f} TFLOP/s), {nwg} workgroups on {len(per_cu)} CUs "
          f"({min(per_cu)}-{max(per_cu)} per CU), {ntile} key tiles, in-kernel clock {ghz:.2f} GHz")
    for gi, gn in enumerate("AB"):
        print(f"   group {gn}: prologue {us(med(pro[:, gi])):6.2f}  loop {us(med(loop[:, gi])):7.2f} ({us(med(loop[:, gi])) / ntile * 1e3:6.1f} ns per key tile)  "
              f"epilogue {us(med(epi[:, gi])):6.2f} us")
    for gi, gn in enumerate("AB"):
        tot_seg = seg[:, gi, :].sum(1)
        sh = np.median(seg[:, gi, :] / np.maximum(tot_seg[:, None], 1), axis=0)
        pt = [us(med(seg[:, gi, k])) / ntile * 1e3 for k in range(4)]
        print(f"   group {gn} per key tile: Y {pt[0]:6.1f}  wait {pt[1]:6.1f}  X {pt[2]:6.1f}  wait {pt[3]:6.1f} ns   "
              f"(shares {sh[0] * 100:4.1f} / {sh[1] * 100:4.1f} / {sh[2] * 100:4.1f} / {sh[3] * 100:4.1f} %; in cycles at {ghz:.2f} GHz: "
              f"{med(seg[:, gi, 0]) / ntile:5.0f} / {med(seg[:, gi, 1]) / ntile:5.0f} / {med(seg[:, gi, 2]) / ntile:5.0f} / {med(seg[:, gi, 3]) / ntile:5.0f})")
    print(f"   workgroup entry -> last end {us(med(tot)):7.2f} us; gap to the next workgroup on the CU: median {us(med(gaps)):5.2f}  p10 {us(np.percentile(gaps, 10)):5.2f}  "
          f"p90 {us(np.percentile(gaps, 90)):5.2f} us; CU span {us(med(spans)):8.1f} us of which in workgroups {us(med(busy)):8.1f} us "
          f"({med(busy) / med(spans) * 100:.1f} %); launch / CU span = {ms * 1e3 / us(med(spans)):.3f}")
    print(f"   real time from the first entry on the chip: CUs enter their first workgroup at median {med(first) / 100:6.1f}  p90 {np.percentile(first, 90) / 100:6.1f}  max {first.max() / 100:6.1f} us; "
          f"leave their last one at min {last.min() / 100:7.1f}  median {med(last) / 100:7.1f}  max {last.max() / 100:7.1f} us")
    per_x = [f"{int(x)}:{med(last[(np.unique(cu) >> 8) == x]) / 100:.0f}" for x in np.unique(np.unique(cu) >> 8)]
    print(f"   median last-end per XCD (us): {' '.join(per_x)}")
    wx = cu >> 8
    print("   per XCD: in-kernel clock GHz / loop us per key tile: " + "  ".join(
        f"{int(x)}: {np.median(clk[wx == x]) / 1e9:.3f} / {np.median(loop[wx == x, 0] / clk[wx == x, 0]) * 1e6 / ntile:.3f}" for x in np.unique(wx)))
