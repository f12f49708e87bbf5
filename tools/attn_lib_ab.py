"""Interleaved A/B of TWO BUILDS of the library on ug_flash_attn_fwd in one process (like gemm_lib_ab.py): the in-tree library ("new") against UG_LIB_B ("base").
Shapes: cfg2's joint sequence (dh 128, 24 heads, B = 4, 4608^2) and cfg5's (dh 64, 24 heads, 2 x B = 16 samples under CFG, 4429^2). Outputs are compared bit
for bit (a diagnostic build may differ on purpose: the mismatch fraction is printed, not asserted).   usage: UG_LIB_B=path python tools/attn_lib_ab.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unigen_amd import lib as L
lib_a = L.load()
lib_b = C.CDLL(os.environ["UG_LIB_B"])
for name, (res, args) in L.SIGNATURES.items():
    fn = getattr(lib_b, name); fn.restype = res; fn.argtypes = args
LIBS = [("base", lib_b), ("new", lib_a)]
from unigen_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
for label, B, H, dh, Lq in (("cfg2 joint dh128", 4, 24, 128, 4608), ("cfg5 joint dh64 (CFG batch)", 16, 24, 64, 4429), ("control img-q dh128 4096x4608", 4, 24, 128, 4096)):
    Lkv = 4608 if Lq == 4096 else Lq
    D = H * dh
    qkv = torch.randn(B, Lkv, 3 * D, generator=g, device=dev).to(torch.bfloat16)
    outs = [torch.empty(B, Lq, D, device=dev, dtype=torch.bfloat16) for _ in LIBS]
    st = (3 * D, Lkv * 3 * D)

    def run(i):
        L._lib = LIBS[i][1]
        ops.flash_attn(qkv[0, Lkv - Lq:], qkv[0, 0, D:], qkv[0, 0, 2 * D:], outs[i], batches=B, heads=H, dh=dh, Lq=Lq, Lkv=Lkv,
                       q_strides=st, k_strides=st, v_strides=st, o_strides=(D, Lq * D))
    for i in range(2):
        run(i); run(i)
    torch.cuda.synchronize()
    mism = float((outs[0] != outs[1]).float().mean())
    times = [[], []]
    for rnd in range(7):
        for i in range(2):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                run(i)
            e1.record(); e1.synchronize()
            times[i].append(e0.elapsed_time(e1) / 3)
    L._lib = lib_a
    fl = 4.0 * B * H * Lq * Lkv * dh
    med = [sorted(t)[len(t) // 2] for t in times]
    print(f"{label:32s} B{B} {Lq}x{Lkv}  " + "  ".join(f"{nm} {fl / m / 1e9:7.1f} ({(med[0] / m - 1) * 100:+.1f} %)" for (nm, _), m in zip(LIBS, med)) +
          f" TFLOP/s   mismatching elements {mism:.2e}", flush=True)
