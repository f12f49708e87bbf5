"""Interleaved A/B of SEVERAL BUILDS of the library on ug_flash_attn_fwd in one process (like gemm_lib_ab.py): the in-tree library ("product") against the
builds named in UG_LIBS="label=path;label=path" (tools/build_variant.py) - or UG_LIB_B=path (one build, label UG_LIB_B_LABEL).
Shapes: cfg2's joint sequence (dh 128, 24 heads, B = 4, 4608^2), cfg5's (dh 64, 24 heads, 2 x B = 16 samples under CFG, 4429^2; attn2: 4096^2) and the
control blocks' image-query form. Per build: TFLOP/s (median of 7 interleaved rounds of 3 launches), the fraction of output elements that differ
from the product build's, and the relative L2 error against an fp32 evaluation of the same attention (two heads of sample 0) - a build whose bits
differ on purpose (row sums of the rounded probabilities) must be no further from the truth than the product.
    usage: UG_LIBS="ls=tools/probe/bin/libunigen_ls.so;..." python tools/attn_lib_ab.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unigen_amd import lib as L
lib_a = L.load()
specs = [kv.split("=", 1) for kv in os.environ.get("UG_LIBS", "").split(";") if kv]
if "UG_LIB_B" in os.environ:
    specs.append([os.environ.get("UG_LIB_B_LABEL", "base"), os.environ["UG_LIB_B"]])
LIBS = [("product", lib_a)]
for label, path in specs:
    lb = C.CDLL(os.path.abspath(path))
    for name, (res, args) in L.SIGNATURES.items():
        fn = getattr(lb, name); fn.restype = res; fn.argtypes = args
    LIBS.append((label, lb))
from unigen_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
for label, B, H, dh, Lq, Lkv in (("cfg2 joint dh128", 4, 24, 128, 4608, 4608), ("control img-q dh128", 4, 24, 128, 4096, 4608),
                                 ("cfg5 joint dh64 (CFG batch)", 16, 24, 64, 4429, 4429), ("cfg5 attn2 dh64", 16, 24, 64, 4096, 4096)):
    D = H * dh
    qkv = torch.randn(B, Lkv, 3 * D, generator=g, device=dev).to(torch.bfloat16)
    outs = [torch.empty(B, Lq, D, device=dev, dtype=torch.bfloat16) for _ in LIBS]
    st = (3 * D, Lkv * 3 * D)

    def run(i):
        L._lib = LIBS[i][1]
        ops.flash_attn(qkv[0, Lkv - Lq:], qkv[0, 0, D:], qkv[0, 0, 2 * D:], outs[i], batches=B, heads=H, dh=dh, Lq=Lq, Lkv=Lkv,
                       q_strides=st, k_strides=st, v_strides=st, o_strides=(D, Lq * D))
    for i in range(len(LIBS)):
        run(i); run(i)
    torch.cuda.synchronize()
    # fp32 truth for heads 0 and H - 1 of sample 0
    errs = []
    for i in range(len(LIBS)):
        e2 = n2 = 0.0
        for h in (0, H - 1):
            q = qkv[0, Lkv - Lq:, h * dh:(h + 1) * dh].float(); k = qkv[0, :, D + h * dh:D + (h + 1) * dh].float(); v = qkv[0, :, 2 * D + h * dh:2 * D + (h + 1) * dh].float()
            ref = torch.softmax(q @ k.t() / dh ** 0.5, dim=-1) @ v
            e2 += float((outs[i][0, :, h * dh:(h + 1) * dh].float() - ref).pow(2).sum()); n2 += float(ref.pow(2).sum())
        errs.append((e2 / n2) ** 0.5)
    mism = [float((outs[i] != outs[0]).float().mean()) for i in range(len(LIBS))]
    times = [[] for _ in LIBS]
    for rnd in range(7):
        for i in range(len(LIBS)):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                run(i)
            e1.record(); e1.synchronize()
            times[i].append(e0.elapsed_time(e1) / 3)
    L._lib = lib_a
    fl = 4.0 * B * H * Lq * Lkv * dh
    med = [sorted(t)[len(t) // 2] for t in times]
    print(f"{label:30s} B{B} {Lq}x{Lkv}", flush=True)
    for (nm, _), m, mm, er in zip(LIBS, med, mism, errs):
        print(f"      {nm:10s} {fl / m / 1e9:7.1f} TFLOP/s ({(med[0] / m - 1) * 100:+5.1f} %)   differs from product in {mm:.2e} of the elements   relL2 vs fp32 {er:.3e}", flush=True)
