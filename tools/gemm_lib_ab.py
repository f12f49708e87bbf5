"""Interleaved A/B of TWO BUILDS of libunigen_hip.so in one process (rule: never rank builds by timings taken on different devices):
the in-tree library vs UG_LIB_B (default tools/probe/bin/libunigen_base.so, a build of an earlier commit's gemm.hip), on the cfg2 GEMM launches
as the forward issues them (row-mapped C / R, gates per sample, the fused q/k RMSNorm + RoPE + GELU launch of the single blocks). Outputs are
compared bit for bit. usage: python tools/gemm_lib_ab.py [label substrings]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
from unigen_amd import lib as L
AB_ENV = os.environ.get("UG_AB_ENV")          # e.g. UG_AB_ENV=UG_GEMM_WALK:0,1,2 - ONE library (the probe build: UG_LIB_PATH) with NAME set to each value in turn, instead of two builds
if AB_ENV:
    os.environ["UG_ENV_DYNAMIC"] = "1"
    os.environ.setdefault("UG_LIB_PATH", os.path.join(ROOT, "tools", "probe", "libunigen_hip_probe.so"))
    import importlib; importlib.reload(L)
lib_a = L.load()
if AB_ENV:
    AB_ENV, _, vals = AB_ENV.partition(":")
    LIBS = [(v, lib_a) for v in (vals.split(",") if vals else ["0", "1"])]
else:
    path_b = os.environ.get("UG_LIB_B", os.path.join(ROOT, "tools", "probe", "bin", "libunigen_base.so"))
    lib_b = C.CDLL(path_b)
    for name, (res, args) in L.SIGNATURES.items():
        fn = getattr(lib_b, name); fn.restype = res; fn.argtypes = args
    LIBS = [(os.environ.get("UG_LIB_B_LABEL", "base"), lib_b), ("new", lib_a)]
    for kv in [x for x in os.environ.get("UG_LIBS", "").split(";") if x]:        # more builds: UG_LIBS="label=path;label=path" (tools/build_variant.py)
        lbl, pth = kv.split("=", 1)
        lx = C.CDLL(os.path.abspath(pth))
        for name, (res, args) in L.SIGNATURES.items():
            fn = getattr(lx, name); fn.restype = res; fn.argtypes = args
        LIBS.append((lbl, lx))
from unigen_amd import ops
from unigen_amd.ops import QkRope, RowMap
dev, BF = torch.device("cuda:0"), torch.bfloat16
g = torch.Generator(device=dev).manual_seed(0)
rn = lambda *s, sc=1.0: (torch.randn(*s, generator=g, device=dev) * sc).to(BF)
B, NI, T, D = 4, 4096, 512, 3072
SHAPES = [
    ("single qkv+mlp (q/k rope | v | gelu)", B * (NI + T), 7 * D, D, "qkrope"),
    ("single out K=15360 res_gate", B * (NI + T), D, 5 * D, L.EPI_RES_GATE),
    ("ff up gelu", B * NI, 4 * D, D, L.EPI_BIAS_GELU),
    ("ff down K=12288 res_gate", B * NI, D, 4 * D, L.EPI_RES_GATE),
    ("qkv image (q/k rope | v)", B * NI, 3 * D, D, "qkrope3"),
    ("attn out res_gate", B * NI, D, D, L.EPI_RES_GATE),
    ("zero-res res_scale", B * (NI + T), D, D, L.EPI_RES_SCALE),
]
if os.environ.get("UG_AB_SHAPES") == "sd3":      # cfg5's projections (UniGenSD3, D = 1536, B = 8, 1024^2): K = 1536 is 24 K-tiles per tile - the per-tile cost weighs twice what it does at K = 3072
    B, D = 8, 1536
    SHAPES = [
        ("sd3 qkv image bias", B * NI, 3 * D, D, L.EPI_BIAS),
        ("sd3 attn out res_gate", B * NI, D, D, L.EPI_RES_GATE),
        ("sd3 ff up gelu", B * NI, 4 * D, D, L.EPI_BIAS_GELU),
        ("sd3 ff down K=6144 res_gate", B * NI, D, 4 * D, L.EPI_RES_GATE),
        ("sd3 joint rows 8 x 4429 out res_gate", B * (NI + 333), D, D, L.EPI_RES_SCALE),
    ]
if os.environ.get("UG_AB_SHAPES") == "smallm":   # round 6: the launches whose tiles are cut into K-slices - batch-1 / 512^2 forms (fewer tiles than CUs) and cfg2's proj_out (remainder round)
    B = 1
    SHAPES = [
        ("text ff down 512 x 3072 x 12288 res_gate", 512, D, 4 * D, L.EPI_RES_GATE),
        ("512^2 ff down 1024 x 3072 x 12288 res_gate", 1024, D, 4 * D, L.EPI_RES_GATE),
        ("512^2 single out 1536 x 3072 x 15360 res_gate", 1536, D, 5 * D, L.EPI_RES_GATE),
        ("cfg2 single out 18432 x 3072 x 15360 res_gate", 18432, D, 5 * D, L.EPI_RES_GATE),
        ("cfg2 text ff down 2048 x 3072 x 12288 res_gate", 2048, D, 4 * D, L.EPI_RES_GATE),
    ]
if os.environ.get("UG_AB_SHAPES") == "w128":     # round 6: launches the dispatcher gives to the 128^2 kernel (4-wave against 8-wave workgroups, -DUG_GEMM128_W8=1|2)
    B = 1
    SHAPES = [
        ("512^2 attn out 1024 x 3072 x 3072 res_gate", 1024, D, D, L.EPI_RES_GATE),
        ("text attn out 512 x 3072 x 3072 res_gate", 512, D, D, L.EPI_RES_GATE),
        ("text k|v 512 x 6144 x 3072 bias", 512, 2 * D, D, L.EPI_BIAS),
        ("text qkv 512 x 9216 x 3072 bias", 512, 3 * D, D, L.EPI_BIAS),
        ("text ff up 512 x 12288 x 3072 gelu", 512, 4 * D, D, L.EPI_BIAS_GELU),
        ("512^2 zero-res 1536 x 3072 x 3072 res_scale", 1536, D, D, L.EPI_RES_SCALE),
        ("cfg2 text attn out 2048 x 3072 x 3072 res_gate", 2048, D, D, L.EPI_RES_GATE),
        ("cfg2 text qkv 2048 x 9216 x 3072 bias", 2048, 3 * D, D, L.EPI_BIAS),
        ("lora T 16384 x 64 x 3072 bias", 16384, 64, D, L.EPI_BIAS),
    ]
if os.environ.get("UG_AB_SHAPES") == "k3072":    # round 6: K = 3072 launches with a partially filled last round (or fewer tiles than CUs): does the cheaper slab round trip make their split pay?
    B = 1
    SHAPES = [
        ("cfg2 zero-res 18432 x 3072 x 3072 res_scale", 18432, D, D, L.EPI_RES_SCALE),
        ("b1 qkv image 4096 x 9216 x 3072 bias", 4096, 3 * D, D, L.EPI_BIAS),
        ("b1 ff up 4096 x 12288 x 3072 gelu", 4096, 4 * D, D, L.EPI_BIAS_GELU),
        ("b1 attn out 4096 x 3072 x 3072 res_gate", 4096, D, D, L.EPI_RES_GATE),
        ("text qkv 512 x 9216 x 3072 bias", 512, 3 * D, D, L.EPI_BIAS),
        ("text ff up 512 x 12288 x 3072 gelu", 512, 4 * D, D, L.EPI_BIAS_GELU),
        ("512^2 attn out 1024 x 3072 x 3072 res_gate", 1024, D, D, L.EPI_RES_GATE),
        ("cfg2 text ff up 2048 x 12288 x 3072 gelu", 2048, 4 * D, D, L.EPI_BIAS_GELU),
    ]
if len(sys.argv) > 1:
    SHAPES = [s for s in SHAPES if any(a in s[0] for a in sys.argv[1:])]
for label, M, N, K, epi in SHAPES:
    rows = M // B
    a, w, b = rn(M, K), rn(N, K, sc=0.03), rn(N, sc=0.1)
    res, gate = rn(M, N), rn(B, N)
    kw = dict(M=M)
    if epi in ("qkrope", "qkrope3"):
        cs = torch.rand(rows, 64, 2, generator=g, device=dev) * 2 - 1
        kw.update(qk_rope=QkRope(rn(128) + 1, rn(128) + 1, cs.contiguous(), rows, 0, 2 * D, 1e-6, 128))
        if epi == "qkrope":
            kw.update(gelu_from_n=3 * D)
    else:
        kw.update(epilogue=epi)
        if epi in (L.EPI_RES_GATE, L.EPI_RES_SCALE):
            kw.update(residual=res, alpha=0.5)
        if epi == L.EPI_RES_GATE:
            kw.update(gate=gate, gate_ld=N, rows_per_sample=rows)
    outs = [torch.empty(M, N, device=dev, dtype=BF) for _ in LIBS]

    def run(i):
        if AB_ENV:
            os.environ[AB_ENV] = LIBS[i][0]
        L._lib = LIBS[i][1]
        ops.gemm(a, w, b, outs[i], **kw)
    for i in range(len(LIBS)):
        run(i); run(i)
    torch.cuda.synchronize()
    mism = max(float((outs[0] != o).float().mean()) for o in outs[1:])
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
    fl = 2.0 * M * N * K
    med = [sorted(t)[len(t) // 2] for t in times]
    print(f"{label:40s} {M}x{N}x{K}  " + "  ".join(f"{nm} {fl / m / 1e9:7.1f} ({(med[0] / m - 1) * 100:+.1f} %)" for (nm, _), m in zip(LIBS, med)) + f" TFLOP/s   mismatching elements {mism:.2e}", flush=True)
