"""Interleaved A/B of the two 256^2 GEMM structures in ONE process (UG_ENV_DYNAMIC=1): the 8-wave kernel (gemm.hip) vs the one-wave-per-SIMD
kernel (gemm_pwg.hip, UG_GEMM_PWG=1) on the cfg2 projection shapes, random data; checks bitwise/near equality of the results first."""
import os, sys
os.environ["UG_ENV_DYNAMIC"] = "1"
# the variants exist only in the probe library: python -m unigen_amd.build --probe (tools/probe/README.md)
os.environ.setdefault("UG_LIB_PATH", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "probe", "libunigen_hip_probe.so"))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unigen_amd import ops, lib as L

dev = torch.device("cuda:0")
SHAPES = [
    (16384, 9216, 3072, L.EPI_BIAS, "qkv image"),
    (16384, 3072, 3072, L.EPI_RES_GATE, "attn out"),
    (16384, 12288, 3072, L.EPI_BIAS_GELU, "ff up"),
    (16384, 3072, 12288, L.EPI_RES_GATE, "ff down"),
    (18432, 21504, 3072, L.EPI_BIAS_GELU, "single qkv+mlp"),
    (18432, 3072, 15360, L.EPI_RES_GATE, "single out"),
    (18432, 3072, 3072, L.EPI_RES_SCALE, "zero-res"),
    (8192, 8192, 8192, L.EPI_BIAS, "8192^3"),
]
if len(sys.argv) > 1:
    SHAPES = [s for s in SHAPES if any(a in s[4] for a in sys.argv[1:])]
g = torch.Generator(device=dev).manual_seed(0)
VARS = ["0", "1"]
PV = os.environ.get("PWG_VARS")
if PV:
    VARS = ["0"] + PV.split(",")          # entries "mode:var", e.g. PWG_VARS=1:0,2:0,2:2
for M, N, K, epi, label in SHAPES:
    PAD = int(os.environ.get("LD_PAD", "0"))          # extra elements in the leading dimension of A and W (L2 channel-conflict probe)
    a = (torch.rand(M, K + PAD, generator=g, device=dev) * 2 - 1).to(torch.bfloat16)[:, :K]
    w = ((torch.rand(N, K + PAD, generator=g, device=dev) * 2 - 1) * 0.05).to(torch.bfloat16)[:, :K]
    b = torch.randn(N, generator=g, device=dev).to(torch.bfloat16)
    outs = [torch.empty(M, N, device=dev, dtype=torch.bfloat16) for _ in VARS]
    r = torch.randn(M, N, generator=g, device=dev).to(torch.bfloat16)
    gate = torch.randn(M // 512 + 1, N, generator=g, device=dev).to(torch.bfloat16)
    kw = dict(M=M, epilogue=epi)
    if epi in (L.EPI_RES_GATE, L.EPI_RES_SCALE):
        kw.update(residual=r, alpha=0.5)
    if epi == L.EPI_RES_GATE:
        kw.update(gate=gate, gate_ld=N, rows_per_sample=512)

    def run(i):
        os.environ["UG_GEMM_PWG"] = VARS[i][0]
        os.environ["UG_PWG_VAR"] = VARS[i][2:] or "0"
        ops.gemm(a, w, b, outs[i], **kw)
    for i in range(len(VARS)):
        run(i); run(i)
    torch.cuda.synchronize()
    diff = (outs[0].float() - outs[1].float())
    rel = float(diff.norm() / outs[0].float().norm())
    mism = float((outs[0] != outs[1]).float().mean())
    times = [[] for _ in VARS]
    for rnd in range(7):
        for i in range(len(VARS)):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                run(i)
            e1.record(); e1.synchronize()
            times[i].append(e0.elapsed_time(e1) / 3)
    fl = 2.0 * M * N * K
    msg = "  ".join(f"pwg={v}: {fl / sorted(t)[len(t) // 2] / 1e9:7.1f} (best {fl / min(t) / 1e9:7.1f})" for v, t in zip(VARS, times))
    print(f"{label:16s} {M}x{N}x{K}  {msg} TFLOP/s   rel diff {rel:.2e} mismatch {mism:.2e}", flush=True)
