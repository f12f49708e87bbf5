"""Small-M GEMMs (the text stream: M = B x 512 rows): which tile kernel / split serves them best. Modes interleaved in one process
(UG_ENV_DYNAMIC=1): the dispatcher's choice, the 128^2 kernel, the 256^2 kernel, the 256^2 kernel with the split-K tail allowed from 8 K-tiles.
usage: python tools/small_gemm_ab.py"""
import os, sys
os.environ["UG_ENV_DYNAMIC"] = "1"
# the variants exist only in the probe library: python -m unigen_amd.build --probe (tools/probe/README.md)
os.environ.setdefault("UG_LIB_PATH", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "probe", "libunigen_hip_probe.so"))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unigen_amd import ops, lib as L
dev, BF = torch.device("cuda:0"), torch.bfloat16
g = torch.Generator(device=dev).manual_seed(0)
rn = lambda *s, sc=1.0: (torch.randn(*s, generator=g, device=dev) * sc).to(BF)
MODES = [("default", {}), ("128", {"UG_GEMM_FORCE_TILE": "128"}), ("256", {"UG_GEMM_FORCE_TILE": "256", "UG_GEMM_SPLITK_TAIL": "0"}),
         ("256+split8", {"UG_GEMM_FORCE_TILE": "256", "UG_GEMM_SPLITK_MIN_KT": "8"})]
KEYS = ["UG_GEMM_FORCE_TILE", "UG_GEMM_SPLITK_TAIL", "UG_GEMM_SPLITK_MIN_KT"]
SHAPES = [(2048, 9216, 3072), (2048, 3072, 3072), (2048, 12288, 3072), (2048, 3072, 12288), (2048, 3072, 4096), (1024, 9216, 3072), (1024, 3072, 3072),
          (1024, 12288, 3072), (1024, 3072, 12288), (4096, 3072, 3072), (4096, 9216, 3072), (512, 3072, 3072), (5328, 1536, 1536), (5328, 4608, 1536), (5328, 6144, 1536)]
for (M, N, K) in SHAPES:
    a, w, b = rn(M, K), rn(N, K, sc=0.03), rn(N, sc=0.1)
    outs, times = {}, {k: [] for k, _ in MODES}
    for rep in range(4):
        for name, env in MODES:
            for k in KEYS: os.environ.pop(k, None)
            os.environ.update(env)
            out = torch.empty(M, N, device=dev, dtype=BF)
            ops.gemm(a, w, b, out, M=M)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): ops.gemm(a, w, b, out, M=M)
            e1.record(); torch.cuda.synchronize()
            times[name].append(e0.elapsed_time(e1) / 10 * 1e3)
            outs[name] = out
    ref = outs["128"].float()
    line = f"{M:5d} x {N:5d} x {K:5d}: " + "  ".join(f"{n} {min(t):7.1f} us ({2.0 * M * N * K / min(t) / 1e6:6.0f} TF)" for n, t in times.items())
    err = max(float((outs[n].float() - ref).abs().max()) for n in outs)
    print(line + f"  max|diff| {err:.3g}")
