"""LoRA K-segment GEMM (base + T . B^T in one launch): 128^2 vs 256^2 kernel, interleaved in one process, cfg2 projection shapes with a
rank-16 adapter padded to 64 (peft Linear as the reference drives it, src/lora_switching_module.py). Also the same shape without LoRA."""
import os, sys, json
os.environ["UG_ENV_DYNAMIC"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unigen_amd import ops, lib as L

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
rn = lambda *s: torch.randn(*s, generator=g, device=dev).to(torch.bfloat16)
SHAPES = [("qkv", 16384, 9216, 3072), ("out", 16384, 3072, 3072), ("ff1", 16384, 12288, 3072), ("ff2", 16384, 3072, 12288), ("single_in", 18432, 21504, 3072)]
res = {}
for name, M, N, K in SHAPES:
    a, w, b, t, lb = rn(M, K), rn(N, K) * 0.02, rn(N), rn(M, 64), rn(N, 64) * 0.02
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    variants = [("128+lora", "128", True), ("256+lora", "256", True), ("256", "256", False)]
    times = {k: [] for k, _, _ in variants}
    for rep in range(6):
        for key, tile, lora in variants:
            os.environ["UG_GEMM_FORCE_TILE"] = tile
            kw = dict(M=M, lora_t=t, lora_b=lb) if lora else dict(M=M)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ops.gemm(a, w, b, out, **kw)
            e0.record()
            for _ in range(5): ops.gemm(a, w, b, out, **kw)
            e1.record(); torch.cuda.synchronize()
            if rep: times[key].append(e0.elapsed_time(e1) / 5)
    res[name] = {k: round(2.0 * M * N * (K + (64 if "lora" in k else 0)) / (min(v) * 1e-3) / 1e12, 1) for k, v in times.items()}
    print("LORA_BENCH", name, json.dumps(res[name]), flush=True)
