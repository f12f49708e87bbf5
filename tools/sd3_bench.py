"""cfg5 timing: UniGenSD3 (SD3.5-medium geometry), 1024^2 (128x128 latents -> N=4096), T=333, B=8 with CFG (16 samples per forward).
Reports seconds per forward and images/s for 28 steps (transformer + Euler only). Random-init weights, synthetic inputs."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unigen_amd.sd3 import UniGenSD3
from unigen_amd import ops

dev = torch.device("cuda:0")
B = int(os.environ.get("SD3_B", "8"))
modulated = os.environ.get("SD3_MODULATED", "0") == "1"
m = UniGenSD3.from_config({}, device=dev, dtype=torch.bfloat16)
m.init_condition_block(condition_nums=1, condition_types=["depth"], control_params=dict(use_shared_expert=True, use_modulate=modulated))
m.init_synthetic_(seed=0, std=0.02)
g = torch.Generator(device=dev).manual_seed(12443)
rn = lambda *s: torch.randn(*s, generator=g, device=dev)
Bc = 2 * B
lat, cond = rn(Bc, 16, 128, 128).bfloat16(), rn(Bc, 16, 128, 128).bfloat16()
enc = (0.1 * rn(Bc, 333, 4096)).bfloat16()
pooled, cpooled = rn(Bc, 2048).bfloat16(), rn(Bc, 2048).bfloat16()
uni = torch.rand(Bc * 4096, 6, generator=g, device=dev)
t = torch.full((Bc,), 500.0, device=dev)
run = lambda: m(hidden_states=lat, condition_hidden_states=cond, encoder_hidden_states=enc, pooled_projections=pooled, condition_pooled_projections=cpooled,
                timestep=t, gate_uniform=uni)[0]
out = run(); torch.cuda.synchronize()
assert torch.isfinite(out.float()).all()
timer = ops.KernelTimer(); ops.set_timer(timer)
t0 = time.perf_counter(); n = 3
for _ in range(n):
    out = run()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
ops.set_timer(None)
s = timer.summary()
print(json.dumps(dict(workload=f"cfg5 UniGenSD3 1024^2 B={B} (CFG -> {Bc} samples/forward), experts={'modulated' if modulated else 'transformer-block'}",
                      s_per_forward=dt, images_per_s_28_steps=B / (28 * dt),
                      gemm_tflops=s["gemm"]["flops"] / s["gemm"]["ms"] / 1e9, gemm_share=s["gemm"]["ms"] / 1e3 / (n * dt),
                      attn_tflops=s["attn"]["flops"] / s["attn"]["ms"] / 1e9, attn_share=s["attn"]["ms"] / 1e3 / (n * dt))))
