"""cfg3 timing: MultiCondtionUniGenFlux depth+canny+openpose (E=12), 1024^2, B=8, 4 steps, 1 GPU."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unigen_amd.flux import MultiCondtionUniGenFlux
from unigen_amd.pipeline import denoise_loop, prepare_latent_image_ids
import bench

dev = torch.device("cuda:0")
B, grid, T, K = int(os.environ.get("CFG3_B", "8")), 64, 512, 3
m = MultiCondtionUniGenFlux.from_config({}, device=dev, dtype=torch.bfloat16)
m.init_condition_block(condition_nums=K, condition_types=["depth", "canny", "openpose"], control_params=dict(bench.CONTROL_PARAMS))
m.init_synthetic_(seed=0, std=0.02)
g = torch.Generator(device=dev).manual_seed(12443)
rn = lambda *s: torch.randn(*s, generator=g, device=dev)
N = grid * grid
lat0 = rn(B, N, 64).bfloat16()
conds = [rn(B, N, 64).bfloat16() for _ in range(K)]
cpools = [rn(B, 768).bfloat16() for _ in range(K)]
prompt, pooled = (0.1 * rn(B, T, 4096)).bfloat16(), rn(B, 768).bfloat16()
ids = prepare_latent_image_ids(grid, grid, dev, torch.bfloat16)
txt = torch.zeros(T, 3, device=dev, dtype=torch.bfloat16)
unis = [[torch.rand(B * N, 12, generator=g, device=dev) for _ in range(K)] for _ in range(4)]
step = lambda: denoise_loop(m, latents=lat0.clone(), control_tokens=conds, prompt_embeds=prompt, pooled_prompt_embeds=pooled,
                            condition_pooled_prompt_embeds=cpools, text_ids=txt, latent_image_ids=ids, condition_ids=[ids] * K,
                            num_inference_steps=4, gate_uniforms=unis)
out = step(); torch.cuda.synchronize(); assert torch.isfinite(out.float()).all()
t0 = time.perf_counter(); out = step(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
fl = 4 * bench.canonical_flops_per_forward(3072, N, T, 19, 38, 9, 19, K)
print(json.dumps(dict(workload=f"cfg3 MultiCondtionUniGenFlux depth+canny+openpose E=12, 1024^2, B={B}, 4 steps", images_per_s=B / dt, s_per_step=dt,
                      flops_per_image=fl, e2e_mfma_frac=B / dt * fl / 2.5e15, mem_gb=torch.cuda.max_memory_allocated() / 2**30)))
