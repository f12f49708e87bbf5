"""What per-condition LoRA adapters cost inside the forward (A12 on the hot path, round 6): cfg3's model (MultiCondtionUniGenFlux, depth + canny +
openpose, 1024^2) with one rank-R adapter per condition on the control branch's attention projections (to_q / to_k / to_v / to_out.0 of the
9 + 19 control blocks: 93 projections), the 4-step loop timed with
  no adapter attached | all adapters switched off (enable_lora(modules, [])) | one condition's adapter live | all three live.
The adapter-free launches must come back bit-identical with the adapters switched off. Interleaved, same process, same inputs.
    python tools/lora_forward_bench.py [--batch 8] [--rank 16]"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unigen_amd.flux import MultiCondtionUniGenFlux
from unigen_amd.lora import enable_lora
from unigen_amd.pipeline import denoise_loop, prepare_latent_image_ids
import bench

ap = argparse.ArgumentParser(); ap.add_argument("--batch", type=int, default=8); ap.add_argument("--rank", type=int, default=16); a = ap.parse_args()
dev, BF = torch.device("cuda:0"), torch.bfloat16
B, grid, T, K = a.batch, 64, 512, 3
CT = ["depth", "canny", "openpose"]
m = MultiCondtionUniGenFlux.from_config({}, device=dev, dtype=BF)
m.init_condition_block(condition_nums=K, condition_types=CT, control_params=dict(bench.CONTROL_PARAMS))
m.init_synthetic_(seed=0, std=0.02)
g = torch.Generator(device=dev).manual_seed(12443)
rn = lambda *s: torch.randn(*s, generator=g, device=dev)
N = grid * grid
lat0 = rn(B, N, 64).to(BF)
conds, cpools = [rn(B, N, 64).to(BF) for _ in range(K)], [rn(B, 768).to(BF) for _ in range(K)]
prompt, pooled = (0.1 * rn(B, T, 4096)).to(BF), rn(B, 768).to(BF)
ids = prepare_latent_image_ids(grid, grid, dev, BF)
txt = torch.zeros(T, 3, device=dev, dtype=BF)
unis = [[torch.rand(B * N, 12, generator=g, device=dev) for _ in range(K)] for _ in range(4)]
step = lambda: denoise_loop(m, latents=lat0.clone(), control_tokens=conds, prompt_embeds=prompt, pooled_prompt_embeds=pooled, condition_pooled_prompt_embeds=cpools,
                            text_ids=txt, latent_image_ids=ids, condition_ids=[ids] * K, num_inference_steps=4, gate_uniforms=unis)


def timed(n=2):
    out = step(); torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); out = step(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return min(ts), out


res = {}
res["no_adapter_attached"], out0 = timed()
hits = []
for i, name in enumerate(CT):
    hits = m.add_lora(["attn.to_q", "attn.to_k", "attn.to_v", "attn.to_out.0"], name, a.rank, 2.0 * a.rank, prefix="control_", init_lora_weights=False, seed=i)
mods = list(m.modules())
for rep in range(2):
    with enable_lora(mods, []):
        res["all_switched_off"], out_off = timed()
    for lay in m._lora_sites.values():
        lay.unscale_layer(None)
    assert torch.equal(out_off, out0), "adapters switched off: the forward must be the adapter-free one, bit for bit"
    with enable_lora(mods, ["canny"]):
        res["one_live"], out1 = timed()
    for lay in m._lora_sites.values():
        lay.unscale_layer(None)
    res["three_live"], out3 = timed()
assert torch.isfinite(out1.float()).all() and torch.isfinite(out3.float()).all() and not torch.equal(out1, out0)
base = res["no_adapter_attached"]
print("LORA_FORWARD_BENCH", json.dumps(dict(
    workload=f"cfg3 model, 1024^2, B={B}, 4 steps; rank-{a.rank} adapters per condition on {len(hits)} control-branch attention projections",
    images_per_s={k: B / v for k, v in res.items()}, seconds_per_step=res, overhead_pct={k: 100.0 * (v / base - 1.0) for k, v in res.items()},
    adapter_flops_share_pct_three_live=None)))
