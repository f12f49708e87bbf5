"""One training step (forward + backward of the control modules, reference train.py:622-662) at the FLUX-schnell geometry + canny control,
random weights and inputs: seconds per step, samples/s, peak memory. Not the headline metric (BASELINE.json measures inference); the
SURVEY 8(f) rank-4 row. usage: python tools/train_bench.py [--batch 1] [--size 1024] [--ckpt] [--layers 19 38]"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unigen_amd.flux import UniGenFlux
from unigen_amd.pipeline import prepare_latent_image_ids

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=1)
ap.add_argument("--size", type=int, default=1024)
ap.add_argument("--ckpt", action="store_true")
ap.add_argument("--layers", type=int, nargs=2, default=None)
ap.add_argument("--steps", type=int, default=3)
ap.add_argument("--sd3", action="store_true", help="UniGenSD3 (SD3.5-medium geometry, depth control, transformer-block experts) instead of UniGenFlux")
ap.add_argument("--no-gradnorm", action="store_true", help="skip the per-step gradient-norm diagnostic (hundreds of small torch kernels: keep it out of rocprofv3 kernel statistics)")
ap.add_argument("--shapes", action="store_true", help="per-shape table of the last step's GEMM / attention launches (grouped by FLOPs per launch)")
a = ap.parse_args()
dev, BF = torch.device("cuda:0"), torch.bfloat16
B = a.batch
g = torch.Generator(device=dev).manual_seed(5)
rn = lambda *s: torch.randn(*s, generator=g, device=dev)
if a.sd3:
    from unigen_amd.sd3 import UniGenSD3
    model = UniGenSD3.from_config({}, device=dev, dtype=BF)
    model.init_condition_block(condition_nums=1, condition_types=["depth"], control_params=dict(use_shared_expert=True, use_modulate=False))
    model.init_synthetic_(seed=0, std=0.02)
    model.init_trainable_param()
    if a.ckpt:
        model.enable_gradient_checkpointing()
    hw = a.size // 8
    c = model.config
    inp = dict(hidden_states=rn(B, c.in_channels, hw, hw).to(BF), condition_hidden_states=rn(B, c.in_channels, hw, hw).to(BF),
               encoder_hidden_states=(0.1 * rn(B, 333, c.joint_attention_dim)).to(BF), pooled_projections=rn(B, c.pooled_projection_dim).to(BF),
               condition_pooled_projections=rn(B, c.pooled_projection_dim).to(BF))
    t = torch.full((B,), 600.0, device=dev)
    fwd = lambda: model(timestep=t, **inp)
    target = rn(B, 16, hw, hw)
else:
    cfg = {} if a.layers is None else {"num_layers": a.layers[0], "num_single_layers": a.layers[1]}
    model = UniGenFlux.from_config(cfg, device=dev, dtype=BF)
    model.init_condition_block(condition_nums=1, condition_types=["canny"], control_params=dict(
        use_rope=True, use_shared_expert=True, use_consis_module=False, use_single_trans_blocks=True, single_control_dev=2,
        single_block_control_method="overall_add", top_num=1, expert_num_each_condition=3))
    model.init_synthetic_(seed=0, std=0.02)
    model.init_trainable_param()
    if a.ckpt:
        model.enable_gradient_checkpointing()
    grid, T = a.size // 16, 512
    N = grid * grid
    inp = dict(hidden_states=rn(B, N, 64).to(BF), condition_hidden_states=rn(B, N, 64).to(BF), encoder_hidden_states=(0.1 * rn(B, T, 4096)).to(BF),
               pooled_projections=rn(B, 768).to(BF), condition_pooled_projections=rn(B, 768).to(BF))
    ids = prepare_latent_image_ids(grid, grid, dev, BF)
    txt = torch.zeros(T, 3, device=dev, dtype=BF)
    t = torch.full((B,), 0.75, device=dev, dtype=BF)
    fwd = lambda: model(timestep=t, img_ids=ids, txt_ids=txt, condition_ids=ids, **inp)
    target = rn(B, N, 64)
n_train = sum(p.numel() for p in model.parameters() if p.requires_grad)
n_all = sum(p.numel() for p in model.parameters())
from unigen_amd import ops
times = []
timer = None
for step in range(a.steps + 1):
    if step == a.steps:            # HIP events around every GEMM / attention launch of the last step (not part of the timed minimum)
        timer = ops.KernelTimer(kinds=("gemm", "attn", "attn_bwd")); ops.set_timer(timer)
    for p in model.parameters():
        p.grad = None
    torch.cuda.synchronize(); t0 = time.time()
    out, losses, _ = fwd()
    torch.cuda.synchronize(); t1 = time.time()
    loss = ((out.float() - target) ** 2).reshape(B, -1).mean(1).mean() + losses["moe_loss"]
    loss.backward()
    from unigen_amd import autograd as A_
    A_.clear_activation_cache()      # the last activation transposes (and the tensors they pin) must not outlive the step
    torch.cuda.synchronize(); t2 = time.time()
    if step and timer is None:
        times.append((t1 - t0, t2 - t1))
    gn = float("nan") if a.no_gradnorm else float(torch.sqrt(sum((p.grad.float() ** 2).sum() for p in model.parameters() if p.grad is not None)))
    print(f"step {step}: loss {float(loss):.5f} grad-norm {gn:.4e} forward {t1 - t0:.3f}s backward {t2 - t1:.3f}s", flush=True)
ops.set_timer(None)
rates = {k: dict(launches=v["launches"], ms=round(v["ms"], 1), tflops=round(v["flops"] / v["ms"] / 1e9, 1)) for k, v in timer.summary().items()}
fw, bw = min(x[0] for x in times), min(x[1] for x in times)
print("TRAIN_BENCH", json.dumps(dict(model="UniGenSD3" if a.sd3 else "UniGenFlux", batch=B, size=a.size, layers=[model.config.num_layers, getattr(model.config, "num_single_layers", 0)], checkpointing=bool(a.ckpt),
      params_total=n_all, params_trainable=n_train, forward_s=round(fw, 3), backward_s=round(bw, 3), samples_per_s=round(B / (fw + bw), 3),
      peak_mem_gb=round(torch.cuda.max_memory_allocated() / 2 ** 30, 1), kernel_rates=rates)))

if a.shapes:
    import collections
    groups = collections.defaultdict(list)
    for kind, flops, e0, e1 in timer.records:
        groups[(kind, round(flops / 1e9))].append(e0.elapsed_time(e1))
    tot = sum(sum(v) for v in groups.values())
    print(f"{'kind':9s} {'GFLOP/launch':>13s} {'launches':>8s} {'avg us':>9s} {'TFLOP/s':>9s} {'share':>7s}")
    for (kind, gf), v in sorted(groups.items(), key=lambda kv: -sum(kv[1]))[:40]:
        avg = sum(v) / len(v)
        print(f"{kind:9s} {gf:13d} {len(v):8d} {avg * 1e3:9.1f} {gf / max(avg, 1e-9):9.1f} {sum(v) / tot:7.1%}")
