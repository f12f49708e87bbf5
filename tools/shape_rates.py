"""Per-shape GEMM / attention rates INSIDE the cfg2 forward (KernelTimer events around every launch, grouped by the launch's FLOP count):
which shapes run below the stand-alone micro-benchmark numbers, and by how much. usage: python tools/shape_rates.py [--batch 4] [--grid 64]
(--batch 1 --grid 32 = cfg1's geometry, 512^2; --batch 1 = the reference script's launch shape at 1024^2). Shapes: GEMM (M, N, K, groups, epilogue), attention (B, H, Lq, Lkv, dh)."""
import argparse, collections, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unigen_amd import ops
from unigen_amd.flux import UniGenFlux
from unigen_amd.pipeline import prepare_latent_image_ids

ap = argparse.ArgumentParser(); ap.add_argument("--batch", type=int, default=4); ap.add_argument("--grid", type=int, default=64)
ap.add_argument("--sd3", action="store_true", help="cfg5's forward: UniGenSD3 (SD3.5-medium), 1024^2, --batch images under CFG = 2 x batch samples")
a = ap.parse_args()
dev, BF = torch.device("cuda:0"), torch.bfloat16
if a.sd3:
    from unigen_amd.sd3 import UniGenSD3
    model = UniGenSD3.from_config({}, device=dev, dtype=BF)
    model.init_condition_block(condition_nums=1, condition_types=["depth"], control_params=dict(use_shared_expert=True))
    model.init_synthetic_(seed=0, std=0.02)
    g = torch.Generator(device=dev).manual_seed(5)
    rn = lambda *s: torch.randn(*s, generator=g, device=dev)
    Bc = 2 * a.batch
    inp = dict(hidden_states=rn(Bc, 16, 128, 128).to(BF), condition_hidden_states=rn(Bc, 16, 128, 128).to(BF), encoder_hidden_states=(0.1 * rn(Bc, 333, 4096)).to(BF),
               pooled_projections=rn(Bc, 2048).to(BF), condition_pooled_projections=rn(Bc, 2048).to(BF))
    t = torch.full((Bc,), 500.0, device=dev)
    uni = torch.rand(Bc * 4096, model._ctl.expert_nums, generator=g, device=dev)
    run = lambda: model(timestep=t, gate_uniform=uni, **inp)
    run(); run()
    timer = ops.KernelTimer(); ops.set_timer(timer)
    for _ in range(3):
        run()
    torch.cuda.synchronize(); ops.set_timer(None)
    groups = collections.defaultdict(list)
    for (kind, flops, e0, e1), tag in zip(timer.records, timer.tags):
        groups[(kind, tag, flops / 1e9)].append(e0.elapsed_time(e1))
    tot = sum(sum(v) for v in groups.values())
    print(f"UniGenSD3, {Bc} samples, 1024^2: sum of the timed launches: {tot / 3:.1f} ms per forward")
    print(f"{'kind':5s} {'shape':>34s} {'GFLOP':>8s} {'launches':>8s} {'avg us':>9s} {'TFLOP/s':>9s} {'share':>7s}  tiles 256^2 / 128^2")
    for (kind, tag, gf), v in sorted(groups.items(), key=lambda kv: -sum(kv[1])):
        avg = sum(v) / len(v)
        tiles = ""
        if kind == "gemm":
            M, N, K, G, epi = tag
            tiles = f"{-(-M // 256) * -(-N // 256) * G} / {-(-M // 128) * -(-N // 128) * G}"
        print(f"{kind:5s} {str(tag):>34s} {gf:8.1f} {len(v) // 3:8d} {avg * 1e3:9.1f} {gf / avg:9.1f} {sum(v) / tot:7.1%}  {tiles}")
    sys.exit(0)
model = UniGenFlux.from_config({}, device=dev, dtype=BF)
model.init_condition_block(condition_nums=1, condition_types=["canny"], control_params=dict(
    use_rope=True, use_shared_expert=True, use_consis_module=False, use_single_trans_blocks=True, single_control_dev=2,
    single_block_control_method="overall_add", top_num=1, expert_num_each_condition=3))
model.init_synthetic_(seed=0, std=0.02)
B, grid, T = a.batch, a.grid, 512
N = grid * grid
g = torch.Generator(device=dev).manual_seed(5)
rn = lambda *s: torch.randn(*s, generator=g, device=dev)
inp = dict(hidden_states=rn(B, N, 64).to(BF), condition_hidden_states=rn(B, N, 64).to(BF), encoder_hidden_states=(0.1 * rn(B, T, 4096)).to(BF),
           pooled_projections=rn(B, 768).to(BF), condition_pooled_projections=rn(B, 768).to(BF))
ids = prepare_latent_image_ids(grid, grid, dev, BF)
txt = torch.zeros(T, 3, device=dev, dtype=BF)
t = torch.full((B,), 0.75, device=dev, dtype=BF)
run = lambda: model(timestep=t, img_ids=ids, txt_ids=txt, condition_ids=ids, **inp)
run(); run()
timer = ops.KernelTimer(); ops.set_timer(timer)
for _ in range(3):
    run()
torch.cuda.synchronize(); ops.set_timer(None)
groups = collections.defaultdict(list)
for (kind, flops, e0, e1), tag in zip(timer.records, timer.tags):
    groups[(kind, tag, flops / 1e9)].append(e0.elapsed_time(e1))
tot = sum(sum(v) for v in groups.values())
print(f"B = {B}, {16 * grid}^2: sum of the timed launches: {tot / 3:.1f} ms per forward")
print(f"{'kind':5s} {'shape':>34s} {'GFLOP':>8s} {'launches':>8s} {'avg us':>9s} {'TFLOP/s':>9s} {'share':>7s}  tiles 256^2 / 128^2")
for (kind, tag, gf), v in sorted(groups.items(), key=lambda kv: -sum(kv[1])):
    avg = sum(v) / len(v)
    tiles = ""
    if kind == "gemm":
        M, N, K, G, epi = tag
        tiles = f"{-(-M // 256) * -(-N // 256) * G} / {-(-M // 128) * -(-N // 128) * G}"
    print(f"{kind:5s} {str(tag):>34s} {gf:8.1f} {len(v) // 3:8d} {avg * 1e3:9.1f} {gf / avg:9.1f} {sum(v) / tot:7.1%}  {tiles}")
