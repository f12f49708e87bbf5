"""Gradient parity at the FULL model depth and width (FLUX-schnell geometry: 19 + 38 base blocks, 9 + 19 control blocks, D = 3072, H = 24; 512^2 ->
N = 1024 image tokens, T = 512, B = 1): one training step (reference train.py:622-662: flow-matching MSE + moe loss, backward through the
control modules) on the same weights, inputs, RTS draw and target:
  truth   = torch autograd of the CPU oracle in fp32
  hip_f32 = the HIP training path (unigen_amd/training.py + autograd.py) through the fp32 verification twins
  hip_bf16 = the HIP product path (bf16 parameters)
Too heavy for the test suite (75 GB of fp32 parameters on both sides, minutes of CPU time); run it as a tool and keep the line under profiles/.
usage: python tests/fullsize_train_parity.py [--layers 19 38] | --sd3 (UniGenSD3 at SD3.5-medium depth and width, N = 1024, T = 333)"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import unigen_ref as R
from unigen_amd.flux import UniGenFlux
from unigen_amd.sd3 import UniGenSD3

ap = argparse.ArgumentParser(); ap.add_argument("--layers", type=int, nargs=2, default=None); ap.add_argument("--sd3", action="store_true"); a = ap.parse_args()
torch.set_num_threads(int(os.environ.get("UG_ORACLE_THREADS", min(16, os.cpu_count() or 16))))     # tests/conftest.py gives the children 13 of the box's 16 cores while the suite runs beside them
dev, BF = torch.device("cuda:0"), torch.bfloat16
if a.sd3:
    Model, oracle_forward, over = UniGenSD3, R.unigen_sd3_forward, {}
    CTL, COND = dict(use_shared_expert=True, use_modulate=False), ["depth"]
    cfg = R.SD3Config()
    inp = R.make_sd3_inputs(cfg, B=1, hw=64, T=333)
    t = torch.full((1,), 600.0)
    res = dict(workload="one training step of UniGenSD3 at SD3.5-medium depth and width (24 joint blocks, dual attention 0-12, transformer-block experts), N=1024, T=333, B=1")
else:
    Model, oracle_forward = UniGenFlux, R.unigen_flux_forward
    CTL = dict(use_rope=True, use_shared_expert=True, use_consis_module=False, use_single_trans_blocks=True, single_control_dev=2,
               single_block_control_method="overall_add", top_num=1, expert_num_each_condition=3)
    COND = ["canny"]
    over = dict(num_layers=a.layers[0], num_single_layers=a.layers[1]) if a.layers else {}
    cfg = R.FluxConfig(**over)
    inp = R.make_inputs(cfg, B=1, grid=32, T=512)
    t = torch.full((1,), 0.75, dtype=BF)
    res = dict(workload=f"one training step, {cfg.num_layers} + {cfg.num_single_layers} base blocks and their control blocks at full width, 512^2 (N=1024, T=512), B=1")
_target = {}


def target_like(out):
    if "t" not in _target:
        _target["t"] = torch.randn(tuple(out.shape), generator=torch.Generator().manual_seed(5))
    return _target["t"]


def step(fwd):
    out, losses, _ = fwd()
    loss = ((out.float() - target_like(out).to(out.device).float()) ** 2).reshape(out.shape[0], -1).mean(1).mean() + losses["moe_loss"]
    loss.backward()
    from unigen_amd import autograd as A_
    A_.clear_activation_cache()      # the last activation transposes (and the tensors they pin) must not outlive the step
    return float(loss)


def build(dtype, state=None):
    m = Model.from_config(dict(over), device=dev, dtype=dtype)
    m.init_condition_block(condition_nums=1, condition_types=list(COND), control_params=dict(CTL))
    if state is None:
        m.init_synthetic_(seed=0, std=0.02)
    else:
        m.load_state_dict({k: (v.to(dtype) if v.is_floating_point() else v) for k, v in state.items()})
    m.init_trainable_param()
    return m


def hip_compare(dtype, state, truth, names, live, dead, floor, tn):
    """One HIP training step; gradients are compared with the truth parameter by parameter (nothing but the truth is kept on the host)."""
    m = build(dtype, state)
    kw = {k: (v.to(dev).to(dtype) if (v.is_floating_point() and k != "gate_uniform") else v.to(dev)) for k, v in inp.items()}
    t0 = time.perf_counter()
    loss = step(lambda: m(timestep=t.to(dev), **kw))
    torch.cuda.synchronize()
    secs = round(time.perf_counter() - t0, 1)
    d2, worst, dead_zero = 0.0, (0.0, ""), True
    for n in names:
        g = m.get_parameter(n).grad
        g = g.float().cpu() if g is not None else torch.zeros(truth_shape[n])
        if n in dead:
            dead_zero = dead_zero and float(g.abs().max()) == 0.0
            continue
        e2 = float((g.double() - truth[n].double()).pow(2).sum())
        d2 += e2
        w = e2 ** 0.5 / max(float(truth[n].norm()), floor)
        if w > worst[0]:
            worst = (w, n)
    del m
    torch.cuda.empty_cache()
    return loss, d2 ** 0.5 / tn, worst, dead_zero, secs


m0 = build(BF)
state = {k: v.detach().cpu() for k, v in m0.state_dict().items()}
names = [n for n, p in m0.named_parameters() if p.requires_grad]
del m0
torch.cuda.empty_cache()
st = {k: (v.float().clone().requires_grad_(True) if k in names else (v.float() if v.is_floating_point() else v)) for k, v in state.items()}
t0 = time.perf_counter()
loss_t = step(lambda: oracle_forward(st, cfg, timestep=t, dtype=torch.float32, **inp))
res["oracle_f32_s"] = round(time.perf_counter() - t0, 1)
truth = {k: st[k].grad for k in names}
truth_shape = {k: tuple(st[k].shape) for k in names}
elements = int(sum(st[k].numel() for k in names))
del st
dead = {k for k in names if truth[k] is None or float(truth[k].abs().max()) == 0.0}
live = [k for k in names if k not in dead]
tn = sum(float(truth[k].double().pow(2).sum()) for k in live) ** 0.5
floor = 1e-3 * tn / len(names) ** 0.5
print("oracle done", loss_t, res, flush=True)
loss32, e32, worst32, dz32, res["hip_f32_s"] = hip_compare(torch.float32, state, truth, names, live, dead, floor, tn)
print("hip f32 done", loss32, e32, flush=True)
loss16, e16, worst16, dz16, res["hip_bf16_s"] = hip_compare(BF, state, truth, names, live, dead, floor, tn)
res.update(parameters_trainable=len(names), parameters_behind_discarded_outputs=len(dead), elements=elements,
           loss_oracle_f32=loss_t, loss_hip_f32=loss32, loss_hip_bf16=loss16,
           grad_rel_l2_hip_f32_vs_oracle_f32=e32, grad_worst_parameter_hip_f32=dict(name=worst32[1], rel_l2=worst32[0]),
           grad_rel_l2_hip_bf16_vs_oracle_f32=e16, dead_parameters_zero_in_hip=bool(dz32 and dz16))
print("FULLSIZE_TRAIN_PARITY", json.dumps(res))
