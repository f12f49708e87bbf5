"""What the reference's own execution model costs on this chip: the oracle's restatement of UniGenFlux.forward (oracle/unigen_ref.py - plain PyTorch
eager, op for op what diffusers / deepspeed / the reference run: F.linear -> hipBLASLt, F.scaled_dot_product_attention, separate bf16 elementwise
ops, dense S x E x C dispatch replaced by the index form so that it fits) executed ON THE MI355X in bf16, beside the HIP engine on the same
parameters, inputs and RTS draw. Not a test and not a product path (the oracle stays test infrastructure): a measurement script, kept under tests/
because only tests/ may import oracle/.

    python tests/eager_reference_timing.py [--batch 4] [--fwd 3]

Prints per-forward times of both, their ratio, the relative difference of the two outputs, and the eager forward's kernel-time split by
torch.profiler-free HIP events around the oracle's GEMM-like and attention calls (monkey-patched timers on F.linear / F.scaled_dot_product_attention)."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F

from oracle import unigen_ref as R
from unigen_amd.flux import UniGenFlux
from unigen_amd.pipeline import prepare_latent_image_ids

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=4)
ap.add_argument("--fwd", type=int, default=3)
ap.add_argument("--only", choices=["both", "eager", "hip"], default="both")
ap.add_argument("--sdpa", choices=["default", "math", "flash", "efficient"], default="default")
a = ap.parse_args()
if a.batch > 1 and a.only != "hip":
    # Round 4, gpurun_out/r04w_eager_b2.log and r04w_eager_b2_math.log: `--only eager --batch 2` ended in "Memory access fault by GPU ... Write access to a
    # read-only page" (core dump) - in the first forward with the default SDPA backends, in the second with the math backend. The faulting op was never
    # identified: the cause is UNIDENTIFIED. What the records do establish: no ug_* entry point ran in those processes (--only eager never calls the
    # engine); the process ran this repository's oracle on the GPU under the engine's state-dict tensors. Round 6 checked what this repository
    # contributes to that process on the bounds-checked CPU build: the same forward at the same token counts over the engine's packed state-dict views under
    # a dispatch mode that inspects EVERY aten call (tests/test_oracle_cpu.py::test_eager_forward_at_the_faulting_token_count_writes_only_its_own_tensors) -
    # every in-place destination is a tensor the oracle allocated itself (never a parameter, an input or an expanded view) and every index handed to
    # index / index_put / gather / scatter lies inside its dimension. Nothing to fix was found here, so nothing changed that would justify a second launch;
    # what differs from B = 1 on the torch side is, e.g., topk / one_hot at capacity 1366 > 1024 and hipBLASLt / SDPA shapes - not examinable offline.
    # A GPU fault can reset the host's GPUs for everyone: the case is not launched again.
    sys.exit("tests/eager_reference_timing.py: the eager path is only run at --batch 1 (B >= 2 faulted the GPU in round 4 - cause unidentified - see "
             "gpurun_out/r04w_eager_b2.log, gpurun_out/r04w_eager_b2_math.log and profiles/r04w_eager_reference.log); `--only hip` takes any batch")
dev, BF = torch.device("cuda:0"), torch.bfloat16
CTL = dict(use_rope=True, use_shared_expert=True, use_consis_module=False, use_single_trans_blocks=True, single_control_dev=2,
           single_block_control_method="overall_add", top_num=1, expert_num_each_condition=3)
model = UniGenFlux.from_config({}, device=dev, dtype=BF)
model.init_condition_block(condition_nums=1, condition_types=["canny"], control_params=dict(CTL))
model.init_synthetic_(seed=0, std=0.02)
cfg = R.FluxConfig()
B, grid, T = a.batch, 64, 512
N = grid * grid
g = torch.Generator(device=dev).manual_seed(5)
rn = lambda *s: torch.randn(*s, generator=g, device=dev)
inp = dict(hidden_states=rn(B, N, 64).to(BF), condition_hidden_states=rn(B, N, 64).to(BF), encoder_hidden_states=(0.1 * rn(B, T, 4096)).to(BF),
           pooled_projections=rn(B, 768).to(BF), condition_pooled_projections=rn(B, 768).to(BF))
ids = prepare_latent_image_ids(grid, grid, dev, BF)
txt = torch.zeros(T, 3, device=dev, dtype=BF)
t = torch.full((B,), 0.75, device=dev, dtype=BF)
uni = torch.rand(B * N, cfg.expert_nums, generator=g, device=dev)

state = dict(model.state_dict())          # the engine's own parameter tensors: no copy
if a.sdpa != "default":
    torch.backends.cuda.enable_flash_sdp(a.sdpa == "flash")
    torch.backends.cuda.enable_mem_efficient_sdp(a.sdpa == "efficient")
    torch.backends.cuda.enable_math_sdp(a.sdpa == "math")


def eager():
    with torch.no_grad():
        return R.unigen_flux_forward(state, cfg, timestep=t, img_ids=ids, txt_ids=txt, condition_ids=ids, gate_uniform=uni, dtype=BF, **inp)[0]


def hip():
    with torch.no_grad():
        return model(timestep=t, img_ids=ids, txt_ids=txt, condition_ids=ids, gate_uniform=uni, **inp)[0]


def note(msg):
    print(f"[{time.strftime('%H:%M:%S')}] {msg}", flush=True)


def timed(fn, n):
    note(f"first call of {fn.__name__}")
    fn(); torch.cuda.synchronize()
    note(f"{fn.__name__} warm")
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); out = fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return min(ts), sum(ts) / len(ts), out


e_min = e_avg = h_min = h_avg = float("nan"); e_out = h_out = None
if a.only in ("both", "eager"):
    torch.set_default_device(dev)             # the oracle allocates its index / zero tensors with plain torch.* calls
    try:
        e_min, e_avg, e_out = timed(eager, a.fwd)
    finally:
        torch.set_default_device("cpu")
    note(f"eager {e_min:.3f} s")
if a.only in ("both", "hip"):
    h_min, h_avg, h_out = timed(hip, a.fwd)
    note(f"hip {h_min:.3f} s")
rel = float((h_out.float() - e_out.float()).norm() / e_out.float().norm()) if (e_out is not None and h_out is not None) else None
if a.only == "hip":
    print("EAGER_REFERENCE_TIMING", json.dumps(dict(hip_engine_s=round(h_min, 4)))); sys.exit(0)

# split of the eager forward: events around every F.linear / SDPA call (the matmul-class work the two share)
rec = {"linear": [], "sdpa": []}
_lin, _sdpa = F.linear, F.scaled_dot_product_attention


def _wrap(kind, fn):
    def w(*args, **kw):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); r = fn(*args, **kw); e1.record(); rec[kind].append((e0, e1)); return r
    return w


F.linear, F.scaled_dot_product_attention = _wrap("linear", _lin), _wrap("sdpa", _sdpa)
torch.set_default_device(dev)
try:
    t0 = time.perf_counter(); eager(); torch.cuda.synchronize(); tot = time.perf_counter() - t0
finally:
    torch.set_default_device("cpu")
    F.linear, F.scaled_dot_product_attention = _lin, _sdpa
split = {k: sum(e0.elapsed_time(e1) for e0, e1 in v) / 1e3 for k, v in rec.items()}
print("EAGER_REFERENCE_TIMING", json.dumps(dict(
    workload=f"UniGenFlux.forward, FLUX-schnell geometry + canny control, 1024^2 (N={N}, T={T}), B={B}, bf16, one forward",
    eager_pytorch_s=round(e_min, 4), eager_pytorch_avg_s=round(e_avg, 4), hip_engine_s=round(h_min, 4), hip_engine_avg_s=round(h_avg, 4),
    speedup=round(e_min / h_min, 3), rel_l2_hip_vs_eager=rel,
    eager_split_s=dict(total=round(tot, 4), f_linear=round(split["linear"], 4), sdpa=round(split["sdpa"], 4), n_linear=len(rec["linear"]), n_sdpa=len(rec["sdpa"])),
    torch=torch.__version__, sdpa_backends=dict(flash=torch.backends.cuda.flash_sdp_enabled(), mem_efficient=torch.backends.cuda.mem_efficient_sdp_enabled(),
                                                  math=torch.backends.cuda.math_sdp_enabled()))))
