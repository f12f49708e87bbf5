"""Generates tests/golden/*.safetensors from the CPU oracle (oracle/unigen_ref.py) in THIS container.

The reference itself cannot produce vectors (not importable here: deepspeed/diffusers/peft absent; UniGenFlux uses undefined
classes) and ships none, so these fixtures pin the ORACLE (and through it the HIP path) against drift; they are not
reference-generated. Run:  python tests/golden/make_golden.py
"""
import json
import os
import sys

import torch
from safetensors.torch import save_file

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import unigen_ref as R  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
TINY = dict(num_layers=2, num_single_layers=4, attention_head_dim=128, num_attention_heads=2, joint_attention_dim=64, pooled_projection_dim=64)
CASES = {
    "flux_tiny_single": dict(n_cond=1, B=2, grid=6, T=24, state_seed=11, input_seed=12443, timestep=0.75),
    "flux_tiny_multi3": dict(n_cond=3, B=1, grid=6, T=24, state_seed=12, input_seed=12444, timestep=0.5),
}


def flatten(inp):
    out = {}
    for k, v in inp.items():
        if isinstance(v, (list, tuple)):
            for i, t in enumerate(v):
                out[f"in.{k}.{i}"] = t.contiguous()
        else:
            out[f"in.{k}"] = v.contiguous()
    return out


def main():
    torch.set_num_threads(4)
    for name, c in CASES.items():
        cfg = R.FluxConfig(condition_nums=c["n_cond"], **TINY)
        st = R.make_state(cfg, seed=c["state_seed"], std=0.05, bias_std=0.02)
        inp = R.make_inputs(cfg, B=c["B"], grid=c["grid"], T=c["T"], seed=c["input_seed"], n_cond=c["n_cond"])
        t = torch.full((c["B"],), c["timestep"], dtype=torch.bfloat16)
        out32, loss32, cnt32 = R.unigen_flux_forward(st, cfg, timestep=t, dtype=torch.float32, **inp)
        out16, loss16, cnt16 = R.unigen_flux_forward(st, cfg, timestep=t, dtype=torch.bfloat16, **inp)
        # 2-step denoise with the bf16 oracle (schedule + Euler step)
        unis = [inp["gate_uniform"], inp["gate_uniform"]]
        fwd = {k: v for k, v in inp.items() if k not in ("hidden_states", "gate_uniform")}
        lat = R.denoise(st, cfg, latents=inp["hidden_states"], num_steps=2, dtype=torch.bfloat16, gate_uniforms=unis, **fwd)
        tensors = flatten(inp)
        tensors.update({"timestep": t, "out.fp32": out32.contiguous(), "out.bf16": out16.contiguous(), "out.moe_loss": loss16["moe_loss"].reshape(1),
                        "out.expert_counts": cnt16["expert_counts"].contiguous(), "out.denoise2.bf16": lat.contiguous()})
        meta = dict(config=json.dumps(TINY), case=json.dumps(c), generator="oracle/unigen_ref.py (CPU, torch %s)" % torch.__version__)
        save_file(tensors, os.path.join(HERE, name + ".safetensors"), metadata=meta)
        print(name, {k: tuple(v.shape) for k, v in tensors.items() if k.startswith("out")})


if __name__ == "__main__":
    main()
