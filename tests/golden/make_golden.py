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


SD3_CASES = {
    # the shipped yaml's transformer-block experts (use_modulate False) and the modulated-linear experts
    "sd3_tiny_blocks": dict(modulated=False, B=2, hw=16, T=24, state_seed=13, input_seed=12445, timestep=600.0),
    "sd3_tiny_modulated": dict(modulated=True, B=1, hw=12, T=33, state_seed=14, input_seed=12446, timestep=250.0),
}


def sd3_goldens():
    from tests.block_cases import SD3_TINY
    for name, c in SD3_CASES.items():
        cfg = R.SD3Config(use_modulate=c["modulated"], **SD3_TINY)
        st = R.make_sd3_state(cfg, seed=c["state_seed"], std=0.05, bias_std=0.02)
        inp = R.make_sd3_inputs(cfg, B=c["B"], hw=c["hw"], T=c["T"], seed=c["input_seed"])
        t = torch.full((c["B"],), c["timestep"])
        out32, _, _ = R.unigen_sd3_forward(st, cfg, timestep=t, dtype=torch.float32, **inp)
        out16, loss16, cnt16 = R.unigen_sd3_forward(st, cfg, timestep=t, dtype=torch.bfloat16, **inp)
        tensors = flatten(inp)
        tensors.update({"timestep": t, "out.fp32": out32.contiguous(), "out.bf16": out16.contiguous(), "out.moe_loss": loss16["moe_loss"].reshape(1),
                        "out.expert_counts": cnt16["expert_counts"].contiguous()})
        meta = dict(config=json.dumps(SD3_TINY), case=json.dumps(c), generator="oracle/unigen_ref.py (CPU, torch %s)" % torch.__version__)
        save_file(tensors, os.path.join(HERE, name + ".safetensors"), metadata=meta)
        print(name, {k: tuple(v.shape) for k, v in tensors.items() if k.startswith("out")})


def block_goldens():
    """Teacher-forced block fixtures (tests/block_cases.py): O(1) inputs of every block type and the oracle's bf16 / fp32 outputs."""
    from tests import block_cases as BC
    cfg = R.FluxConfig(condition_nums=1, **BC.FLUX_TINY)
    st = R.make_state(cfg, seed=BC.FLUX_CASE["state_seed"], std=BC.STD, bias_std=BC.BIAS_STD)
    inp = BC.flux_inputs(R, cfg)
    tensors = {f"in.{k}": v.contiguous() for k, v in inp.items()}
    for tag, dt in (("bf16", torch.bfloat16), ("fp32", torch.float32)):
        tensors.update({f"out.{tag}.{k}": v.contiguous() for k, v in BC.flux_oracle(R, st, cfg, inp, dt).items()})
    save_file(tensors, os.path.join(HERE, "blocks_flux_tiny.safetensors"),
              metadata=dict(config=json.dumps(BC.FLUX_TINY), case=json.dumps(BC.FLUX_CASE), generator="oracle/unigen_ref.py via tests/block_cases.py"))
    print("blocks_flux_tiny", sorted(k for k in tensors if k.startswith("out.bf16")))
    for name, modulated in (("blocks_sd3_tiny", False), ("blocks_sd3_tiny_modulated", True)):
        scfg = R.SD3Config(use_modulate=modulated, **BC.SD3_TINY)
        sst = R.make_sd3_state(scfg, seed=BC.SD3_CASE["state_seed"], std=BC.STD, bias_std=BC.BIAS_STD)
        sinp = BC.sd3_inputs(R, scfg)
        tensors = {f"in.{k}": v.contiguous() for k, v in sinp.items()}
        for tag, dt in (("bf16", torch.bfloat16), ("fp32", torch.float32)):
            o = BC.sd3_oracle(R, sst, scfg, sinp, dt)
            if modulated:
                o = {k: v for k, v in o.items() if k.startswith("sd3_comoe")}      # the joint blocks do not depend on the expert variant
            tensors.update({f"out.{tag}.{k}": v.contiguous() for k, v in o.items()})
        save_file(tensors, os.path.join(HERE, name + ".safetensors"),
                  metadata=dict(config=json.dumps(BC.SD3_TINY), case=json.dumps(dict(BC.SD3_CASE, modulated=modulated)),
                                generator="oracle/unigen_ref.py via tests/block_cases.py"))
        print(name, sorted(k for k in tensors if k.startswith("out.bf16")))


def vae_golden():
    """AutoencoderKL fixture (oracle/vae_ref.py): tiny config, image -> scaled latent sample, latents -> image, bf16 and fp32."""
    from oracle import vae_ref as V
    cfg_d = dict(block_out_channels=(64, 128), layers_per_block=1, norm_num_groups=32)
    case = dict(state_seed=31, input_seed=12447, B=2, H=32, W=32)
    cfg = V.VAEConfig(**cfg_d)
    st = V.make_vae_state(cfg, seed=case["state_seed"])
    g = torch.Generator().manual_seed(case["input_seed"])
    img = (torch.rand(case["B"], 3, case["H"], case["W"], generator=g) * 2 - 1).to(torch.bfloat16)
    noise = torch.randn(case["B"], 16, case["H"] // 2, case["W"] // 2, generator=g).to(torch.bfloat16)
    lat = torch.randn(case["B"], 16, 8, 8, generator=g).to(torch.bfloat16)
    tensors = {"in.image": img, "in.noise": noise, "in.latents": lat}
    for tag, dt in (("bf16", torch.bfloat16), ("fp32", torch.float32)):
        tensors[f"out.{tag}.z"] = V.encode_condition(st, cfg, img, noise, dt).contiguous()
        tensors[f"out.{tag}.image"] = V.decode_latents(st, cfg, lat, dt).contiguous()
    save_file(tensors, os.path.join(HERE, "vae_tiny.safetensors"), metadata=dict(config=json.dumps(cfg_d), case=json.dumps(case), generator="oracle/vae_ref.py"))
    print("vae_tiny", {k: tuple(v.shape) for k, v in tensors.items()})


TRAIN_CTL = dict(use_rope=True, use_shared_expert=True, use_single_trans_blocks=True, single_control_dev=2, single_block_control_method="overall_add")


def train_golden():
    """One training step (train.py:622-662) of the tiny FLUX configuration on the oracle under autograd, fp32: the loss and, per trainable
    parameter, the gradient's L2 norm and its first 8 entries - small enough to commit, enough to pin the oracle's backward and the HIP backward."""
    cfg_d = dict(TINY, num_single_layers=2)
    cfg = R.FluxConfig(condition_nums=1, **cfg_d)
    case = dict(B=2, grid=8, T=64, state_seed=21, input_seed=5, timestep=0.75, target_seed=9)
    st = R.make_state(cfg, seed=case["state_seed"], std=0.05, bias_std=0.02, dtype=torch.float32)
    trainable = sorted(k for k in st if k.split(".")[0] in ("control_time_text_embed", "control_condition_embed", "control_context_embedder", "control_x_embedder",
                                                           "control_joint_trans_blocks", "controlnet_add_joint_blocks", "moe", "control_single_trans_blocks",
                                                           "controlnet_add_single_blocks", "shared_expert"))
    for k in trainable:
        st[k] = st[k].clone().requires_grad_(True)
    inp = R.make_inputs(cfg, B=case["B"], grid=case["grid"], T=case["T"], seed=case["input_seed"])
    t = torch.full((case["B"],), case["timestep"], dtype=torch.bfloat16)
    target = torch.randn(case["B"], case["grid"] ** 2, 64, generator=torch.Generator().manual_seed(case["target_seed"]))
    out, losses, _ = R.unigen_flux_forward(st, cfg, timestep=t, dtype=torch.float32, **inp)
    loss = ((out - target) ** 2).reshape(case["B"], -1).mean(1).mean() + losses["moe_loss"]
    loss.backward()
    tensors = flatten(inp)
    tensors.update({"timestep": t, "target": target, "out.loss": loss.detach().reshape(1)})
    names = []
    for k in trainable:
        g = st[k].grad if st[k].grad is not None else torch.zeros_like(st[k])
        names.append(k)
        tensors["grad." + k] = torch.cat([g.norm().reshape(1), g.flatten()[:8].clone(), torch.zeros(max(0, 8 - g.numel()))])[:9].contiguous()
    meta = dict(config=json.dumps(cfg_d), case=json.dumps(case), trainable=json.dumps(names), generator="oracle/unigen_ref.py under torch autograd (CPU, fp32)")
    save_file(tensors, os.path.join(HERE, "train_flux_tiny.safetensors"), metadata=meta)
    print("train_flux_tiny", len(names), "parameters, loss", float(loss))


def main():
    torch.set_num_threads(4)
    if "--train-only" in sys.argv:
        return train_golden()
    if "--vae-only" in sys.argv:
        return vae_golden()
    if "--only-new" not in sys.argv:
        flux_goldens()
    sd3_goldens()
    block_goldens()
    vae_golden()
    train_golden()


def flux_goldens():
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
