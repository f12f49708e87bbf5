"""GPU parity, UniGenSD3 (SD3.5 backbone) forward against the CPU oracle at reduced size: both expert variants (the shipped yaml's
transformer-block experts with per-token AdaLN and self-attention over capacity slots, and the modulated linears), dual attention
layers, context_pre_only last block, PatchEmbed with the cropped sincos table, NCHW in/out. Tolerance as in tests/test_flux_gpu.py."""
import importlib

import pytest
import torch

from oracle import unigen_ref as R
from tests.util import report, rel_l2

pytestmark = pytest.mark.gpu
BF = torch.bfloat16
TINY = dict(sample_size=16, num_layers=3, attention_head_dim=64, num_attention_heads=2, joint_attention_dim=64, caption_projection_dim=128,
            pooled_projection_dim=64, pos_embed_max_size=12, dual_attention_layers=(0, 1))


@pytest.mark.parametrize("modulated,B,hw,T", [(False, 2, 16, 24), (True, 2, 16, 24), (False, 1, 12, 33)])
def test_sd3_forward_matches_oracle(gpu, modulated, B, hw, T):
    cls = importlib.import_module("src.UniGenTransformer").UniGenSD3
    model = cls.from_config(dict(TINY), device=gpu, dtype=BF)
    model.init_condition_block(condition_nums=1, condition_types=["depth"], control_params=dict(use_shared_expert=True, use_modulate=modulated))
    model.init_synthetic_(seed=5, std=0.05, bias_std=0.02)
    state = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    rcfg = R.SD3Config(use_modulate=modulated, **TINY)
    assert set(state) == set(R.sd3_state_shapes(rcfg))
    inp = R.make_sd3_inputs(rcfg, B=B, hw=hw, T=T)
    t = torch.full((B,), 600.0)
    truth, _, _ = R.unigen_sd3_forward(state, rcfg, timestep=t, dtype=torch.float32, **inp)
    ref16, loss16, cnt16 = R.unigen_sd3_forward(state, rcfg, timestep=t, dtype=BF, **inp)
    out, losses, outs = model(timestep=t.to(gpu), **{k: v.to(gpu) for k, v in inp.items()})
    torch.cuda.synchronize()
    assert out.shape == truth.shape == (B, 16, hw, hw) and out.dtype == BF and torch.isfinite(out.float()).all()
    err_hip, err_ref = rel_l2(out, truth), rel_l2(ref16, truth)
    m = report(f"sd3_forward_mod{int(modulated)}_B{B}_hw{hw}", out, ref16, err_hip_vs_fp32=err_hip, err_oraclebf16_vs_fp32=err_ref)
    assert torch.equal(outs["expert_counts"].cpu(), cnt16["expert_counts"]), (outs["expert_counts"], cnt16["expert_counts"])
    assert err_hip <= 1.25 * err_ref + 1e-3, m
    assert m["rel_l2"] <= 2.5e-2, m
    assert abs(float(losses["moe_loss"]) - float(loss16["moe_loss"])) <= 1e-3 * abs(float(loss16["moe_loss"]))


def test_sd3_pipeline_cfg_loop(gpu):
    """UniGenSD3Pipeline surface: CFG batch doubling + guidance combine + Euler step; compared with the same loop on the oracle."""
    cls = importlib.import_module("src.UniGenTransformer").UniGenSD3
    model = cls.from_config(dict(TINY), device=gpu, dtype=BF)
    model.init_condition_block(condition_nums=1, condition_types=["depth"], control_params=dict(use_shared_expert=True, use_modulate=True))
    model.init_synthetic_(seed=6, std=0.05, bias_std=0.02)
    state = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    rcfg = R.SD3Config(use_modulate=True, **TINY)
    inp = R.make_sd3_inputs(rcfg, B=1, hw=16, T=24)
    neg = R.make_sd3_inputs(rcfg, B=1, hw=16, T=24, seed=7)
    uni2 = torch.rand(2 * 64, rcfg.expert_nums, generator=torch.Generator().manual_seed(1))
    pipe = importlib.import_module("src.UniGenPipeline").UniGenSD3Pipeline.from_pretrained(None, transformer=model)
    steps, gs = 2, 5.0
    res = pipe(control_image=inp["condition_hidden_states"], latents=inp["hidden_states"], prompt_embeds=inp["encoder_hidden_states"],
               negative_prompt_embeds=neg["encoder_hidden_states"], pooled_prompt_embeds=inp["pooled_projections"],
               negative_pooled_prompt_embeds=neg["pooled_projections"], condition_pooled_prompt_embeds=inp["condition_pooled_projections"],
               num_inference_steps=steps, guidance_scale=gs, gate_uniforms=[uni2.to(gpu)] * steps).images
    # the same loop on the oracle
    from unigen_amd.pipeline import flow_match_sigmas, sd3_default_sigmas
    sig = flow_match_sigmas(steps, sigmas=sd3_default_sigmas(steps, 3.0), shift=3.0)      # the schedule itself is pinned in tests/test_host_cpu.py
    lat = inp["hidden_states"].clone()
    for i in range(steps):
        t = torch.full((2,), sig[i] * 1000.0)
        o = R.unigen_sd3_forward(state, rcfg, hidden_states=torch.cat([lat] * 2), condition_hidden_states=torch.cat([inp["condition_hidden_states"]] * 2),
                                 encoder_hidden_states=torch.cat([neg["encoder_hidden_states"], inp["encoder_hidden_states"]]),
                                 pooled_projections=torch.cat([neg["pooled_projections"], inp["pooled_projections"]]),
                                 condition_pooled_projections=torch.cat([inp["condition_pooled_projections"]] * 2), timestep=t, gate_uniform=uni2, dtype=BF)[0]
        u, tx = o.chunk(2)
        pred = u + gs * (tx - u)
        lat = R.euler_step(lat, pred, sig[i], sig[i + 1])
    m = report("sd3_pipeline_cfg_2steps", res, lat)
    assert m["rel_l2"] <= 3e-2, m


def test_sd3_pipeline_guidance_window_sigmas_and_callback(gpu):
    """src/UniGenPipeline.py:364-427: `conditioning_scale * controlnet_keep[i]` per step (control_guidance_start / _end), caller-given sigmas, and
    `callback_on_step_end` replacing the latents - the same loop on the oracle, step by step."""
    from unigen_amd.pipeline import control_keep, flow_match_sigmas
    cls = importlib.import_module("src.UniGenTransformer").UniGenSD3
    model = cls.from_config(dict(TINY), device=gpu, dtype=BF)
    model.init_condition_block(condition_nums=1, condition_types=["depth"], control_params=dict(use_shared_expert=True, use_modulate=True))
    model.init_synthetic_(seed=8, std=0.05, bias_std=0.02)
    with torch.no_grad():                                                      # zero-initialised in the reference: give the control path a voice
        for n_, p_ in model.named_parameters():
            if "controlnet_add" in n_:
                p_.normal_(0.0, 0.05, generator=torch.Generator(device=gpu).manual_seed(3))
    state = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    rcfg = R.SD3Config(use_modulate=True, **TINY)
    inp = R.make_sd3_inputs(rcfg, B=1, hw=16, T=24)
    uni = torch.rand(64, rcfg.expert_nums, generator=torch.Generator().manual_seed(2))
    pipe = importlib.import_module("src.UniGenPipeline").UniGenSD3Pipeline.from_pretrained(None, transformer=model)
    my_sigmas, scale = [1.0, 0.7, 0.35], [0.8]
    assert control_keep(3, 0.0, 0.5) == [1.0, 0.0, 0.0] and control_keep(4, [0.25], [1.0]) == [0.0, 1.0, 1.0, 1.0]
    seen = []

    def cb(p_, i, t, kw):
        seen.append((i, float(t), sorted(kw)))
        return {"latents": kw["latents"] * 0.5} if i == 0 else {}

    res = pipe(control_image=inp["condition_hidden_states"], latents=inp["hidden_states"], prompt_embeds=inp["encoder_hidden_states"],
               pooled_prompt_embeds=inp["pooled_projections"], condition_pooled_prompt_embeds=inp["condition_pooled_projections"],
               num_inference_steps=3, sigmas=my_sigmas, guidance_scale=1.0, conditioning_scale=scale, control_guidance_start=0.0, control_guidance_end=0.5,
               gate_uniforms=[uni.to(gpu)] * 3, callback_on_step_end=cb, callback_on_step_end_tensor_inputs=["latents", "noise_pred"]).images
    sig = flow_match_sigmas(3, sigmas=my_sigmas, shift=3.0)
    keep = control_keep(3, 0.0, 0.5)
    lat, lat_nowindow = inp["hidden_states"].clone(), None
    for i in range(3):
        fw = lambda cs: R.unigen_sd3_forward(state, rcfg, hidden_states=lat, condition_hidden_states=inp["condition_hidden_states"],
                                             encoder_hidden_states=inp["encoder_hidden_states"], pooled_projections=inp["pooled_projections"],
                                             condition_pooled_projections=inp["condition_pooled_projections"], timestep=torch.full((1,), sig[i] * 1000.0),
                                             gate_uniform=uni, conditioning_scale=cs, dtype=BF)[0]
        pred = fw(scale[0] * keep[i])
        if i == 1:
            assert rel_l2(fw(scale[0]), pred) > 1e-3          # the window matters in this setup: step 1 with the control path on differs
        lat = R.euler_step(lat, pred, sig[i], sig[i + 1])
        if i == 0:
            lat = lat * 0.5
    assert [c[0] for c in seen] == [0, 1, 2] and abs(seen[1][1] - sig[1] * 1000.0) < 1e-3 and seen[0][2] == ["latents", "noise_pred"]
    m = report("sd3_pipeline_window_callback_3steps", res, lat)
    assert m["rel_l2"] <= 3e-2, m


def test_sd3_full_depth_fp32(gpu):
    """SD3.5-medium DEPTH (24 joint blocks, dual attention in 0-12, the last block context_pre_only; 24 control blocks) at toy width through
    the fp32 verification twins against the fp32 oracle."""
    cls = importlib.import_module("src.UniGenTransformer").UniGenSD3
    cfg = dict(TINY, num_layers=24, dual_attention_layers=tuple(range(13)))
    model = cls.from_config(cfg, device=gpu, dtype=torch.float32)
    model.init_condition_block(condition_nums=1, condition_types=["depth"], control_params=dict(use_shared_expert=True, use_modulate=False))
    model.init_synthetic_(seed=9, std=0.03, bias_std=0.01)
    state = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    rcfg = R.SD3Config(use_modulate=False, **cfg)
    assert set(state) == set(R.sd3_state_shapes(rcfg))
    inp = R.make_sd3_inputs(rcfg, B=2, hw=16, T=24)
    t = torch.full((2,), 600.0)
    truth, _, cnt = R.unigen_sd3_forward(state, rcfg, timestep=t, dtype=torch.float32, **inp)
    out, _, outs = model(timestep=t.to(gpu), **{k: (v.to(gpu).float() if (v.is_floating_point() and k != "gate_uniform") else v.to(gpu)) for k, v in inp.items()})
    m = report("sd3_full_depth_f32", out, truth)
    assert m["rel_l2"] <= 1e-3 and torch.equal(outs["expert_counts"].cpu(), cnt["expert_counts"]), m
