"""A12 on the hot path: per-condition LoRA adapters attached to the transformer's own projections (HipModule.add_lora), switched by the reference's
`enable_lora(list(model.modules()), [...])` (src/lora_switching_module.py:11-38), executed as the adapter K-segment of the engines' GEMM launches
(ug_gemm_desc.lora_T / lora_B) - against the CPU oracle's forward with peft 0.15's LoRA Linear (R.lora_linear) substituted at those projections.

Tolerances as tests/test_flux_gpu.py: fp32 verification twins <= 1e-3 vs the fp32 oracle (north_star); the bf16 product path no further from
the fp32 oracle than 1.25 x the oracle's own bf16 evaluation."""
import importlib

import pytest
import torch

from oracle import unigen_ref as R
from tests.test_flux_gpu import CONTROL, TINY, _to_dev
from tests.util import report, rel_l2

pytestmark = pytest.mark.gpu
BF = torch.bfloat16
SPECS = [("canny", 8, 16.0), ("depth", 4, 4.0), ("openpose", 16, 8.0)]
ATTN = ["attn.to_q", "attn.to_k", "attn.to_v", "attn.to_out.0"]


def _adapters(m):
    """per-condition adapters on the control branch's attention projections + 'canny' on a few more launch kinds: a base single block's fused
    q|k|v|proj_mlp launch and proj_out, a base joint block's context projections / to_add_out / feed-forwards, shared_expert.1's K/V-only context launch"""
    for i, (name, r, alpha) in enumerate(SPECS):
        m.add_lora(ATTN, name, r, alpha, prefix="control_", init_lora_weights=False, seed=10 + i)
    m.add_lora(["attn.to_q", "attn.to_k", "attn.to_v", "proj_mlp", "proj_out"], "canny", 8, 16.0, prefix="single_transformer_blocks.1.", init_lora_weights=False, seed=20)
    m.add_lora(["attn.add_q_proj", "attn.add_k_proj", "attn.add_v_proj", "attn.to_add_out", "ff.net.0.proj", "ff.net.2", "ff_context.net.0.proj", "ff_context.net.2"],
               "canny", 8, 16.0, prefix="transformer_blocks.0.", init_lora_weights=False, seed=21)
    m.add_lora(["attn.add_k_proj", "attn.add_v_proj", "attn.to_q"], "depth", 4, 4.0, prefix="shared_expert.1.", init_lora_weights=False, seed=22)


def _oracle_adapters(model, state):
    """{projection: [(A, B, scaling)]} of the adapters that are live on the model right now, from the model's own tensors."""
    out = {}
    for name, lay in model._lora_sites.items():
        live = [(state[f"{name}.lora_A.{a}.weight"], state[f"{name}.lora_B.{a}.weight"], lay.scaling[a]) for a in lay.live_adapters()]
        if live:
            out[name] = live
    return out


@pytest.fixture(scope="module")
def setup(gpu):
    cls = importlib.import_module("src.UniGenTransformer").MultiCondtionUniGenFlux
    ctl = dict(condition_nums=3, condition_types=["canny", "depth", "openpose"], control_params=dict(CONTROL))
    model = cls.from_config(dict(TINY), device=gpu, dtype=BF)
    model.init_condition_block(**ctl)
    model.init_synthetic_(seed=7, std=0.05, bias_std=0.02)
    rcfg = R.FluxConfig(condition_nums=3, **TINY)
    inp = R.make_inputs(rcfg, B=2, grid=8, T=32, n_cond=3)
    t = torch.full((2,), 0.75, dtype=BF)
    dinp = {k: _to_dev(v, gpu) for k, v in inp.items()}
    out0 = model(timestep=t.to(gpu), **dinp)[0].clone()
    _adapters(model)
    state = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    m32 = cls.from_config(dict(TINY), device=gpu, dtype=torch.float32)
    m32.init_condition_block(**ctl)
    _adapters(m32)
    res = m32.load_state_dict({k: v.float() for k, v in model.state_dict().items()}, strict=False)
    assert not res.missing_keys and not res.unexpected_keys, res
    return model, m32, state, rcfg, inp, dinp, t, out0


def test_no_live_adapter_is_bit_identical_to_the_adapter_free_model(setup, gpu):
    model, _, _, _, _, dinp, t, out0 = setup
    mod = importlib.import_module("src.lora_switching_module")
    with mod.enable_lora(list(model.modules()), []):                 # every adapter's scale set to 0 -> the launches of the adapter-free model
        out = model(timestep=t.to(gpu), **dinp)[0]
        assert torch.equal(out, out0)
    for lay in model._lora_sites.values():                           # undo the Q10 "restore" for the tests below (alpha != r on two adapters)
        lay.unscale_layer(None)


@pytest.mark.parametrize("enabled", [["canny"], ["depth", "openpose"]])
def test_multicondition_forward_under_enable_lora_matches_oracle(setup, gpu, enabled):
    model, m32, state, rcfg, inp, dinp, t, out0 = setup
    mod = importlib.import_module("src.lora_switching_module")
    with mod.enable_lora(list(model.modules()), enabled), mod.enable_lora(list(m32.modules()), enabled):
        ad = _oracle_adapters(model, state)
        assert ad and all(len(v) == len([a for a in enabled if f"{k}.lora_A.{a}.weight" in state]) for k, v in ad.items())
        out, _, outs = model(timestep=t.to(gpu), **dinp)
        out32 = m32(timestep=t.to(gpu), **dinp)[0]
        torch.cuda.synchronize()
    st = dict(state); st[R.LORA_KEY] = ad
    truth = R.unigen_flux_forward(st, rcfg, timestep=t, dtype=torch.float32, **inp)[0]
    ref16, _, cnt16 = R.unigen_flux_forward(st, rcfg, timestep=t, dtype=BF, **inp)
    plain = R.unigen_flux_forward(state, rcfg, timestep=t, dtype=torch.float32, **inp)[0]
    moved = rel_l2(truth, plain)
    # a dropped K-segment would put the fp32 twins `moved` away from the truth: it must sit well above their 1e-3 tolerance to be seen
    assert moved > 3e-3, f"the adapters barely change the oracle's output ({moved:.2e}): the test would not see a dropped K-segment"
    e32 = rel_l2(out32, truth)
    err_hip, err_ref = rel_l2(out, truth), rel_l2(ref16, truth)
    m = report(f"lora_forward_{'+'.join(enabled)}", out, ref16, err_f32_twins=e32, err_hip_vs_fp32=err_hip, err_oraclebf16_vs_fp32=err_ref, adapters_move_output=moved)
    assert e32 <= 1e-3, m
    assert err_hip <= 1.25 * err_ref + 1e-3, m
    assert torch.equal(outs["expert_counts"].cpu(), cnt16["expert_counts"])
    for mdl in (model, m32):
        for lay in mdl._lora_sites.values():
            lay.unscale_layer(None)


def test_restore_quirk_and_forward_scale_kwarg(setup, gpu):
    """After `enable_lora.__exit__` the scalings are saved * alpha / r (Q10); `joint_attention_kwargs={"scale": w}` multiplies them for one
    forward (scale_lora_layers / unscale_lora_layers, src/UniGenTransformer.py:1200-1208, 1266-1269). All three adapters live at once in every
    control launch (three rank blocks in one K-segment)."""
    model, m32, state, rcfg, inp, dinp, t, _ = setup
    mod = importlib.import_module("src.lora_switching_module")
    with mod.enable_lora(list(m32.modules()), ["canny"]):
        pass
    lay = m32.get_submodule("control_joint_trans_blocks.0.attn.to_q")
    assert lay.scaling == {"canny": 4.0, "depth": 1.0, "openpose": 0.25}
    st = dict(state)
    for lay_ in m32._lora_sites.values():
        lay_.scale_layer(0.5)
    st[R.LORA_KEY] = _oracle_adapters(m32, state)
    for lay_ in m32._lora_sites.values():
        lay_.unscale_layer(0.5)
    out32 = m32(timestep=t.to(gpu), joint_attention_kwargs={"scale": 0.5}, **dinp)[0]
    assert lay.scaling == {"canny": 4.0, "depth": 1.0, "openpose": 0.25}
    truth = R.unigen_flux_forward(st, rcfg, timestep=t, dtype=torch.float32, **inp)[0]
    e32 = rel_l2(out32, truth)
    report("lora_forward_quirk_scale", out32, truth, err_f32_twins=e32)
    assert e32 <= 1e-3
    for lay_ in m32._lora_sites.values():
        lay_.unscale_layer(None)


def test_lora_segment_at_full_width(gpu):
    """One control joint block at FLUX width (D = 3072, 24 heads, 1024 + 512 tokens, B = 2: the 256^2 kernel's LoRA instantiation on the fused
    q|k|v launch with a block-diagonal 3 x 16-rank segment, to_out.0 behind a row-mapped A) against the oracle's joint block with R.lora_linear."""
    from unigen_amd.engine import _Stream
    cls = importlib.import_module("src.UniGenTransformer").UniGenFlux
    cfg = dict(num_layers=1, num_single_layers=1, attention_head_dim=128, num_attention_heads=24, joint_attention_dim=64, pooled_projection_dim=64)
    model = cls.from_config(cfg, device=gpu, dtype=BF)
    model.init_synthetic_(seed=3, std=0.02, bias_std=0.02)
    p = "transformer_blocks.0"
    model.add_lora(ATTN + ["attn.add_q_proj", "attn.add_k_proj", "attn.add_v_proj", "attn.to_add_out"], "canny", 16, 32.0, prefix=p + ".", init_lora_weights=False, seed=5)
    B, N, T, D = 2, 1024, 512, 3072
    g = torch.Generator().manual_seed(11)
    x, enc, temb = torch.randn(B, N, D, generator=g).to(BF), torch.randn(B, T, D, generator=g).to(BF), torch.randn(B, D, generator=g).to(BF)
    state = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    st = dict(state); st[R.LORA_KEY] = _oracle_adapters(model, state)
    rcfg = R.FluxConfig(**cfg)
    ids = torch.cat([torch.zeros(T, 3), R.make_ids(32, 32).float()], 0)
    rope = R.flux_pos_embed(ids, rcfg.axes_dims_rope)
    with torch.no_grad():
        e_ref, x_ref = R.flux_double_block(st, p, rcfg, x.float(), enc.float(), temb.float(), rope)
        e_16, x_16 = R.flux_double_block(st, p, rcfg, x, enc, temb, rope)
        e_pl, x_pl = R.flux_double_block(state, p, rcfg, x.float(), enc.float(), temb.float(), rope)
    xs, es = x.to(gpu).view(B * N, D).clone(), enc.to(gpu).view(B * T, D).clone()
    model._emb_tab.clear()
    tab = model._rope([torch.zeros(T, 3, device=gpu), R.make_ids(32, 32).to(gpu)], None)
    with torch.no_grad():
        model._double_block(p, B, _Stream(xs, N), _Stream(xs, N), _Stream(es, T), _Stream(es, T), temb.to(gpu), tab, "t")
    torch.cuda.synchronize()
    ex, ee = rel_l2(xs.view(B, N, D), x_ref), rel_l2(es.view(B, T, D), e_ref)
    rx, re = rel_l2(x_16, x_ref), rel_l2(e_16, e_ref)
    moved = rel_l2(x_ref, x_pl)
    m = report("lora_block_full_width", xs.view(B, N, D), x_16, err_hip_vs_fp32=ex, err_oraclebf16_vs_fp32=rx, err_ctx_hip=ee, err_ctx_oraclebf16=re, adapters_move_output=moved)
    assert moved > 5 * ex, m
    assert ex <= 1.25 * rx + 1e-3 and ee <= 1.25 * re + 1e-3, m


def test_sd3_forward_with_adapters_matches_oracle(gpu):
    """UniGenSD3 (the shared `_double_block`: dual attention `attn2`, the `context_pre_only` last block, the control blocks) with adapters on its joint
    blocks' attention / feed-forward projections, switched by `enable_lora`, against the oracle's SD3 forward with R.lora_linear at those projections.
    The transformer-block experts' own projections are grouped launches over experts and are refused by add_lora."""
    from unigen_amd import lib as L
    from tests.test_sd3_gpu import TINY as SD3_TINY
    mod = importlib.import_module("src.lora_switching_module")
    cls = importlib.import_module("src.UniGenTransformer").UniGenSD3

    def build(dtype):
        m = cls.from_config(dict(SD3_TINY), device=gpu, dtype=dtype)
        m.init_condition_block(condition_nums=1, condition_types=["depth"], control_params=dict(use_shared_expert=True, use_modulate=False))      # the shipped yaml's transformer-block experts
        return m

    def attach(m):
        targets = ["attn.to_q", "attn.to_k", "attn.to_v", "attn.to_out.0", "attn.add_q_proj", "attn.add_k_proj", "attn.add_v_proj", "attn.to_add_out",
                   "attn2.to_q", "attn2.to_k", "attn2.to_v", "attn2.to_out.0", "ff.net.0.proj", "ff.net.2"]
        m.add_lora(targets, "depth", 8, 16.0, prefix="transformer_blocks.", init_lora_weights=False, seed=31)
        hits = m.add_lora(ATTN, "canny", 4, 8.0, prefix="control_", init_lora_weights=False, seed=32)
        assert hits and all(h.startswith("control_") for h in hits)

    model = build(BF)
    model.init_synthetic_(seed=5, std=0.05, bias_std=0.02)
    attach(model)
    with pytest.raises(L.UniGenHipError, match="not a projection"):
        model.add_lora(["attn.to_q"], "x", 4, 4.0, prefix="moe.")
    state = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    m32 = build(torch.float32)
    attach(m32)
    res = m32.load_state_dict({k: v.float() for k, v in model.state_dict().items()}, strict=False)
    assert not res.missing_keys and not res.unexpected_keys, res
    rcfg = R.SD3Config(use_modulate=False, **SD3_TINY)
    inp = R.make_sd3_inputs(rcfg, B=2, hw=16, T=24)
    t = torch.full((2,), 600.0)
    dinp = {k: v.to(gpu) for k, v in inp.items()}
    with mod.enable_lora(list(model.modules()), ["depth"]), mod.enable_lora(list(m32.modules()), ["depth"]):
        ad = _oracle_adapters(model, state)
        assert ad and all(k.startswith("transformer_blocks.") for k in ad)          # the control blocks' "canny" adapters are switched off
        out = model(timestep=t.to(gpu), **dinp)[0]
        out32 = m32(timestep=t.to(gpu), **dinp)[0]
        torch.cuda.synchronize()
    st = dict(state); st[R.LORA_KEY] = ad
    truth = R.unigen_sd3_forward(st, rcfg, timestep=t, dtype=torch.float32, **inp)[0]
    ref16 = R.unigen_sd3_forward(st, rcfg, timestep=t, dtype=BF, **inp)[0]
    plain = R.unigen_sd3_forward(state, rcfg, timestep=t, dtype=torch.float32, **inp)[0]
    moved, e32 = rel_l2(truth, plain), rel_l2(out32, truth)
    err_hip, err_ref = rel_l2(out, truth), rel_l2(ref16, truth)
    m = report("lora_sd3_forward", out, ref16, err_f32_twins=e32, err_hip_vs_fp32=err_hip, err_oraclebf16_vs_fp32=err_ref, adapters_move_output=moved)
    assert moved > 3e-3 and e32 <= 1e-3, m
    assert err_hip <= 1.25 * err_ref + 1e-3, m
