"""Backward of the control modules (reference train.py:622-662): one training step's loss and the gradients of every
parameter of `trainable_control_modules` from the HIP path (unigen_amd/training.py + autograd.py) against torch autograd of the CPU oracle on the
same weights, inputs, RTS draw and target. fp32 parameters run the `_f32` verification twins: gradients must agree to 1e-3 (measured ~1e-5).
bf16: the HIP gradients must be as close to the fp32 gradients as the oracle's own bf16 autograd."""
import pytest
import torch

from oracle import unigen_ref as R
from tests.util import report

pytestmark = pytest.mark.gpu
BF = torch.bfloat16

TINY = dict(num_layers=2, num_single_layers=2, attention_head_dim=128, num_attention_heads=2, joint_attention_dim=64, pooled_projection_dim=64)
CONTROL = dict(use_rope=True, use_shared_expert=True, use_single_trans_blocks=True, single_control_dev=2, single_block_control_method="overall_add",
               top_num=1, expert_num_each_condition=3)


def _dev(v, gpu, dt=None):
    if v is None:
        return None
    if isinstance(v, (list, tuple)):
        return [_dev(t, gpu, dt) for t in v]
    return v.to(gpu) if (dt is None or not v.is_floating_point()) else v.to(gpu).to(dt)


def _step(fwd, target, dtype):
    out, losses, extra = fwd()
    flow = ((out.float() - target.to(out.device).float()) ** 2).reshape(out.shape[0], -1).mean(1)       # train.py:644-650 with weighting = 1
    loss = flow.mean() + losses["moe_loss"]
    loss.backward()
    return out.detach(), float(loss), extra


@pytest.mark.parametrize("n_cond,cls_name,top_num", [(1, "UniGenFlux", 1), (2, "MultiCondtionUniGenFlux", 1), (1, "UniGenFlux", 2), (1, "UniGenFlux", 3)])
def test_control_module_gradients_match_oracle_autograd(gpu, n_cond, cls_name, top_num):
    """top_num = 2: deepspeed top2gating - the gradient reaches the gate through BOTH kept probabilities and their normalising sum. top_num = 3:
    topkgating (no random draw) - through all kept probabilities of a token and their sum."""
    import importlib
    cls = getattr(importlib.import_module("src.UniGenTransformer"), cls_name)
    B, grid, T = 2, 8, 64                     # N = 64 image tokens, every joint length a multiple of 64 (attention backward contraction lengths)
    CONTROL = dict(globals()["CONTROL"], top_num=top_num)
    rcfg = R.FluxConfig(condition_nums=n_cond, top_num=top_num, **TINY)
    base = cls.from_config(dict(TINY), device=gpu, dtype=BF)
    base.init_condition_block(condition_nums=n_cond, condition_types=["canny", "depth"][:n_cond], control_params=dict(CONTROL))
    base.init_synthetic_(seed=3, std=0.05, bias_std=0.02)
    state = {k: v.detach().cpu() for k, v in base.state_dict().items()}
    inp = R.make_inputs(rcfg, B=B, grid=grid, T=T, n_cond=n_cond)
    if top_num == 2:                          # the gate's random draw is the Gumbel(0, 1) sample of the second choice
        u = torch.rand(B * grid * grid, rcfg.expert_nums, generator=torch.Generator().manual_seed(8)).clamp_(1e-7, 1 - 1e-7)
        inp["gate_uniform"] = -torch.log(-torch.log(u))
    if top_num > 2:
        inp["gate_uniform"] = None
    t = torch.full((B,), 0.75, dtype=BF)
    target = torch.randn(B, grid * grid, 64, generator=torch.Generator().manual_seed(5))
    base.init_trainable_param()
    names = [n for n, p in base.named_parameters() if p.requires_grad]
    assert any(n.startswith("control_joint_trans_blocks.") for n in names) and not any(n.startswith("transformer_blocks.") for n in names)

    def oracle_grads(dtype):
        st = {k: (v.to(dtype).clone().requires_grad_(True) if k in names else v.to(dtype)) for k, v in state.items()}
        out, loss, _ = _step(lambda: R.unigen_flux_forward(st, rcfg, timestep=t, dtype=dtype, **inp), target, dtype)
        return out, loss, {k: st[k].grad for k in names}

    def hip_grads(dtype):
        model = cls.from_config(dict(TINY), device=gpu, dtype=dtype)
        model.init_condition_block(condition_nums=n_cond, condition_types=["canny", "depth"][:n_cond], control_params=dict(CONTROL))
        model.load_state_dict({k: v.to(dtype) for k, v in state.items()})
        model.init_trainable_param()
        kw = {k: _dev(v, gpu, dtype if k != "gate_uniform" and not k.endswith("_ids") else None) for k, v in inp.items()}
        out, loss, extra = _step(lambda: model(timestep=t.to(gpu), **kw), target, dtype)
        return out, loss, {k: model.get_parameter(k).grad for k in names}, extra

    truth_out, truth_loss, truth = oracle_grads(torch.float32)
    # Parameters that only feed a DISCARDED context output (to_add_out / ff_context / add_q_proj ... of the control joint blocks and of
    # shared_expert[1], src/UniGenTransformer.py:1097,1022) get no gradient, or an all-zero one, in the reference too: compared as zeros.
    dead = {k for k in names if truth[k] is None or float(truth[k].abs().max()) == 0.0}
    live = [k for k in names if k not in dead]
    assert len(live) > len(names) // 2 and all(torch.isfinite(truth[k]).all() for k in live)
    z = lambda d, k: (d[k].detach().float().cpu() if d[k] is not None else torch.zeros(state[k].shape))
    cat = lambda d: torch.cat([z(d, k).flatten() for k in names])
    rel = lambda a, b: float((a - b).norm() / b.norm())
    # fp32 verification path
    out32, loss32, g32, _ = hip_grads(torch.float32)
    m = report(f"train_{cls_name}_k{top_num}_f32_forward", out32, truth_out)
    e_all = rel(cat(g32), cat(truth))
    floor = 1e-3 * float(cat(truth).norm()) / len(names) ** 0.5          # gradients that are themselves rounding noise (e.g. a key bias) do not count
    worst = max((float((z(g32, k) - truth[k]).norm() / max(float(truth[k].norm()), floor)), k) for k in live)
    assert all(float(z(g32, k).abs().max()) == 0.0 for k in dead), "a parameter behind a discarded output received a gradient"
    print(f"training fp32: loss {loss32:.6f} vs {truth_loss:.6f}; all gradients rel_l2 {e_all:.3e}; worst parameter {worst[1]} {worst[0]:.3e}")
    assert m["rel_l2"] <= 1e-4 and abs(loss32 - truth_loss) <= 1e-5 * abs(truth_loss) + 1e-7 and e_all <= 1e-3 and worst[0] <= 5e-3, (m, e_all, worst)
    # bf16 product path vs the oracle's own bf16 autograd
    ref_out, ref_loss, gref = oracle_grads(BF)
    out16, loss16, g16, extra = hip_grads(BF)
    e_hip, e_ref = rel(cat(g16), cat(truth)), rel(cat(gref), cat(truth))
    print(f"training bf16: loss {loss16:.5f} (oracle bf16 {ref_loss:.5f}, fp32 {truth_loss:.5f}); gradients vs fp32: hip {e_hip:.3e}, oracle bf16 {e_ref:.3e}")
    report(f"train_{cls_name}_k{top_num}_bf16_grads", cat(g16), cat(truth), err_hip_vs_fp32=e_hip, err_oraclebf16_vs_fp32=e_ref)
    assert e_hip <= 1.5 * e_ref + 5e-3 and abs(loss16 - truth_loss) <= 3e-2 * abs(truth_loss), (e_hip, e_ref, loss16, truth_loss)
    assert int(extra["expert_counts"].sum()) == top_num * B * grid * grid


SD3_TINY = dict(sample_size=16, num_layers=3, attention_head_dim=64, num_attention_heads=2, joint_attention_dim=64, caption_projection_dim=128,
                pooled_projection_dim=64, pos_embed_max_size=12, dual_attention_layers=(0, 1))


@pytest.mark.parametrize("modulated", [False, True])
def test_sd3_control_module_gradients_match_oracle_autograd(gpu, modulated):
    """UniGenSD3 under autograd (dual attention, context_pre_only last block, transformer-block experts fed per-token tembs, T = 24: every
    attention length is padded to the backward GEMMs' contraction granularity)."""
    import importlib
    cls = importlib.import_module("src.UniGenTransformer").UniGenSD3
    B, hw, T = 2, 16, 24
    ctl = dict(use_shared_expert=True, use_modulate=modulated)
    rcfg = R.SD3Config(use_modulate=modulated, **SD3_TINY)
    base = cls.from_config(dict(SD3_TINY), device=gpu, dtype=BF)
    base.init_condition_block(condition_nums=1, condition_types=["depth"], control_params=dict(ctl))
    base.init_synthetic_(seed=5, std=0.05, bias_std=0.02)
    state = {k: v.detach().cpu() for k, v in base.state_dict().items()}
    inp = R.make_sd3_inputs(rcfg, B=B, hw=hw, T=T)
    t = torch.full((B,), 600.0)
    target = torch.randn(B, 16, hw, hw, generator=torch.Generator().manual_seed(5))
    base.init_trainable_param()
    names = [n for n, p in base.named_parameters() if p.requires_grad]
    assert any(n.startswith("control_transformer_blocks.") for n in names) and not any(n.startswith("transformer_blocks.") for n in names)

    def oracle_grads(dtype):
        st = {k: (v.to(dtype).clone().requires_grad_(True) if k in names else (v.to(dtype) if v.is_floating_point() else v)) for k, v in state.items()}
        out, loss, _ = _step(lambda: R.unigen_sd3_forward(st, rcfg, timestep=t, dtype=dtype, **inp), target, dtype)
        return out, loss, {k: st[k].grad for k in names}

    def hip_grads(dtype):
        model = cls.from_config(dict(SD3_TINY), device=gpu, dtype=dtype)
        model.init_condition_block(condition_nums=1, condition_types=["depth"], control_params=dict(ctl))
        model.load_state_dict({k: (v.to(dtype) if v.is_floating_point() else v) for k, v in state.items()})
        model.init_trainable_param()
        kw = {k: _dev(v, gpu, dtype if k != "gate_uniform" else None) for k, v in inp.items()}
        out, loss, extra = _step(lambda: model(timestep=t.to(gpu), **kw), target, dtype)
        return out, loss, {k: model.get_parameter(k).grad for k in names}, extra

    truth_out, truth_loss, truth = oracle_grads(torch.float32)
    dead = {k for k in names if truth[k] is None or float(truth[k].abs().max()) == 0.0}
    live = [k for k in names if k not in dead]
    z = lambda d, k: (d[k].detach().float().cpu() if d[k] is not None else torch.zeros(state[k].shape))
    cat = lambda d: torch.cat([z(d, k).flatten() for k in names])
    rel = lambda a, b: float((a - b).norm() / b.norm())
    out32, loss32, g32, _ = hip_grads(torch.float32)
    e_all = rel(cat(g32), cat(truth))
    floor = 1e-3 * float(cat(truth).norm()) / len(names) ** 0.5
    worst = max((float((z(g32, k) - truth[k]).norm() / max(float(truth[k].norm()), floor)), k) for k in live)
    print(f"training sd3 mod={int(modulated)} fp32: loss {loss32:.6f} vs {truth_loss:.6f}; all gradients rel_l2 {e_all:.3e}; worst parameter {worst[1]} {worst[0]:.3e}")
    assert rel(out32.float().cpu(), truth_out) <= 1e-4 and abs(loss32 - truth_loss) <= 1e-5 * abs(truth_loss) + 1e-7 and e_all <= 1e-3 and worst[0] <= 5e-3, (e_all, worst)
    assert all(float(z(g32, k).abs().max()) == 0.0 for k in dead)
    ref_out, ref_loss, gref = oracle_grads(BF)
    out16, loss16, g16, extra = hip_grads(BF)
    e_hip, e_ref = rel(cat(g16), cat(truth)), rel(cat(gref), cat(truth))
    print(f"training sd3 mod={int(modulated)} bf16: loss {loss16:.5f} (oracle bf16 {ref_loss:.5f}, fp32 {truth_loss:.5f}); gradients vs fp32: hip {e_hip:.3e}, oracle bf16 {e_ref:.3e}")
    report(f"train_sd3_mod{int(modulated)}_bf16_grads", cat(g16), cat(truth), err_hip_vs_fp32=e_hip, err_oraclebf16_vs_fp32=e_ref)
    assert e_hip <= 1.5 * e_ref + 5e-3 and abs(loss16 - truth_loss) <= 3e-2 * abs(truth_loss), (e_hip, e_ref, loss16, truth_loss)


def test_gradient_checkpointing_gives_identical_gradients(gpu):
    """train.py:317 `transformer.enable_gradient_checkpointing()`: blocks are recomputed in the backward; every kernel is deterministic, so the
    gradients must be bit-identical to the stored-activation run."""
    import importlib
    cls = importlib.import_module("src.UniGenTransformer").UniGenFlux
    rcfg = R.FluxConfig(condition_nums=1, **TINY)
    model = cls.from_config(dict(TINY), device=gpu, dtype=BF)
    model.init_condition_block(condition_nums=1, condition_types=["canny"], control_params=dict(CONTROL))
    model.init_synthetic_(seed=3, std=0.05, bias_std=0.02)
    model.init_trainable_param()
    inp = {k: _dev(v, gpu) for k, v in R.make_inputs(rcfg, B=2, grid=8, T=64).items()}
    t = torch.full((2,), 0.5, dtype=BF, device=gpu)
    target = torch.randn(2, 64, 64, generator=torch.Generator().manual_seed(1))
    names = [n for n, p in model.named_parameters() if p.requires_grad]

    def grads():
        for p in model.parameters():
            p.grad = None
        _step(lambda: model(timestep=t, **inp), target, BF)
        return {k: (None if model.get_parameter(k).grad is None else model.get_parameter(k).grad.clone()) for k in names}

    plain = grads()
    model.enable_gradient_checkpointing()
    ckpt = grads()
    for k in names:
        assert (plain[k] is None) == (ckpt[k] is None) and (plain[k] is None or torch.equal(plain[k], ckpt[k])), k


@pytest.mark.parametrize("over", [dict(single_block_control_method="single_add"), dict(use_shared_expert=False), dict(use_single_trans_blocks=False)],
                         ids=["single_add", "no_shared_expert", "no_single_control"])
def test_training_variants_fp32(gpu, over):
    """Control-path variants of the differentiable forward (image-token-only single-block injection, no shared experts, no control single blocks):
    fp32 verification twins vs the oracle's autograd."""
    import importlib
    cls = importlib.import_module("src.UniGenTransformer").UniGenFlux
    cp = dict(CONTROL); cp.update(over)
    rcfg = R.FluxConfig(condition_nums=1, **TINY, **over)
    model = cls.from_config(dict(TINY), device=gpu, dtype=torch.float32)
    model.init_condition_block(condition_nums=1, condition_types=["canny"], control_params=cp)
    model.init_synthetic_(seed=4, std=0.05, bias_std=0.02)
    state = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    model.init_trainable_param()
    names = [n for n, p in model.named_parameters() if p.requires_grad]
    inp = R.make_inputs(rcfg, B=1, grid=8, T=64)
    t = torch.full((1,), 0.25, dtype=BF)
    target = torch.randn(1, 64, 64, generator=torch.Generator().manual_seed(2))
    kw = {k: _dev(v, gpu, torch.float32 if k != "gate_uniform" and not k.endswith("_ids") else None) for k, v in inp.items()}
    _, loss_h, _ = _step(lambda: model(timestep=t.to(gpu), **kw), target, torch.float32)
    st = {k: (v.clone().requires_grad_(True) if k in names else v) for k, v in state.items()}
    _, loss_r, _ = _step(lambda: R.unigen_flux_forward(st, rcfg, timestep=t, dtype=torch.float32, **inp), target, torch.float32)
    z = lambda g, k: (g.detach().float().cpu() if g is not None else torch.zeros(state[k].shape))
    gh = torch.cat([z(model.get_parameter(k).grad, k).flatten() for k in names])
    gr = torch.cat([z(st[k].grad, k).flatten() for k in names])
    e = float((gh - gr).norm() / gr.norm())
    print(f"training variant {list(over)[0]}: loss {loss_h:.6f} vs {loss_r:.6f}, gradients rel_l2 {e:.3e}")
    assert abs(loss_h - loss_r) <= 1e-5 * abs(loss_r) + 1e-7 and e <= 1e-3, (loss_h, loss_r, e)


def test_training_gradients_at_flux_width_fp32(gpu):
    """FLUX WIDTH (D = 3072, 24 heads of 128, pooled 768, text 4096) at reduced depth (2 + 2 base blocks and their control blocks, both shared
    experts, CoMoE E = 6): loss and control-module gradients of the fp32 verification path against the fp32 oracle's autograd - the 24-head attention
    backward, K = 15360 proj_out and 3072-wide AdaLN / RMSNorm backward at their real sizes."""
    import importlib
    cls = importlib.import_module("src.UniGenTransformer").UniGenFlux
    cfg = dict(num_layers=2, num_single_layers=2)
    rcfg = R.FluxConfig(condition_nums=1, **cfg)
    model = cls.from_config(cfg, device=gpu, dtype=torch.float32)
    model.init_condition_block(condition_nums=1, condition_types=["canny"], control_params=dict(CONTROL))
    model.init_synthetic_(seed=6, std=0.02, bias_std=0.01)
    state = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    model.init_trainable_param()
    names = [n for n, p in model.named_parameters() if p.requires_grad]
    inp = R.make_inputs(rcfg, B=1, grid=8, T=64)
    t = torch.full((1,), 0.5, dtype=BF)
    target = torch.randn(1, 64, 64, generator=torch.Generator().manual_seed(8))
    kw = {k: _dev(v, gpu, torch.float32 if k != "gate_uniform" and not k.endswith("_ids") else None) for k, v in inp.items()}
    _, loss_h, _ = _step(lambda: model(timestep=t.to(gpu), **kw), target, torch.float32)
    st = {k: (v.clone().requires_grad_(True) if k in names else v) for k, v in state.items()}
    _, loss_r, _ = _step(lambda: R.unigen_flux_forward(st, rcfg, timestep=t, dtype=torch.float32, **inp), target, torch.float32)
    z = lambda g, k: (g.detach().float().cpu() if g is not None else torch.zeros(state[k].shape))
    gh = torch.cat([z(model.get_parameter(k).grad, k).flatten() for k in names])
    gr = torch.cat([z(st[k].grad, k).flatten() for k in names])
    e = float((gh - gr).norm() / gr.norm())
    print(f"training FLUX width fp32: loss {loss_h:.6f} vs {loss_r:.6f}, {gh.numel() / 1e9:.2f} B gradient elements, rel_l2 {e:.3e}")
    report("train_flux_width_f32_grads", gh, gr)
    assert abs(loss_h - loss_r) <= 1e-5 * abs(loss_r) + 1e-7 and e <= 1e-3, (loss_h, loss_r, e)


def test_training_golden_fixture_fp32(gpu):
    """The committed training fixture (oracle autograd, fp32: loss, per-parameter gradient norms and first entries) against the HIP fp32 path."""
    import importlib, json, os
    from safetensors import safe_open
    from tests.test_oracle_cpu import load_golden, GOLD
    cfg_d, case, inp, g = load_golden("train_flux_tiny")
    with safe_open(os.path.join(GOLD, "train_flux_tiny.safetensors"), "pt") as f:
        names = json.loads(f.metadata()["trainable"])
    rcfg = R.FluxConfig(condition_nums=1, **cfg_d)
    state = R.make_state(rcfg, seed=case["state_seed"], std=0.05, bias_std=0.02, dtype=torch.float32)
    cls = importlib.import_module("src.UniGenTransformer").UniGenFlux
    model = cls.from_config(cfg_d, device=gpu, dtype=torch.float32)
    model.init_condition_block(condition_nums=1, condition_types=["canny"], control_params=dict(CONTROL))
    res = model.load_state_dict({k: v.to(gpu) for k, v in state.items()}, strict=False)
    assert not res.missing_keys and not res.unexpected_keys
    model.init_trainable_param()
    assert sorted(n for n, p in model.named_parameters() if p.requires_grad) == names
    kw = {k: _dev(v, gpu, torch.float32 if k != "gate_uniform" and not k.endswith("_ids") else None) for k, v in inp.items()}
    _, loss, _ = _step(lambda: model(timestep=g["timestep"].to(gpu), **kw), g["target"], torch.float32)
    assert abs(loss - float(g["out.loss"])) <= 1e-5 * abs(loss)
    worst = 0.0
    total = sum(float(g["grad." + k][0]) ** 2 for k in names) ** 0.5
    for k in names:
        gr = model.get_parameter(k).grad
        gr = gr.float().cpu() if gr is not None else torch.zeros(state[k].shape)
        ref = g["grad." + k]
        worst = max(worst, abs(float(gr.norm()) - float(ref[0])) / max(float(ref[0]), 1e-3 * total / len(names) ** 0.5))
    print(f"training golden: loss {loss:.6f}, worst per-parameter gradient-norm deviation {worst:.3e}")
    assert worst <= 1e-3, worst


def test_sd3_training_gradients_at_sd35_width_fp32(gpu):
    """SD3.5-medium WIDTH (D = 1536, 24 heads of 64, pooled 2048, text 4096) at reduced depth (2 joint blocks, the first with dual attention, the last
    context_pre_only; their control blocks; transformer-block experts; both shared experts): fp32 verification path vs the fp32 oracle's autograd."""
    import importlib
    cls = importlib.import_module("src.UniGenTransformer").UniGenSD3
    cfg = dict(sample_size=16, num_layers=2, pos_embed_max_size=12, dual_attention_layers=(0,))
    ctl = dict(use_shared_expert=True, use_modulate=False)
    rcfg = R.SD3Config(use_modulate=False, **cfg)
    model = cls.from_config(cfg, device=gpu, dtype=torch.float32)
    model.init_condition_block(condition_nums=1, condition_types=["depth"], control_params=dict(ctl))
    model.init_synthetic_(seed=8, std=0.02, bias_std=0.01)
    state = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    model.init_trainable_param()
    names = [n for n, p in model.named_parameters() if p.requires_grad]
    inp = R.make_sd3_inputs(rcfg, B=1, hw=16, T=24)
    t = torch.full((1,), 600.0)
    target = torch.randn(1, 16, 16, 16, generator=torch.Generator().manual_seed(4))
    kw = {k: _dev(v, gpu, torch.float32 if k != "gate_uniform" else None) for k, v in inp.items()}
    _, loss_h, _ = _step(lambda: model(timestep=t.to(gpu), **kw), target, torch.float32)
    st = {k: (v.clone().requires_grad_(True) if k in names else v) for k, v in state.items()}
    _, loss_r, _ = _step(lambda: R.unigen_sd3_forward(st, rcfg, timestep=t, dtype=torch.float32, **inp), target, torch.float32)
    z = lambda g, k: (g.detach().float().cpu() if g is not None else torch.zeros(state[k].shape))
    gh = torch.cat([z(model.get_parameter(k).grad, k).flatten() for k in names])
    gr = torch.cat([z(st[k].grad, k).flatten() for k in names])
    e = float((gh - gr).norm() / gr.norm())
    print(f"training SD3.5 width fp32: loss {loss_h:.6f} vs {loss_r:.6f}, {gh.numel() / 1e9:.2f} B gradient elements, rel_l2 {e:.3e}")
    assert abs(loss_h - loss_r) <= 1e-5 * abs(loss_r) + 1e-7 and e <= 1e-3, (loss_h, loss_r, e)


def test_train_validate_train_keeps_weight_transposes_fresh(gpu):
    """train step -> no-grad validation forward (the inference engine packs QKV / AdaLN / expert weights into new
    buffers and re-points the parameters, freeing the old storages) -> train step, the sequence of the reference's train.py with
    log_validation. The backward's cache of frozen-weight transposes is keyed by address: it must be dropped when storage moves, or the
    second step reads the transposed copy of a different weight. Same weights and inputs: both steps must give the same gradients."""
    import importlib
    from unigen_amd import autograd as AG
    cls = importlib.import_module("src.UniGenTransformer").UniGenFlux
    B, grid, T = 1, 8, 64
    rcfg = R.FluxConfig(condition_nums=1, **TINY)
    model = cls.from_config(dict(TINY), device=gpu, dtype=BF)
    model.init_condition_block(condition_nums=1, condition_types=["canny"], control_params=dict(CONTROL))
    model.init_synthetic_(seed=11, std=0.05, bias_std=0.02)
    model.init_trainable_param()
    names = [n for n, p in model.named_parameters() if p.requires_grad]
    inp = R.make_inputs(rcfg, B=B, grid=grid, T=T)
    kw = {k: _dev(v, gpu, BF if k != "gate_uniform" and not k.endswith("_ids") else None) for k, v in inp.items()}
    t = torch.full((B,), 0.5, dtype=BF, device=gpu)
    target = torch.randn(B, grid * grid, 64, generator=torch.Generator().manual_seed(2))

    def grads():
        model.zero_grad(set_to_none=True)
        _step(lambda: model(timestep=t, **kw), target, BF)
        return torch.cat([(model.get_parameter(k).grad if model.get_parameter(k).grad is not None else torch.zeros_like(model.get_parameter(k))).float().flatten()
                          for k in names])

    g1 = grads()
    assert len(AG._wt_cache) > 0                      # frozen base weights were transposed for dX and cached
    # (the training forward packs to_q | to_k | to_v itself since round 3; the AdaLN linears are packed by the inference engine only)
    ptr_before = model.get_parameter("transformer_blocks.0.norm1.linear.weight").data_ptr()
    with torch.no_grad():
        model(timestep=t, **kw)                       # inference engine: packs, re-points p.data
    assert model.get_parameter("transformer_blocks.0.norm1.linear.weight").data_ptr() != ptr_before
    assert len(AG._wt_cache) == 0 and len(AG._xt_cache) == 0
    g2 = grads()
    e = float((g2 - g1).norm() / g1.norm())
    assert e <= 1e-4, e                               # a stale transposed weight is an O(1) error
    AG.clear_caches()


def test_training_from_the_reference_initial_values_is_alive(gpu):
    """from `init_condition_block`'s own initial values (no synthetic re-init) a train.py run must not start from a dead
    network. Step 1 behaves like any zero-initialised ControlNet: the zero-res projections (and the gate, through l_aux) receive gradients,
    everything upstream of a zero projection gets exactly zero. After ONE update of the zero-res projections every trainable parameter that
    feeds a kept output has a non-zero gradient (round 2 zeroed all control parameters: q = k = 0, tied gate, most gradients stayed zero)."""
    import importlib
    torch.manual_seed(7)
    cls = importlib.import_module("src.UniGenTransformer").UniGenFlux
    B, grid, T = 2, 8, 64
    rcfg = R.FluxConfig(condition_nums=1, **TINY)
    model = cls.from_config(dict(TINY), device=gpu, dtype=BF)              # base: torch default init (a stand-in for from_pretrained)
    model.init_condition_block(condition_nums=1, condition_types=["canny"], control_params=dict(CONTROL))
    model.init_trainable_param()
    names = [n for n, p in model.named_parameters() if p.requires_grad]
    inp = R.make_inputs(rcfg, B=B, grid=grid, T=T)
    kw = {k: _dev(v, gpu, BF if k != "gate_uniform" and not k.endswith("_ids") else None) for k, v in inp.items()}
    t = torch.full((B,), 0.5, dtype=BF, device=gpu)
    target = torch.randn(B, grid * grid, 64, generator=torch.Generator().manual_seed(2))
    gmax = lambda k: 0.0 if model.get_parameter(k).grad is None else float(model.get_parameter(k).grad.float().abs().max())

    _step(lambda: model(timestep=t, **kw), target, BF)
    zero_res = [k for k in names if k.startswith("controlnet_add_")]
    assert zero_res and all(gmax(k) > 0 for k in zero_res if k.endswith(".weight")), "zero-res projections must learn at step 1"
    assert gmax("moe.moe_layer.gate.wg.weight") > 0                        # l_aux
    assert all(gmax(k) == 0.0 for k in names if k.startswith(("control_joint_trans_blocks.", "control_single_trans_blocks.", "shared_expert.")))
    with torch.no_grad():
        for k in zero_res:
            p = model.get_parameter(k)
            p.add_(p.grad.to(p.dtype), alpha=-50.0)      # one plain SGD step on the zero-res projections
    model.zero_grad(set_to_none=True)
    _step(lambda: model(timestep=t, **kw), target, BF)
    # parameters that only feed a DISCARDED context output (src/UniGenTransformer.py:1097,1022) get zero in the reference too
    dead_pat = (".attn.to_add_out.", ".ff_context.", ".attn.add_q_proj.", ".attn.norm_added_q.")
    dead = [k for k in names if k.startswith(("control_joint_trans_blocks.", "shared_expert.1.")) and any(d in k for d in dead_pat)]
    live = [k for k in names if k not in dead]
    still_zero = [k for k in live if gmax(k) == 0.0]
    assert not still_zero, f"{len(still_zero)} of {len(live)} live parameters have zero gradient, e.g. {still_zero[:6]}"
    assert all(gmax(k) == 0.0 for k in dead)
    assert all(torch.isfinite(model.get_parameter(k).grad.float()).all() for k in live)


def test_consistency_module_gradients_match_oracle_autograd(gpu):
    """`use_consis_module` under autograd (round 3): gradients of the control modules - consis_module.0 included, consis_module.1 exactly zero / None as
    in the reference, where it is built but never called - through the fp32 twins against torch autograd of the fp32 oracle."""
    import importlib
    cls = importlib.import_module("src.UniGenTransformer").UniGenFlux
    B, grid, T = 1, 8, 64
    cp = dict(CONTROL); cp.update(use_consis_module=True)
    rcfg = R.FluxConfig(condition_nums=1, use_consis_module=True, **TINY)
    base = cls.from_config(dict(TINY), device=gpu, dtype=torch.float32)
    base.init_condition_block(condition_nums=1, condition_types=["canny"], control_params=cp)
    base.init_synthetic_(seed=13, std=0.05, bias_std=0.02)
    base.init_trainable_param()
    state = {k: v.detach().cpu() for k, v in base.state_dict().items()}
    names = [n for n, p in base.named_parameters() if p.requires_grad]
    assert any(n.startswith("consis_module.0.") for n in names) and any(n.startswith("consis_module.1.") for n in names)
    inp = R.make_inputs(rcfg, B=B, grid=grid, T=T)
    t = torch.full((B,), 0.75, dtype=BF)
    target = torch.randn(B, grid * grid, 64, generator=torch.Generator().manual_seed(5))
    st = {k: (v.clone().requires_grad_(True) if k in names else v) for k, v in state.items()}
    _, loss_t, _ = _step(lambda: R.unigen_flux_forward(st, rcfg, timestep=t, dtype=torch.float32, **inp), target, torch.float32)
    kw = {k: _dev(v, gpu, torch.float32 if k != "gate_uniform" and not k.endswith("_ids") else None) for k, v in inp.items()}
    _, loss_h, _ = _step(lambda: base(timestep=t.to(gpu), **kw), target, torch.float32)
    z = lambda g, k: g.detach().float().cpu() if g is not None else torch.zeros(state[k].shape)
    gt = torch.cat([z(st[k].grad, k).flatten() for k in names])
    gh = torch.cat([z(base.get_parameter(k).grad, k).flatten() for k in names])
    e = float((gh - gt).norm() / gt.norm())
    print(f"consis training fp32: loss {loss_h:.6f} vs {loss_t:.6f}; gradients rel_l2 {e:.3e}")
    assert abs(loss_h - loss_t) <= 1e-5 * abs(loss_t) + 1e-7 and e <= 1e-3, (loss_h, loss_t, e)
    c0 = torch.cat([z(st[k].grad, k).flatten() for k in names if k.startswith("consis_module.0.attn.to_q")])
    assert float(c0.abs().max()) > 0, "consis_module.0 must receive a gradient"
    for k in names:
        if k.startswith("consis_module.1."):
            g = base.get_parameter(k).grad
            assert g is None or float(g.abs().max()) == 0.0, k
