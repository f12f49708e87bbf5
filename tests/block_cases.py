"""Teacher-forced block-level parity cases: every block type of the hot path is run ALONE, on the same O(1)
inputs, by the CPU oracle (oracle side, used by tests/golden/make_golden.py to write the fixtures and by the CPU pin test) and by
the HIP engine (hip side, used by the -m gpu tests). Inputs are drawn so that every branch (attention, MLP, gates) contributes O(1):
a structural slip (a wrong stream, a stale LayerNorm input, a swapped chunk) moves the output by O(0.1..1), far above the tolerance.

Reference call sites the cases restate: src/UniGenTransformer.py:1129 (base double), :1151 (base single), :1097 / :1102 (control
joint / single), :1015 / :1020 (shared experts), :969-1026 (CoMoE); UniGenSD3: :539-623, src/UniGenUtils.py:440-522.
"""
from __future__ import annotations

from typing import Dict

import torch

BF = torch.bfloat16
FLUX_TINY = dict(num_layers=2, num_single_layers=4, attention_head_dim=128, num_attention_heads=2, joint_attention_dim=64, pooled_projection_dim=64)
SD3_TINY = dict(sample_size=16, num_layers=3, attention_head_dim=64, num_attention_heads=2, joint_attention_dim=64, caption_projection_dim=128,
                pooled_projection_dim=64, pos_embed_max_size=12, dual_attention_layers=(0, 1))
FLUX_CASE = dict(B=2, grid=6, T=24, state_seed=21, input_seed=4711)
SD3_CASE = dict(B=2, grid=8, T=24, state_seed=22, input_seed=4712)
STD, BIAS_STD = 0.05, 0.02


def _rn(g, *shape, scale=1.0):
    return (scale * torch.randn(*shape, generator=g)).to(BF)


def flux_inputs(R, cfg) -> Dict[str, torch.Tensor]:
    c = FLUX_CASE
    g = torch.Generator().manual_seed(c["input_seed"])
    B, N, T, D = c["B"], c["grid"] ** 2, c["T"], cfg.inner_dim
    ids = R.make_ids(c["grid"], c["grid"])
    return dict(x=_rn(g, B, N, D), x2=_rn(g, B, N, D), enc=_rn(g, B, T, D), ctrl_enc=_rn(g, B, T, D), cond=_rn(g, B, N, D), h=_rn(g, B, T + N, D),
                temb=_rn(g, B, D), control_temb=_rn(g, B, D), condition_temb=_rn(g, B, D), cond_tokens=_rn(g, B, N, cfg.in_channels),
                pooled=_rn(g, B, cfg.pooled_projection_dim), cond_pooled=_rn(g, B, cfg.pooled_projection_dim),
                uniform=torch.rand(B * N, cfg.expert_nums, generator=g), img_ids=ids, cond_ids=ids.clone(), txt_ids=torch.zeros(T, 3, dtype=BF))


def flux_oracle(R, st, cfg, inp, dtype) -> Dict[str, torch.Tensor]:
    """Every flux block case on the oracle, arithmetic `dtype` (bf16 = the reference's rounding points, fp32 = truth)."""
    f = lambda k: inp[k].to(dtype)
    img, txt, cid = inp["img_ids"], inp["txt_ids"], inp["cond_ids"]
    rope_base = R.flux_pos_embed(torch.cat([txt, img], 0), cfg.axes_dims_rope)
    out = {}
    e, x = R.flux_double_block(st, "transformer_blocks.1", cfg, f("x"), f("enc"), f("temb"), rope_base, text_first=True)
    out["flux_double.x"], out["flux_double.enc"] = x, e
    out["flux_single.h"] = R.flux_single_block(st, "single_transformer_blocks.2", cfg, f("h"), f("temb"), rope_base)
    crope_j = R.control_rope(cfg, torch.cat([img, txt], 0))
    out["ctl_joint.z"] = R.flux_double_block(st, "control_joint_trans_blocks.0", cfg, f("x"), f("ctrl_enc"), f("condition_temb"), crope_j, text_first=False)[1]
    out["ctl_joint.z2"] = R.flux_double_block(st, "control_joint_trans_blocks.0", cfg, f("x2"), f("ctrl_enc"), f("condition_temb"), crope_j, text_first=False)[1]
    crope_s = R.control_rope(cfg, torch.cat([txt, img], 0))
    out["ctl_single.z"] = R.flux_single_block(st, "control_single_trans_blocks.1", cfg, f("h"), f("condition_temb"), crope_s)
    rope0 = R.control_rope(cfg, torch.cat([img, cid], 0))
    cs, xs = R.flux_double_block(st, "shared_expert.0", cfg, f("x"), f("cond"), f("condition_temb"), rope0, text_first=False)
    out["shared0.x"], out["shared0.c"] = xs, cs
    rope1 = R.control_rope(cfg, torch.cat([img, cid, txt], 0))
    out["shared1.xc"] = R.flux_double_block(st, "shared_expert.1", cfg, torch.cat([f("x"), f("cond")], 1), f("ctrl_enc"), f("control_temb"), rope1,
                                            text_first=False)[1]
    eh, ec, l_aux, cnt, routing = R.comoe(st, cfg, f("x"), f("cond_tokens"), f("ctrl_enc"), f("control_temb"), f("condition_temb"), f("pooled"),
                                          f("cond_pooled"), dict(img_ids=img, prompt_ids=txt, condition_ids=cid), inp["uniform"])
    out["comoe.z0"], out["comoe.counts"], out["comoe.l_aux"] = eh + ec, cnt, l_aux.reshape(1).float()
    out["comoe.idx"] = routing["idx"].to(torch.int32)
    return out


def flux_hip(model, inp) -> Dict[str, torch.Tensor]:
    """The same cases on the HIP engine (bf16, or the fp32 verification twins when the model's parameters are fp32)."""
    from unigen_amd.engine import _Stream
    dev, dt, c = model.device, model.dtype, FLUX_CASE
    B, N, T, D = c["B"], c["grid"] ** 2, c["T"], model.inner_dim
    t2 = lambda k: inp[k].to(dev, dt).reshape(-1, inp[k].shape[-1]).clone()
    v = lambda k: inp[k].to(dev, dt).contiguous()
    img, txt, cid = inp["img_ids"].to(dev), inp["txt_ids"].to(dev), inp["cond_ids"].to(dev)
    rope_base = model._rope([txt, img], None)
    rope_ctl = model._rope([txt, img], img.dtype)
    out = {}
    model._emb_tab.clear()
    x, e = t2("x"), t2("enc")
    model._double_block("transformer_blocks.1", B, _Stream(x, N), _Stream(x, N), _Stream(e, T), _Stream(e, T), v("temb"), rope_base, "base")
    out["flux_double.x"], out["flux_double.enc"] = x.view(B, N, D).clone(), e.view(B, T, D).clone()
    h = t2("h")
    model._single_block("single_transformer_blocks.2", B, _Stream(h, T + N), _Stream(h, T + N), v("temb"), rope_base)
    out["flux_single.h"] = h.view(B, T + N, D).clone()
    # control joint block: reads x, writes z; second use with the context K/V cached (same block, same step)
    z, ce = torch.empty(B * N, D, device=dev, dtype=dt), t2("ctrl_enc")
    model._double_block("control_joint_trans_blocks.0", B, _Stream(t2("x"), N), _Stream(z, N), _Stream(ce, T), None, v("condition_temb"), rope_ctl, "ctl")
    out["ctl_joint.z"] = z.view(B, N, D).clone()
    model._double_block("control_joint_trans_blocks.0", B, _Stream(t2("x2"), N), _Stream(z, N), _Stream(ce, T), None, v("condition_temb"), rope_ctl, "ctl",
                        ctx_cached=True)
    out["ctl_joint.z2"] = z.view(B, N, D).clone()
    zj = torch.empty(B * (T + N), D, device=dev, dtype=dt)
    model._single_block("control_single_trans_blocks.1", B, _Stream(t2("h"), T + N), _Stream(zj, T + N), v("condition_temb"), rope_ctl)
    out["ctl_single.z"] = zj.view(B, T + N, D).clone()
    # shared experts, exactly as _comoe drives them: [image | condition] halves of one [B, 2N, D] buffer
    xc = torch.empty(B * 2 * N, D, device=dev, dtype=dt)
    rope0 = model._rope([cid, img], img.dtype)
    model._double_block("shared_expert.0", B, _Stream(t2("x"), N), _Stream(xc, N, 2 * N), _Stream(t2("cond"), N), _Stream(xc[N:], N, 2 * N),
                        v("condition_temb"), rope0, "se0")
    out["shared0.x"], out["shared0.c"] = xc.view(B, 2 * N, D)[:, :N].clone(), xc.view(B, 2 * N, D)[:, N:].clone()
    xc1 = torch.cat([inp["x"], inp["cond"]], 1).to(dev, dt).reshape(B * 2 * N, D).clone()
    rope1 = model._rope([txt, img, cid], img.dtype)
    model._double_block("shared_expert.1", B, _Stream(xc1, 2 * N), _Stream(xc1, 2 * N), _Stream(t2("ctrl_enc"), T), None, v("control_temb"), rope1, "se1")
    out["shared1.xc"] = xc1.view(B, 2 * N, D).clone()
    z0 = torch.empty(B * N, D, device=dev, dtype=dt)
    l_aux, cnt = model._comoe(B, N, T, t2("x"), v("cond_tokens"), t2("ctrl_enc"), v("control_temb"), v("condition_temb"), v("pooled"), v("cond_pooled"),
                              img, txt, cid, inp["uniform"].to(dev), z0, accumulate=False)
    out["comoe.z0"], out["comoe.counts"], out["comoe.l_aux"] = z0.view(B, N, D).clone(), cnt.clone(), l_aux.clone()
    out["comoe.idx"] = model._w("moe_idx", (B * N,), torch.int32).clone()
    return out


# ---------------------------------------------------------------------------------------------------------------------
# UniGenSD3
# ---------------------------------------------------------------------------------------------------------------------

def sd3_inputs(R, cfg) -> Dict[str, torch.Tensor]:
    c = SD3_CASE
    g = torch.Generator().manual_seed(c["input_seed"])
    B, N, T, D = c["B"], c["grid"] ** 2, c["T"], cfg.inner_dim
    return dict(x=_rn(g, B, N, D), enc=_rn(g, B, T, D), ctrl_enc=_rn(g, B, T, D), cond=_rn(g, B, N, D), temb=_rn(g, B, D), control_temb=_rn(g, B, D),
                condition_temb=_rn(g, B, D), pooled=_rn(g, B, cfg.pooled_projection_dim), cond_pooled=_rn(g, B, cfg.pooled_projection_dim),
                uniform=torch.rand(B * N, cfg.expert_nums, generator=g))


def sd3_oracle(R, st, cfg, inp, dtype) -> Dict[str, torch.Tensor]:
    f = lambda k: inp[k].to(dtype)
    H = cfg.num_attention_heads
    out = {}
    e, x = R.sd3_joint_block(st, "transformer_blocks.0", H, f("x"), f("enc"), f("temb"), context_pre_only=False, dual=True)
    out["sd3_dual.x"], out["sd3_dual.enc"] = x, e
    out["sd3_last.x"] = R.sd3_joint_block(st, "transformer_blocks.2", H, f("x"), f("enc"), f("temb"), context_pre_only=True, dual=False)[1]
    out["sd3_ctl_dual.z"] = R.sd3_joint_block(st, "control_transformer_blocks.1", H, f("x"), f("ctrl_enc"), f("condition_temb"), context_pre_only=False, dual=True)[1]
    eh, ec, l_aux, cnt, routing = R.sd3_comoe(st, cfg, f("x"), f("cond"), f("ctrl_enc"), f("control_temb"), f("condition_temb"), f("pooled"),
                                              f("cond_pooled"), inp["uniform"])
    out["sd3_comoe.z0"], out["sd3_comoe.counts"] = eh + ec, cnt
    return out


def sd3_hip(model, inp) -> Dict[str, torch.Tensor]:
    from unigen_amd.engine import _Stream
    dev, dt, c = model.device, model.dtype, SD3_CASE
    B, N, T, D = c["B"], c["grid"] ** 2, c["T"], model.inner_dim
    t2 = lambda k: inp[k].to(dev, dt).reshape(-1, inp[k].shape[-1]).clone()
    v = lambda k: inp[k].to(dev, dt).contiguous()
    out = {}
    model._emb_tab.clear()
    x, e = t2("x"), t2("enc")
    model._double_block("transformer_blocks.0", B, _Stream(x, N), _Stream(x, N), _Stream(e, T), _Stream(e, T), v("temb"), None, "base", dual=True)   # IN PLACE
    out["sd3_dual.x"], out["sd3_dual.enc"] = x.view(B, N, D).clone(), e.view(B, T, D).clone()
    x = t2("x")
    model._double_block("transformer_blocks.2", B, _Stream(x, N), _Stream(x, N), _Stream(t2("enc"), T), None, v("temb"), None, "base", ctx_continuous=True)
    out["sd3_last.x"] = x.view(B, N, D).clone()
    z = torch.empty(B * N, D, device=dev, dtype=dt)
    model._double_block("control_transformer_blocks.1", B, _Stream(t2("x"), N), _Stream(z, N), _Stream(t2("ctrl_enc"), T), None, v("condition_temb"), None,
                        "ctl", dual=True)
    out["sd3_ctl_dual.z"] = z.view(B, N, D).clone()
    z0 = torch.empty(B * N, D, device=dev, dtype=dt)
    l_aux, cnt = model._comoe(B, N, T, t2("x"), t2("cond"), t2("ctrl_enc"), v("control_temb"), v("condition_temb"), v("pooled"), v("cond_pooled"),
                              inp["uniform"].to(dev), z0)
    out["sd3_comoe.z0"], out["sd3_comoe.counts"] = z0.view(B, N, D).clone(), cnt.clone()
    return out
