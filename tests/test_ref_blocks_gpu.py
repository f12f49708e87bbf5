"""The engine's SD3 block launches against fixtures that ORIGINATE IN THE REFERENCE (tests/golden/ref_blocks.safetensors: outputs of the reference's
own JointTransformerBlock.forward / SD3SingleTransformerBlock.forward, src/UniGenUtils.py:386-522, executed from /root/reference by
tests/golden/make_ref_wiring_golden.py in the build container; no q/k norm - the branch behind diffusers' RMSNorm stays restated).

A UniGenSD3 engine of the fixture's width (D = 128, H = 2, dh = 64, qk_norm = None) takes the fixture's parameters at the reference's state-dict
names and runs the product's own launch sequences through the C ABI:
  HipModule._double_block       joint.plain  -> transformer_blocks.0      (both streams written back)
                                joint.dual   -> transformer_blocks.1      (SD35AdaLayerNormZeroX + attn2 on the block's INPUT)
                                joint.cpo    -> transformer_blocks.2      (last block: AdaLayerNormContinuous context, K/V only)
                                joint.dual_cpo -> shared_expert.1         (the second shared expert's shape)
  UniGenSD3._expert_blocks      single.expert{0,1} -> moe.moe_layer.experts.deepspeed_experts.{e}.0: batch 1 over an expert's capacity slots, per-token
                                AdaLN rows gathered per (sample, expert), empty slots zero rows that still take part in the attention

Tolerance: fp32 verification twins (same orchestration, fp32 storage) <= 1e-3 against the reference's fp32 outputs (measured ~1e-6); the bf16 product
launches no further from the reference's fp32 result than 1.25 x the reference's own bf16 eager evaluation, and <= 8e-3 against that bf16 output."""
import os

import pytest
import torch
from safetensors import safe_open

from tests.util import rel_l2, report

pytestmark = pytest.mark.gpu
BF, F32 = torch.bfloat16, torch.float32
FIX = os.path.join(os.path.dirname(__file__), "golden", "ref_blocks.safetensors")
CFG = dict(sample_size=16, num_layers=3, attention_head_dim=64, num_attention_heads=2, joint_attention_dim=64, caption_projection_dim=128,
           pooled_projection_dim=64, pos_embed_max_size=12, dual_attention_layers=(1,), qk_norm=None)
PLACE = {"joint.plain": "transformer_blocks.0", "joint.dual": "transformer_blocks.1", "joint.cpo": "transformer_blocks.2", "joint.dual_cpo": "shared_expert.1",
         "single.expert0": "moe.moe_layer.experts.deepspeed_experts.0.0", "single.expert1": "moe.moe_layer.experts.deepspeed_experts.1.0"}


@pytest.fixture(scope="module")
def fx():
    with safe_open(FIX, "pt") as f:
        return {k: f.get_tensor(k) for k in f.keys()}


def _model(gpu, fx, dt):
    from unigen_amd.sd3 import UniGenSD3
    model = UniGenSD3.from_config(CFG, device=gpu, dtype=dt)
    model.init_condition_block(condition_nums=1, condition_types=["depth"], control_params=dict(use_shared_expert=True, expert_num=2))
    sd, names = {}, {n for n, _ in model.named_parameters()}
    for kind, prefix in PLACE.items():
        pre = f"w.{kind}."
        for k, v in fx.items():
            if k.startswith(pre):
                name = f"{prefix}.{k[len(pre):]}"
                assert name in names, f"the engine has no parameter {name} (reference state-dict name)"
                sd[name] = v.to(gpu, dt)
    res = model.load_state_dict(sd, strict=False)
    assert not res.unexpected_keys
    return model


def _check(name, got, fx, key, dtag):
    want32 = fx[key.format(tag="f32")]
    m = report(f"refblocks_{name}_{dtag}", got, fx[key.format(tag=dtag)])
    if dtag == "f32":
        assert m["rel_l2"] <= 1e-3, m
    else:
        ref_err = rel_l2(fx[key.format(tag="bf16")].float(), want32)
        err = rel_l2(got.float(), want32)
        assert err <= 1.25 * ref_err, (name, err, ref_err)
        assert m["rel_l2"] <= 8e-3, m


@pytest.mark.parametrize("dtag", ["f32", "bf16"])
def test_joint_block_launches_match_the_reference_forward(gpu, fx, dtag):
    from unigen_amd.engine import _Stream
    dt = F32 if dtag == "f32" else BF
    model = _model(gpu, fx, dt)
    B, N, D = fx["in.x"].shape
    T = fx["in.enc"].shape[1]
    t2 = lambda k: fx[f"in.{k}"].to(gpu, dt).reshape(-1, D).clone()
    temb = fx["in.temb"].to(gpu, dt).contiguous()
    with torch.no_grad():
        # plain: both streams, in place
        x, e = t2("x"), t2("enc")
        model._emb_tab.clear()
        model._double_block(PLACE["joint.plain"], B, _Stream(x, N), _Stream(x, N), _Stream(e, T), _Stream(e, T), temb, None, "base")
        _check("joint_plain_x", x.view(B, N, D), fx, "out.joint.plain.sample.{tag}.x", dtag)
        _check("joint_plain_enc", e.view(B, T, D), fx, "out.joint.plain.sample.{tag}.enc", dtag)
        # dual attention, in place (attn2 must read the block's input)
        x, e = t2("x"), t2("enc")
        model._double_block(PLACE["joint.dual"], B, _Stream(x, N), _Stream(x, N), _Stream(e, T), _Stream(e, T), temb, None, "base", dual=True)
        _check("joint_dual_x", x.view(B, N, D), fx, "out.joint.dual.sample.{tag}.x", dtag)
        _check("joint_dual_enc", e.view(B, T, D), fx, "out.joint.dual.sample.{tag}.enc", dtag)
        # context_pre_only (the last base block): the context only contributes K/V through AdaLayerNormContinuous
        x = t2("x")
        model._double_block(PLACE["joint.cpo"], B, _Stream(x, N), _Stream(x, N), _Stream(t2("enc"), T), None, temb, None, "base", ctx_continuous=True)
        _check("joint_cpo_x", x.view(B, N, D), fx, "out.joint.cpo.sample.{tag}.x", dtag)
        # context_pre_only + dual attention: shared_expert[1], out of place as the control branch runs it
        z = torch.empty(B * N, D, device=gpu, dtype=dt)
        model._double_block(PLACE["joint.dual_cpo"], B, _Stream(t2("x"), N), _Stream(z, N), _Stream(t2("enc"), T), None, temb, None, "se1", dual=True,
                            ctx_continuous=True)
        _check("joint_dual_cpo_x", z.view(B, N, D), fx, "out.joint.dual_cpo.sample.{tag}.x", dtag)


@pytest.mark.parametrize("dtag", ["f32", "bf16"])
def test_expert_block_launches_match_the_reference_single_block(gpu, fx, dtag):
    """UniGenSD3._expert_blocks (expert[0] of every expert in grouped launches) on a token table and a token_of_slot map that reproduce the
    fixture's dispatched slots: SD3SingleTransformerBlock.forward with per-token temb, :261."""
    dt = F32 if dtag == "f32" else BF
    model = _model(gpu, fx, dt)
    xs, rows, sos = fx["in.single_tok.x"], fx["in.single_tok.temb_rows"], fx["in.single_tok.sample_of_slot"].long()
    E, C, D = xs.shape
    B = rows.shape[0]
    N = int(max((sos == b).sum() for b in range(B)))          # tokens per sample: every filled slot gets its own token of its sample
    src = torch.zeros(B * N, D)
    tos = torch.full((E, C), -1, dtype=torch.int32)
    nxt = [0] * B
    for e in range(E):
        for c in range(C):
            b = int(sos[e, c])
            if b >= 0:
                tok = b * N + nxt[b]
                nxt[b] += 1
                src[tok], tos[e, c] = xs[e, c].float(), tok
    tos = tos.to(gpu)
    e_ar = torch.arange(E, device=gpu, dtype=torch.int32).view(E, 1)
    sidx = (e_ar + E * torch.where(tos >= 0, torch.div(tos, N, rounding_mode="floor"), torch.full_like(tos, B))).to(torch.int32).reshape(-1).contiguous()
    with torch.no_grad():
        model._emb_tab.clear()
        y = model._expert_blocks(0, B, N, src.to(gpu, dt), rows.to(gpu, dt).contiguous(), tos, sidx, C, "moe_yh")
    for e in range(E):
        _check(f"single_expert{e}", y[e][None], fx, "out.single.expert%d.token.{tag}.x" % e, dtag)
