"""The differentiable forward of UniGenFlux: the second caller of the same forward. Reference: train.py:622-662 runs
`transformer(...)` under autograd, adds `flow_loss.mean() + sum(add_losses.values())` and calls `accelerator.backward(loss)`; the trainable set is
`trainable_control_modules` (src/UniGenTransformer.py:785-787), everything else is frozen but still carries gradients to the inputs of the control
modules.

Same arithmetic and rounding points as the inference engine (unigen_amd/flux.py, engine.py) but functional - every op allocates its result, so
torch.autograd can be the tape, as it is in the reference. The heavy ops are the HIP kernels behind unigen_amd/autograd.py (GEMM forward / dgrad /
wgrad, flash attention and its GEMM-based backward, AdaLN modulate, q/k RMSNorm + RoPE, GELU); the glue that the reference also leaves to torch
elementwise kernels (residual adds, gate multiplies, SiLU on [B, D] vectors, concatenations, the [S, E] gate softmax and the dispatch / combine
gathers) stays torch on the GPU. Activation memory is the reference's too: wrap blocks in torch.utils.checkpoint for full-size runs.
fp32 parameters run the `_f32` verification twins end to end."""
from __future__ import annotations

import math
from typing import Optional

import torch
import torch.nn.functional as F

from . import autograd as A
from . import ops


def _p(model, name: str) -> Optional[torch.Tensor]:
    return model.get_parameter(name) if name in model._pnames() else None


def _lin(model, prefix: str, x: torch.Tensor) -> torch.Tensor:
    return A.linear(x, _p(model, prefix + ".weight"), _p(model, prefix + ".bias"))


def _time_text_embed(model, prefix: str, t_f32, pooled, g_f32):
    """CombinedTimestep(Guidance)TextProjEmbeddings (diffusers embeddings.py)."""
    B, dt = pooled.shape[0], pooled.dtype
    tp = ops.timestep_embed(t_f32, torch.empty(B, 256, device=pooled.device, dtype=dt))
    temb = _lin(model, prefix + ".timestep_embedder.linear_2", F.silu(_lin(model, prefix + ".timestep_embedder.linear_1", tp)))
    if g_f32 is not None:
        gp = ops.timestep_embed(g_f32, torch.empty(B, 256, device=pooled.device, dtype=dt))
        temb = temb + _lin(model, prefix + ".guidance_embedder.linear_2", F.silu(_lin(model, prefix + ".guidance_embedder.linear_1", gp)))
    pe = _lin(model, prefix + ".text_embedder.linear_2", F.silu(_lin(model, prefix + ".text_embedder.linear_1", pooled)))
    return temb + pe


def _adaln(model, prefix: str, x, emb, nchunks: int):
    """AdaLayerNormZero / ZeroSingle: (modulated x, remaining chunks...)."""
    e = _lin(model, prefix + ".linear", F.silu(emb))
    ch = e.chunk(nchunks, dim=1)
    return (A.adaln_modulate(x, ch[0], ch[1]),) + tuple(ch[2:])


def _feed_forward(model, prefix: str, x):
    return _lin(model, prefix + ".net.2", A.GeluTanh.apply(_lin(model, prefix + ".net.0.proj", x)))


def _attention(model, prefix: str, x, enc, rope, text_first: bool):
    """Attention of a Flux block (FluxAttnProcessor2_0: [context | sample]; JointAttnRopeProcessor src/UniGenUtils.py:532-622: [sample | context]).
    RoPE rows follow the joint order, so each stream is rotated at its offset before the concatenation."""
    H = model.config.num_attention_heads
    N = x.shape[1]
    T = enc.shape[1] if enc is not None else 0
    x_off, e_off = (T, 0) if text_first else (0, N)
    q = A.qk_norm_rope(_lin(model, prefix + ".to_q", x), _p(model, prefix + ".norm_q.weight"), rope, H, x_off)
    k = A.qk_norm_rope(_lin(model, prefix + ".to_k", x), _p(model, prefix + ".norm_k.weight"), rope, H, x_off)
    v = _lin(model, prefix + ".to_v", x)
    if enc is None:
        return A.attention(q, k, v, H), None
    eq = A.qk_norm_rope(_lin(model, prefix + ".add_q_proj", enc), _p(model, prefix + ".norm_added_q.weight"), rope, H, e_off)
    ek = A.qk_norm_rope(_lin(model, prefix + ".add_k_proj", enc), _p(model, prefix + ".norm_added_k.weight"), rope, H, e_off)
    ev = _lin(model, prefix + ".add_v_proj", enc)
    order = (lambda a, b: torch.cat([a, b], 1)) if text_first else (lambda a, b: torch.cat([b, a], 1))
    o = A.attention(order(eq, q), order(ek, k), order(ev, v), H)
    eo, xo = (o[:, :T], o[:, T:]) if text_first else (o[:, N:], o[:, :N])
    return _lin(model, prefix + ".to_out.0", xo), _lin(model, prefix + ".to_add_out", eo)


def _ckpt(model, fn, *tensors):
    """Block-level activation checkpointing when `enable_gradient_checkpointing()` was called (train.py:317), as diffusers does per block."""
    if getattr(model, "_grad_checkpoint", False):
        from torch.utils.checkpoint import checkpoint
        return checkpoint(fn, *tensors, use_reentrant=False)
    return fn(*tensors)


def _double_block(model, prefix: str, x, enc, temb, rope, text_first: bool = True):
    return _ckpt(model, lambda x_, enc_, temb_: _double_block_body(model, prefix, x_, enc_, temb_, rope, text_first), x, enc, temb)


def _single_block(model, prefix: str, h, temb, rope):
    return _ckpt(model, lambda h_, temb_: _single_block_body(model, prefix, h_, temb_, rope), h, temb)


def _double_block_body(model, prefix: str, x, enc, temb, rope, text_first: bool = True):
    """diffusers FluxTransformerBlock.forward. Returns (enc, x)."""
    n, g, sh, sc, gm = _adaln(model, prefix + ".norm1", x, temb, 6)
    nc, cg, csh, csc, cgm = _adaln(model, prefix + ".norm1_context", enc, temb, 6)
    a, ca = _attention(model, prefix + ".attn", n, nc, rope, text_first)
    x = x + g.unsqueeze(1) * a
    x = x + gm.unsqueeze(1) * _feed_forward(model, prefix + ".ff", A.adaln_modulate(x, sh, sc))
    enc = enc + cg.unsqueeze(1) * ca
    enc = enc + cgm.unsqueeze(1) * _feed_forward(model, prefix + ".ff_context", A.adaln_modulate(enc, csh, csc))
    return enc, x


def _single_block_body(model, prefix: str, h, temb, rope):
    """diffusers FluxSingleTransformerBlock.forward."""
    n, gate = _adaln(model, prefix + ".norm", h, temb, 3)
    mlp = A.GeluTanh.apply(_lin(model, prefix + ".proj_mlp", n))
    a, _ = _attention(model, prefix + ".attn", n, None, rope, True)
    return h + gate.unsqueeze(1) * _lin(model, prefix + ".proj_out", torch.cat([a, mlp], dim=2))


def _comoe(model, x, cond_tokens, ctrl_enc, control_temb, condition_temb, pooled, cond_pooled, img_ids, txt_ids, cond_ids, uniform):
    """preprocess_moe_forward + moe_forward for one condition (src/UniGenTransformer.py:969-1068; deepspeed top1gating).
    Routing decisions (argmax, capacity, Random Token Selection) are the HIP kernel's; what is differentiable - the gate probabilities in
    the combine weights and in l_aux - stays on the tape."""
    ctl, dt = model._ctl, x.dtype
    B, N, D = x.shape
    E, S = ctl.expert_nums, B * N
    C = max(int(math.ceil(S / E)), 4)
    c = _lin(model, "control_x_embedder", cond_tokens)
    logits = F.linear((x + c).reshape(S, D).float(), model.get_parameter("moe.moe_layer.gate.wg.weight").float())     # TopKGate: fp32
    gates = F.softmax(logits, dim=1)
    idx = torch.argmax(gates, dim=1).to(torch.int32)
    if uniform is None:
        uniform = torch.rand(S, E, device=x.device, dtype=torch.float32)
    slot, tos = torch.empty(S, device=x.device, dtype=torch.int32), torch.empty(E, C, device=x.device, dtype=torch.int32)
    exp_counts, l_aux_k = torch.empty(E, device=x.device, dtype=torch.int64), torch.empty(1, device=x.device, dtype=torch.float32)
    ops.moe_capacity_rts(gates.detach().contiguous(), idx, uniform.contiguous(), C, slot, tos, exp_counts, l_aux_k)
    ce = F.one_hot(idx.long(), E).float().mean(0)
    l_aux = torch.sum(gates.mean(0) * ce) * E
    # dispatch = row gather (einsum("sec,sm->ecm") with a one-hot mask), zeros in empty slots
    tos_l = tos.view(-1).long()
    valid = tos_l >= 0
    src = tos_l.clamp_min(0)

    def dispatch(t2d):
        return (t2d[src] * valid.unsqueeze(1).to(t2d.dtype)).view(E, C, -1)

    xd, cd = dispatch(x.reshape(S, D)), dispatch(c.reshape(S, D))
    sample_of = (src // N)
    pe = "moe.moe_layer.experts.deepspeed_experts."
    yh, yc = [], []
    for e in range(E):
        rows = slice(e * C, (e + 1) * C)
        s_c = _lin(model, f"{pe}{e}.0.1", cond_pooled)[sample_of[rows]]          # the modulation vector of each slot's sample
        s_h = _lin(model, f"{pe}{e}.1.1", pooled)[sample_of[rows]]
        c_e = _lin(model, f"{pe}{e}.0.0", s_c * cd[e])                          # expert_forward :957-959: Linear_W(s * x) + b
        h_e = _lin(model, f"{pe}{e}.1.0", s_h * (xd[e] + c_e))
        yh.append(h_e); yc.append(c_e)
    yh, yc = torch.stack(yh).view(E * C, D), torch.stack(yc).view(E * C, D)
    kept = slot >= 0
    flat = (idx.long() * C + slot.long()).clamp_min(0)
    w = (gates.gather(1, idx.long().unsqueeze(1)).squeeze(1) * kept.float()).to(dt).unsqueeze(1)     # combine weight, rounded as `cw.to(dt)`
    eh, ec = (w * yh[flat]).view(B, N, D), (w * yc[flat]).view(B, N, D)
    if not ctl.use_shared_expert:
        return eh, ec, l_aux, exp_counts
    round_to = img_ids.dtype if ctl.use_rope else None
    rope0 = model._rope([img_ids, cond_ids], round_to) if ctl.use_rope else None
    cond_s, x_s = _double_block(model, "shared_expert.0", x, c, condition_temb, rope0, text_first=False)
    rope1 = model._rope([img_ids, cond_ids, txt_ids], round_to) if ctl.use_rope else None
    _, hc = _double_block(model, "shared_expert.1", torch.cat([x_s, cond_s], 1), ctrl_enc, control_temb, rope1, text_first=False)
    return hc[:, :N] + eh, hc[:, N:] + ec, l_aux, exp_counts


def flux_forward(model, hidden_states, condition_hidden_states=None, conditioning_scale: float = 1.0, encoder_hidden_states=None,
                 pooled_projections=None, condition_pooled_projections=None, timestep=None, img_ids=None, txt_ids=None, guidance=None,
                 condition_ids=None, gate_uniform=None, **_):
    """src/UniGenTransformer.py:1182-1271 (lists of conditions: :1360-1450) under autograd. Same return as the inference forward."""
    ctl, cfg = model._ctl, model.config
    multi = isinstance(condition_hidden_states, (list, tuple))
    dt = model._check_dtype(hidden_states, encoder_hidden_states)
    if txt_ids.ndim == 3:
        txt_ids = txt_ids[0]
    if img_ids.ndim == 3:
        img_ids = img_ids[0]
    BF = torch.bfloat16
    t_f32 = (timestep.to(BF) * 1000).float().contiguous()
    g_f32 = (guidance.to(BF) * 1000).float().contiguous() if (guidance is not None and cfg.guidance_embeds) else None
    pooled = pooled_projections.to(dt).contiguous()
    x = _lin(model, "x_embedder", hidden_states.to(dt))
    temb = _time_text_embed(model, "time_text_embed", t_f32, pooled, g_f32)
    enc = _lin(model, "context_embedder", encoder_hidden_states.to(dt))
    T = enc.shape[1]
    rope_base = model._rope([txt_ids, img_ids], None)
    rope_ctl_j = model._rope([img_ids, txt_ids], img_ids.dtype) if ctl.use_rope else None       # control joint blocks: [sample | context]
    rope_ctl_s = model._rope([txt_ids, img_ids], img_ids.dtype) if ctl.use_rope else None
    n_d, n_cj, n_s, n_cs = cfg.num_layers, ctl.cn_joint_layers, cfg.num_single_layers, ctl.cn_single_layers
    moe = None
    for i in range(n_d):
        enc, x = _double_block(model, f"transformer_blocks.{i}", x, enc, temb, rope_base, True)
        m = int(i / (n_d / n_cj))
        if moe is None:
            control_pooled = pooled if ctl.use_pooled_prompt_embeds else torch.zeros_like(pooled)
            control_temb = _time_text_embed(model, "control_time_text_embed", t_f32, control_pooled, g_f32)
            ctrl_enc = _lin(model, "control_context_embedder", enc)
            conds = list(zip(condition_hidden_states, condition_pooled_projections, condition_ids)) if multi else \
                [(condition_hidden_states, condition_pooled_projections, condition_ids)]
            unis = gate_uniform if isinstance(gate_uniform, (list, tuple)) else [gate_uniform] * len(conds)
            merged = merged_temb = None
            for (ct, cp, cid), uni in zip(conds, unis):
                cp = (cp.unsqueeze(0) if cp.ndim == 1 else cp).to(dt).contiguous()
                ct = (ct.unsqueeze(0) if ct.ndim == 2 else ct).to(dt)
                cid = cid[0] if cid.ndim == 3 else cid
                cond_temb = _time_text_embed(model, "control_condition_embed", t_f32, cp, g_f32)
                eh, ec, l_aux, exp_counts = _comoe(model, x, ct, ctrl_enc, control_temb, cond_temb, pooled, cp, img_ids, txt_ids, cid, uni)
                z = eh + ec
                merged = z if merged is None else merged + z
                merged_temb = cond_temb if merged_temb is None else merged_temb + cond_temb
            moe = dict(ctrl_enc=ctrl_enc, condition_temb=merged_temb, l_aux=l_aux, exp_counts=exp_counts)
            z_in = merged
        else:
            z_in = x
        _, z = _double_block(model, f"control_joint_trans_blocks.{m}", z_in, moe["ctrl_enc"], moe["condition_temb"], rope_ctl_j, text_first=False)
        x = x + _lin(model, f"controlnet_add_joint_blocks.{m}", z) * conditioning_scale
    h = torch.cat([enc, x], dim=1)
    for j in range(n_s):
        h = _single_block(model, f"single_transformer_blocks.{j}", h, temb, rope_base)
        if ctl.use_single_trans_blocks:
            m = int(j / (n_s / n_cs))
            z = _single_block(model, f"control_single_trans_blocks.{m}", h, moe["condition_temb"], rope_ctl_s)
            y = _lin(model, f"controlnet_add_single_blocks.{m}", z) * conditioning_scale
            h = h + y if ctl.single_block_control_method == "overall_add" else torch.cat([h[:, :T], h[:, T:] + y[:, T:]], dim=1)
    x = h[:, T:]
    e = _lin(model, "norm_out.linear", F.silu(temb).to(dt))                     # AdaLayerNormContinuous: (scale, shift)
    scale, shift = e.chunk(2, dim=1)
    out = _lin(model, "proj_out", A.adaln_modulate(x.contiguous(), shift, scale))
    return out, dict(moe_loss=moe["l_aux"] * 0.1), dict(expert_counts=moe["exp_counts"])
