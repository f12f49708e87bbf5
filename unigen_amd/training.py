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


def _lin_n(model, x: torch.Tensor, prefixes):
    return A.linear_n(x, [_p(model, p + ".weight") for p in prefixes], [_p(model, p + ".bias") for p in prefixes])


def _zero_res(model, prefix: str, x, z, scale: float):
    """x + Linear(z) * conditioning_scale (src/UniGenTransformer.py:1104 + :1141 / :1166-1172): one GEMM with the residual epilogue."""
    return A.linear_res_scale(x, z, _p(model, prefix + ".weight"), _p(model, prefix + ".bias"), scale)


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


def _attention(model, prefix: str, x, enc, rope, text_first: bool, mlp_prefix: Optional[str] = None):
    """Attention of a Flux block (FluxAttnProcessor2_0: [context | sample]; JointAttnRopeProcessor src/UniGenUtils.py:532-622: [sample | context]).
    RoPE rows follow the joint order, so each stream is rotated at its offset before the concatenation."""
    H = model.config.num_attention_heads
    N = x.shape[1]
    T = enc.shape[1] if enc is not None else 0
    x_off, e_off = (T, 0) if text_first else (0, N)
    # to_q | to_k | to_v read the same rows: one GEMM over the packed weights (engine._attn_qkv makes the three parameters views of one
    # buffer), one accumulated d x in the backward instead of three GEMM outputs summed by autograd's add kernels
    # single block: the SAME grouping as flux._single_block (engine._single_qkv_mlp: four-way [q | k | v | proj_mlp] when the split is tile-aligned,
    # else the three-way pack) - q, k, v are the first three row blocks either way
    if mlp_prefix is None or model._single_qkv_mlp(mlp_prefix.rsplit(".", 1)[0]) is None:
        model._attn_qkv(prefix)
    q, k, v = _lin_n(model, x, [prefix + ".to_q", prefix + ".to_k", prefix + ".to_v"])
    q = A.qk_norm_rope(q, _p(model, prefix + ".norm_q.weight"), rope, H, x_off)
    k = A.qk_norm_rope(k, _p(model, prefix + ".norm_k.weight"), rope, H, x_off)
    if enc is None:
        return A.attention(q, k, v, H), None
    model._attn_add_qkv(prefix)
    eq, ek, ev = _lin_n(model, enc, [prefix + ".add_q_proj", prefix + ".add_k_proj", prefix + ".add_v_proj"])
    eq = A.qk_norm_rope(eq, _p(model, prefix + ".norm_added_q.weight"), rope, H, e_off)
    ek = A.qk_norm_rope(ek, _p(model, prefix + ".norm_added_k.weight"), rope, H, e_off)
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
    # x + gate * a: ONE HIP kernel forward (ug_gate_residual), two backward (gate * dy, per-sample column sums of dy * a) - round 2 left these to
    # torch's mul + add kernels (at::native: 14 % of a training step with the cats and copies below)
    x = A.gate_residual(x, a, g)
    x = A.gate_residual(x, _feed_forward(model, prefix + ".ff", A.adaln_modulate(x, sh, sc)), gm)
    enc = A.gate_residual(enc, ca, cg)
    enc = A.gate_residual(enc, _feed_forward(model, prefix + ".ff_context", A.adaln_modulate(enc, csh, csc)), cgm)
    return enc, x


def _single_block_body(model, prefix: str, h, temb, rope):
    """diffusers FluxSingleTransformerBlock.forward."""
    n, gate = _adaln(model, prefix + ".norm", h, temb, 3)
    a, _ = _attention(model, prefix + ".attn", n, None, rope, True, mlp_prefix=prefix + ".proj_mlp")
    mlp = A.GeluTanh.apply(_lin(model, prefix + ".proj_mlp", n))      # its own GEMM: GELU wants a contiguous [M, 4 D] (a view of the fused output would be copied)
    # proj_out(cat[a, mlp]) without the [B, L, 5 D] concatenation: the mlp columns ride as the GEMM's K-segment extension (same bits)
    return A.gate_residual(h, A.linear_cat2(a, mlp, _p(model, prefix + ".proj_out.weight"), _p(model, prefix + ".proj_out.bias")), gate)


class _GatherRows(torch.autograd.Function):
    """y[m] = x[idx[m]] * valid[m]. A slot holds at most one token and a token sits in at most top_num <= 2 capacity slots, so the backward is an
    atomic index_add with at most two non-zero addends per row (a + b in either order: the same bits) plus zeros - torch's generic index
    backward sorts the indices first (1.1 ms per call at 4096 x 3072)."""

    @staticmethod
    def forward(ctx, x, idx, valid):
        ctx.save_for_backward(idx, valid)
        ctx.rows = x.shape[0]
        return x[idx] * valid.unsqueeze(1).to(x.dtype)

    @staticmethod
    def backward(ctx, dy):
        idx, valid = ctx.saved_tensors
        dx = torch.zeros(ctx.rows, dy.shape[1], device=dy.device, dtype=dy.dtype)
        dx.index_add_(0, idx, dy * valid.unsqueeze(1).to(dy.dtype))
        return dx, None, None


class _BroadcastSamples(torch.autograd.Function):
    """y[m] = x[sample_of[m]] for a per-sample vector table x [B, D]; backward = one masked row sum per sample (B is small; deterministic)."""

    @staticmethod
    def forward(ctx, x, sample_of):
        ctx.save_for_backward(sample_of)
        ctx.B = x.shape[0]
        return x[sample_of]

    @staticmethod
    def backward(ctx, dy):
        (sample_of,) = ctx.saved_tensors
        d32 = dy.float()
        return torch.stack([(d32 * (sample_of == b).unsqueeze(1)).sum(0) for b in range(ctx.B)]).to(dy.dtype), None


def _route(model, x, c, uniform):
    """TopKGate + top1gating / top2gating (deepspeed 0.16.5; control_params.top_num): fp32 gate softmax on (x + c); routing decisions
    (arg-max, capacity, Random Token Selection or the Gumbel-max second choice) from the HIP kernels; what is differentiable - the gate
    probabilities in the combine weights and in l_aux - stays on the tape. `uniform`: the gate's random draw (Uniform for k = 1, Gumbel for k = 2)."""
    B, N, D = x.shape
    E, S, K = model._ctl.expert_nums, B * N, getattr(model._ctl, "top_k", 1)
    dev = x.device
    wg = model.get_parameter("moe.moe_layer.gate.wg.weight")
    x2, c2 = x.reshape(S, D).contiguous(), c.reshape(S, D).contiguous()
    exp_counts, l_aux_k = torch.empty(E, device=dev, dtype=torch.int64), torch.empty(1, device=dev, dtype=torch.float32)
    if K == 1:
        C = max(int(math.ceil(S / E)), 4)
        # gate logits / softmax / arg-max and their backward: HIP kernels (round 2 ran F.linear + F.softmax here: vendor BLAS on a product path)
        gates, idx, _ = A.MoeGate.apply(x2, c2, wg)
        if uniform is None:
            uniform = torch.rand(S, E, device=dev, dtype=torch.float32)
        slot, tos = torch.empty(S, device=dev, dtype=torch.int32), torch.empty(E, C, device=dev, dtype=torch.int32)
        ops.moe_capacity_rts(gates.detach().contiguous(), idx, uniform.contiguous(), C, slot, tos, exp_counts, l_aux_k)
        l_aux = torch.sum(gates.mean(0) * F.one_hot(idx.long(), E).float().mean(0)) * E
        kept = (slot >= 0).unsqueeze(0)
        flat = (idx.long() * C + slot.long()).clamp_min(0).unsqueeze(0)
        w = (gates.gather(1, idx.long().unsqueeze(1)).squeeze(1) * kept[0].float()).to(x.dtype).unsqueeze(0)      # combine weight, rounded as `cw.to(dt)`
    elif K > 2:                                          # deepspeed topkgating: no random draw; the capacity rule ranks the chosen LOGITS against zeros
        C = max(int(math.ceil((S / E) * float(K))), 4)
        gates, idx, logits = A.MoeGate.apply(x2, c2, wg, K, None)
        slot, tos = torch.empty(K, S, device=dev, dtype=torch.int32), torch.empty(E, C, device=dev, dtype=torch.int32)
        w_dev = torch.empty(K, S, device=dev, dtype=torch.float32)
        ops.moe_capacity_topk(gates.detach().contiguous(), logits, idx, C, slot, tos, w_dev, exp_counts, l_aux_k)
        chosen = torch.zeros(S, E, device=dev, dtype=torch.float32).scatter_(1, idx.long().t(), 1.0)
        l_aux = torch.mean(gates.mean(0) * chosen.mean(0)) * E * E / K
        kept = slot >= 0
        flat = (idx.long() * C + slot.long()).clamp_min(0)
        g = torch.stack([gates.gather(1, idx[k].long().unsqueeze(1)).squeeze(1) for k in range(K)]) * kept.float()
        w = (g / torch.clamp(g.sum(0, keepdim=True), min=torch.finfo(torch.float32).eps)).to(x.dtype)
    else:
        C = max(int(math.ceil((S / E) * 2.0)), 4)
        if uniform is None:                            # gumbel_rsample
            uniform = torch.distributions.gumbel.Gumbel(torch.tensor(0.0, device=dev), torch.tensor(1.0, device=dev)).rsample((S, E))
        gates, idx, _ = A.MoeGate.apply(x2, c2, wg, 2, uniform.to(torch.float32).contiguous())
        slot, tos = torch.empty(2, S, device=dev, dtype=torch.int32), torch.empty(E, C, device=dev, dtype=torch.int32)
        w_dev = torch.empty(2, S, device=dev, dtype=torch.float32)      # the kernel's weights: not on the tape, recomputed below from `gates`
        ops.moe_capacity_top2(gates.detach().contiguous(), idx, C, slot, tos, w_dev, exp_counts, l_aux_k)
        l_aux = torch.mean(gates.mean(0) * F.one_hot(idx[0].long(), E).float().mean(0)) * E * E
        kept = slot >= 0
        flat = (idx.long() * C + slot.long()).clamp_min(0)
        g = torch.stack([gates.gather(1, idx[k].long().unsqueeze(1)).squeeze(1) for k in range(2)]) * kept.float()
        w = (g / torch.clamp(g.sum(0, keepdim=True), min=torch.finfo(torch.float32).eps)).to(x.dtype)             # gates1_s / denom_s, gates2_s / denom_s
    tos_l = tos.view(-1).long()
    valid, src = tos_l >= 0, tos_l.clamp_min(0)
    dispatch = lambda t2d: _GatherRows.apply(t2d, src, valid).view(E, C, -1)                    # einsum("sec,sm->ecm") with a one-hot mask

    def combine(y):                                                                             # einsum("sec,ecm->sm"): fp32 sum, one rounding
        y2 = y.reshape(E * C, D)
        if K == 1:
            return (w[0].unsqueeze(1) * _GatherRows.apply(y2, flat[0], kept[0])).view(B, N, D)
        acc = sum(w[k].float().unsqueeze(1) * _GatherRows.apply(y2, flat[k], kept[k]).float() for k in range(K))
        return acc.to(y.dtype).view(B, N, D)

    return dispatch, combine, src // N, l_aux, exp_counts, E, C


def _experts_modulated(model, x, c, pooled, cond_pooled, uniform):
    """expert_forward with modulated linears (src/UniGenTransformer.py:957-959): c' = W_c (s_c * c) + b_c ; h' = W_h (s_h * (h + c')) + b_h."""
    B, N, D = x.shape
    dispatch, combine, sample_of, l_aux, exp_counts, E, C = _route(model, x, c, uniform)
    xd, cd = dispatch(x.reshape(B * N, D)), dispatch(c.reshape(B * N, D))
    pe = "moe.moe_layer.experts.deepspeed_experts."
    yh, yc = [], []
    for e in range(E):
        rows = slice(e * C, (e + 1) * C)
        s_c = _BroadcastSamples.apply(_lin(model, f"{pe}{e}.0.1", cond_pooled), sample_of[rows])     # the modulation vector of each slot's sample
        s_h = _BroadcastSamples.apply(_lin(model, f"{pe}{e}.1.1", pooled), sample_of[rows])
        c_e = _lin(model, f"{pe}{e}.0.0", s_c * cd[e])
        h_e = _lin(model, f"{pe}{e}.1.0", s_h * (xd[e] + c_e))
        yh.append(h_e); yc.append(c_e)
    return combine(torch.stack(yh)), combine(torch.stack(yc)), l_aux, exp_counts


def _comoe(model, x, cond_tokens, ctrl_enc, control_temb, condition_temb, pooled, cond_pooled, img_ids, txt_ids, cond_ids, uniform):
    """preprocess_moe_forward + moe_forward for one condition (src/UniGenTransformer.py:969-1068)."""
    ctl = model._ctl
    N = x.shape[1]
    c = _lin(model, "control_x_embedder", cond_tokens)
    eh, ec, l_aux, exp_counts = _experts_modulated(model, x, c, pooled, cond_pooled, uniform)
    round_to = img_ids.dtype if ctl.use_rope else None
    if ctl.use_consis_module and ctl.use_shared_expert:     # "V2" consistency module (:984-1004): consis_module[0] twice, sample-stream outputs only;
        # without shared experts the reference discards its result (the return tuple is rebuilt only at :1024) - tests/test_ref_wiring_cpu.py
        rope_a = model._rope([cond_ids, cond_ids], round_to) if ctl.use_rope else None
        _, cech = _double_block(model, "consis_module.0", ec, c, condition_temb, rope_a, text_first=False)
        rope_b = model._rope([img_ids, cond_ids, img_ids], round_to) if ctl.use_rope else None
        _, y = _double_block(model, "consis_module.0", torch.cat([eh, cech], 1), x, control_temb, rope_b, text_first=False)
        eh, ec = eh + y[:, :N], ec + y[:, N:]
    if not ctl.use_shared_expert:
        return eh, ec, l_aux, exp_counts
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
        x = _zero_res(model, f"controlnet_add_joint_blocks.{m}", x, z, conditioning_scale)
    h = torch.cat([enc, x], dim=1)
    for j in range(n_s):
        h = _single_block(model, f"single_transformer_blocks.{j}", h, temb, rope_base)
        if ctl.use_single_trans_blocks:
            m = int(j / (n_s / n_cs))
            z = _single_block(model, f"control_single_trans_blocks.{m}", h, moe["condition_temb"], rope_ctl_s)
            if ctl.single_block_control_method == "overall_add":
                h = _zero_res(model, f"controlnet_add_single_blocks.{m}", h, z, conditioning_scale)
            else:
                y = _lin(model, f"controlnet_add_single_blocks.{m}", z) * conditioning_scale
                h = torch.cat([h[:, :T], h[:, T:] + y[:, T:]], dim=1)
    x = h[:, T:]
    e = _lin(model, "norm_out.linear", F.silu(temb).to(dt))                     # AdaLayerNormContinuous: (scale, shift)
    scale, shift = e.chunk(2, dim=1)
    out = _lin(model, "proj_out", A.adaln_modulate(x.contiguous(), shift, scale))
    return out, dict(moe_loss=moe["l_aux"] * 0.1), dict(expert_counts=moe["exp_counts"])


# ---------------------------------------------------------------------------------------------------------------------
# UniGenSD3 (src/UniGenTransformer.py:625-710; blocks as restated by the reference at src/UniGenUtils.py:340-522)
# ---------------------------------------------------------------------------------------------------------------------
def _gate(g, y):
    return g * y if g.dim() == y.dim() else g.unsqueeze(1) * y


def _gres(x, g, y):
    """x + gate * y: per-sample gates [B, D] through ug_gate_residual, per-token gates (the experts' [B, L, D] tembs) stay elementwise."""
    return x + g * y if g.dim() == y.dim() else A.gate_residual(x, y, g.contiguous())


def _mod(x, shift, scale):
    """LayerNorm(x) * (1 + scale) + shift with per-sample [B, D] or per-token [B, L, D] modulation (the reference's extension, :354-363)."""
    if scale.dim() == 3:
        B, Ls, D = x.shape
        return A.adaln_modulate(x.reshape(B * Ls, 1, D), shift.reshape(B * Ls, D), scale.reshape(B * Ls, D)).view(B, Ls, D)
    return A.adaln_modulate(x, shift, scale)


def _sd3_attention(model, prefix: str, x, enc, context_pre_only: bool = False):
    """Attention + JointAttnProcessor2_0 (sample-first concat, no RoPE); q/k RMSNorm when the weights exist."""
    H, N = model._heads, x.shape[1]
    nrm = lambda t, name: A.qk_norm_rope(t, _p(model, f"{prefix}.{name}.weight"), None, H)
    q, k, v = nrm(_lin(model, prefix + ".to_q", x), "norm_q"), nrm(_lin(model, prefix + ".to_k", x), "norm_k"), _lin(model, prefix + ".to_v", x)
    if enc is None:
        return _lin(model, prefix + ".to_out.0", A.attention(q, k, v, H)), None
    eq, ek = nrm(_lin(model, prefix + ".add_q_proj", enc), "norm_added_q"), nrm(_lin(model, prefix + ".add_k_proj", enc), "norm_added_k")
    ev = _lin(model, prefix + ".add_v_proj", enc)
    o = A.attention(torch.cat([q, eq], 1), torch.cat([k, ek], 1), torch.cat([v, ev], 1), H)
    xo = _lin(model, prefix + ".to_out.0", o[:, :N])
    return xo, (None if context_pre_only else _lin(model, prefix + ".to_add_out", o[:, N:]))


def _sd3_joint_block(model, prefix: str, x, enc, temb, context_pre_only: bool = False, dual: bool = False):
    """JointTransformerBlock.forward as restated by the reference (src/UniGenUtils.py:440-522). Returns (enc or None, x)."""
    ch = _lin(model, prefix + ".norm1.linear", F.silu(temb)).chunk(9 if dual else 6, dim=-1)
    n = _mod(x, ch[0], ch[1])
    g, shm, scm, gm = ch[2:6]
    if context_pre_only:                                        # AdaLayerNormContinuous: (scale, shift)
        e = _lin(model, prefix + ".norm1_context.linear", F.silu(temb).to(enc.dtype)).chunk(2, dim=-1)
        nc = _mod(enc, e[1], e[0])
    else:
        cc = _lin(model, prefix + ".norm1_context.linear", F.silu(temb)).chunk(6, dim=-1)
        nc = _mod(enc, cc[0], cc[1])
    a, ca = _sd3_attention(model, prefix + ".attn", n, nc, context_pre_only)
    if dual:                                                    # SD35AdaLayerNormZeroX: norm_hidden_states2 from the block's INPUT
        a2, _ = _sd3_attention(model, prefix + ".attn2", _mod(x, ch[6], ch[7]), None)
    x = _gres(x, g, a)
    if dual:
        x = _gres(x, ch[8], a2)
    x = _gres(x, gm, _feed_forward(model, prefix + ".ff", _mod(x, shm, scm)))
    if context_pre_only:
        return None, x
    enc = _gres(enc, cc[2], ca)
    enc = _gres(enc, cc[5], _feed_forward(model, prefix + ".ff_context", _mod(enc, cc[3], cc[4])))
    return enc, x


def _sd3_single_block(model, prefix: str, x, temb):
    """SD3SingleTransformerBlock.forward (src/UniGenUtils.py:386-414); temb per sample [B, D] or per token [B, L, D]."""
    ch = _lin(model, prefix + ".norm1.linear", F.silu(temb)).chunk(6, dim=-1)
    a, _ = _sd3_attention(model, prefix + ".attn", _mod(x, ch[0], ch[1]), None)
    x = _gres(x, ch[2], a)
    return _gres(x, ch[5], _feed_forward(model, prefix + ".ff", _mod(x, ch[3], ch[4])))


def _sd3_patch_embed(model, prefix: str, latents):
    """PatchEmbed: Conv2d(k = s = patch) as a GEMM over unfolded patches, + the centre-cropped sincos table (fp32 add, cast back)."""
    cfg, D = model.config, model.inner_dim
    B, C, Hh, Ww = latents.shape
    p = cfg.patch_size
    h, w = Hh // p, Ww // p
    patches = latents.view(B, C, h, p, w, p).permute(0, 2, 4, 1, 3, 5).reshape(B, h * w, C * p * p)
    x = A.linear(patches, model.get_parameter(prefix + ".proj.weight").view(D, C * p * p), model.get_parameter(prefix + ".proj.bias"))
    mx = cfg.pos_embed_max_size
    top, left = (mx - h) // 2, (mx - w) // 2
    tab = getattr(model, prefix).pos_embed.reshape(mx, mx, D)[top:top + h, left:left + w].reshape(1, h * w, D).float()
    return (x + tab).to(x.dtype)


def _sd3_comoe(model, x, c, ctrl_enc, control_temb, condition_temb, pooled, cond_pooled, uniform):
    """UniGenBase.moe_forward + expert_forward (src/UniGenTransformer.py:225-296): experts = modulated linears, or two SD3 single blocks per
    expert fed the dispatched per-token tembs and attending over the expert's capacity slots; then the two shared joint blocks."""
    ctl = model._ctl
    B, N, D = x.shape
    if ctl.modulated:
        eh, ec, l_aux, exp_counts = _experts_modulated(model, x, c, pooled, cond_pooled, uniform)
    else:
        dispatch, combine, sample_of, l_aux, exp_counts, E, C = _route(model, x, c, uniform)
        xd, cd = dispatch(x.reshape(B * N, D)), dispatch(c.reshape(B * N, D))
        td, ctd = dispatch(control_temb[:, None].expand(-1, N, -1).reshape(B * N, D)), dispatch(condition_temb[:, None].expand(-1, N, -1).reshape(B * N, D))
        pe = "moe.moe_layer.experts.deepspeed_experts."
        yh = [_sd3_single_block(model, f"{pe}{e}.0", xd[e][None], td[e][None])[0] for e in range(E)]
        yc = [_sd3_single_block(model, f"{pe}{e}.1", cd[e][None], ctd[e][None])[0] for e in range(E)]
        eh, ec = combine(torch.stack(yh)), combine(torch.stack(yc))
    if not ctl.use_shared_expert:
        return eh, ec, l_aux, exp_counts
    cond_s, x_s = _sd3_joint_block(model, "shared_expert.0", x, c, condition_temb)
    _, hc = _sd3_joint_block(model, "shared_expert.1", torch.cat([x_s, cond_s], 1), ctrl_enc, control_temb, context_pre_only=True, dual=True)
    return hc[:, :N] + eh, hc[:, N:] + ec, l_aux, exp_counts


def sd3_forward(model, hidden_states, condition_hidden_states=None, conditioning_scale: float = 1.0, encoder_hidden_states=None,
                pooled_projections=None, condition_pooled_projections=None, timestep=None, gate_uniform=None, **_):
    """UniGenSD3.forward (src/UniGenTransformer.py:625-710) under autograd: NCHW latents in, NCHW out; the timestep is used as given."""
    cfg, ctl = model.config, model._ctl
    dt = model._check_dtype(hidden_states, encoder_hidden_states)
    B, _, height, width = hidden_states.shape
    p, Lyr = cfg.patch_size, cfg.num_layers
    h, w = height // p, width // p
    t_f32 = timestep.to(hidden_states.device).float().expand(B).contiguous()
    pooled, cpooled = pooled_projections.to(dt).contiguous(), condition_pooled_projections.to(dt).contiguous()
    x = _sd3_patch_embed(model, "pos_embed", hidden_states.to(dt))
    temb = _time_text_embed(model, "time_text_embed", t_f32, pooled, None)
    enc = _lin(model, "context_embedder", encoder_hidden_states.to(dt))
    moe = None
    for i in range(Lyr):
        last, dual = i == Lyr - 1, i in cfg.dual_attention_layers
        enc_new, x = _sd3_joint_block(model, f"transformer_blocks.{i}", x, enc, temb, context_pre_only=last, dual=dual)
        if moe is None:
            c = _sd3_patch_embed(model, "control_pos_embed_input", condition_hidden_states.to(dt))
            control_pooled = pooled if ctl.use_pooled_prompt_embeds else torch.zeros_like(pooled)
            control_temb = _time_text_embed(model, "control_time_text_embed", t_f32, control_pooled, None)
            condition_temb = _time_text_embed(model, "control_condition_embed", t_f32, cpooled, None)
            ctrl_enc = _lin(model, "control_context_embedder", enc_new)
            eh, ec, l_aux, exp_counts = _sd3_comoe(model, x, c, ctrl_enc, control_temb, condition_temb, pooled, cpooled, gate_uniform)
            moe = dict(ctrl_enc=ctrl_enc, condition_temb=condition_temb, l_aux=l_aux, exp_counts=exp_counts)
            z_in = eh + ec
        else:
            z_in = x
        enc = enc_new
        _, z = _sd3_joint_block(model, f"control_transformer_blocks.{i}", z_in, moe["ctrl_enc"], moe["condition_temb"], dual=dual)
        x = _zero_res(model, f"controlnet_add_blocks.{i}", x, z, conditioning_scale)
    e = _lin(model, "norm_out.linear", F.silu(temb).to(dt)).chunk(2, dim=1)
    out = _lin(model, "proj_out", A.adaln_modulate(x.contiguous(), e[1], e[0]))
    out = out.view(B, h, w, p, p, model.out_channels).permute(0, 5, 1, 3, 2, 4).reshape(B, model.out_channels, h * p, w * p)
    return out, dict(moe_loss=moe["l_aux"] * 0.1), dict(expert_counts=moe["exp_counts"])
