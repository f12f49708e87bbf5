"""CPU oracle: a plain-PyTorch restatement of UniGen's condition-weaving + expert-modulation forward pass.

TEST INFRASTRUCTURE ONLY. Nothing in the product path (unigen_amd/, src/) imports this module; only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg do, as the checker / the timed CPU baseline.

PARITY: REFERENCE-PINNED WHEREVER THE REFERENCE'S OWN CODE CAN EXECUTE WITH TORCH ALONE (rounds 4-5); the rest is restated, PARITY UNPINNED.
The reference (gavin-gqzhang/UniGen @ /root/reference) ships no tests, golden vectors or fixtures, it cannot be imported here (deepspeed /
diffusers / peft / ipdb are absent: ordinary ModuleNotFoundError) and `UniGenFlux` uses classes that are defined nowhere
(FluxJointRoPETransformerBlock, FluxSingleRoPETransformerBlock). Its self-contained torch-only definitions ARE run: tests/golden/ref_harness.py
compiles them from the reference's files at run time (build container only; sha256-pinned sources, a namespace of torch symbols, a builtin
whitelist, every free name checked), the make_ref_*_golden.py generators write their outputs as tensors, and
tests/test_ref_leaf_cpu.py::test_committed_reference_fixtures_regenerate_bit_identically re-runs the generators against the committed files.
Pinned rows of this file (22 reference definitions):
  leaves (tests/test_ref_leaf_cpu.py, ref_leaf.safetensors)
    modulated_flatten_literal / modulated_linear  <- modulated_flatten          src/UniGenUtils.py:204-228
    adaln_zero / adaln_zero_any                   <- adanorm_forward            src/UniGenUtils.py:354-363
    adaln_zero_x                                  <- sd35adanormX_forward       src/UniGenUtils.py:340-352
    adaln_continuous                              <- adanormContinuous_forward  src/UniGenUtils.py:365-373
    (zero-res init)                               <- zero_module                src/UniGenUtils.py:194-197
    expert_forward                                <- UniGenFlux.expert_forward  src/UniGenTransformer.py:925-967 == UniGenBase :225-267
    sd3_attention (sample-first joint attention:  <- JointAttnRopeProcessor.__call__  src/UniGenUtils.py:533-622, run without RoPE and q/k norms
     concatenation order, split, projections)        (those two branches need diffusers symbols and stay restated)
  block bodies (tests/test_ref_wiring_cpu.py, ref_blocks.safetensors)
    sd3_joint_block                               <- JointTransformerBlock.forward      src/UniGenUtils.py:438-522 (plain / context_pre_only / dual / both)
    sd3_single_block                              <- SD3SingleTransformerBlock.forward  src/UniGenUtils.py:386-414 (per-sample, per-token, expert call)
  wiring (tests/test_ref_wiring_cpu.py, ref_wiring.safetensors; the forwards below run THROUGH these functions)
    flux_moe_forward                              <- UniGenFlux.moe_forward :969-1026, UniGenBase.moe_forward :269-296
    flux_preprocess_moe_forward, flux_base_forward (+ control)  <- UniGenFlux.preprocess_moe_forward :1028, .control_forward :1070, .base_forward :1106;
                                                     MultiCondtionUniGenFlux.preprocess_moe_forward :1275, .control_forward :1324
    sd3_preprocess_moe_forward, sd3_base_forward  <- UniGenSD3.preprocess_moe_forward :498, .control_forward :539, .base_forward :581
  pipelines (tests/test_host_cpu.py, ref_pipeline.safetensors): prepare_image of both pipelines  src/UniGenPipeline.py:107-141, :457-483
UNPINNED (restated from the published algorithms; the source is in packages that are absent here, pinned in /root/reference/environment.yaml):
  diffusers==0.32.2 (FluxTransformerBlock, FluxSingleTransformerBlock, FluxAttnProcessor2_0, AdaLayerNormZero*, RMSNorm, FluxPosEmbed,
  apply_rotary_emb, CombinedTimestepTextProjEmbeddings, PatchEmbed, FlowMatchEulerDiscreteScheduler), deepspeed==0.16.5 (sharded_moe.top1gating /
  top2gating / topkgating / TopKGate, and MOELayer.forward's einsums behind src/UniGenUtils.py:74-191), peft==0.15.0 (LoRA Linear); and the
  top-level forwards / pipeline __call__s of the reference (src/UniGenTransformer.py:1182-1271, :625-710, src/UniGenPipeline.py:143-455, :486-807),
  which reach those packages on live paths. These rows are held against drift only by fixtures this file generated itself (tests/golden/make_golden.py).

Everything runs on a flat `state` dict {reference state-dict key: tensor}. `dtype` chooses the arithmetic:
  torch.bfloat16 -> the reference's own eager rounding points (every torch op rounds to bf16),
  torch.float32  -> the same graph on bf16-rounded weights without intermediate rounding ("fp32 truth").
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

State = Dict[str, torch.Tensor]


@dataclass
class FluxConfig:
    """diffusers FluxTransformer2DModel config + UniGen control params (config/unigen.yaml)."""
    in_channels: int = 64
    num_layers: int = 19
    num_single_layers: int = 38
    attention_head_dim: int = 128
    num_attention_heads: int = 24
    joint_attention_dim: int = 4096
    pooled_projection_dim: int = 768
    guidance_embeds: bool = False
    axes_dims_rope: Tuple[int, int, int] = (16, 56, 56)
    # control params (src/UniGenTransformer.py:720-783, 807-893)
    condition_nums: int = 1
    use_rope: bool = True
    use_pooled_prompt_embeds: bool = True
    use_shared_expert: bool = True
    use_consis_module: bool = False
    use_single_trans_blocks: bool = True
    single_control_dev: int = 2
    single_block_control_method: str = "overall_add"
    expert_num_each_condition: int = 3
    expert_num: Optional[int] = None
    top_num: int = 1

    @property
    def inner_dim(self) -> int:
        return self.num_attention_heads * self.attention_head_dim

    @property
    def expert_nums(self) -> int:  # src/UniGenTransformer.py:807
        return self.expert_num if self.expert_num is not None else (self.condition_nums + 1) * self.expert_num_each_condition

    @property
    def cn_joint_layers(self) -> int:  # :744
        return self.num_layers // self.single_control_dev

    @property
    def cn_single_layers(self) -> int:
        return self.num_single_layers // self.single_control_dev


# ---------------------------------------------------------------------------------------------------------------------
# primitives (diffusers 0.32.2 semantics)
# ---------------------------------------------------------------------------------------------------------------------

LORA_KEY = "__lora__"   # optional entry of a state: {projection prefix: [(A [r, K], B [N, r], scaling), ...]} - the ACTIVE PEFT adapters of that Linear


def linear(state: State, prefix: str, x: torch.Tensor) -> torch.Tensor:
    """nn.Linear at `prefix`; when the state carries LoRA adapters for it (LORA_KEY) the projection is peft 0.15's LoRA Linear (lora_linear
    below): what `enable_lora` (src/lora_switching_module.py:11-38) switches per condition by setting the scalings of the others to 0."""
    w = state[prefix + ".weight"].to(x.dtype)
    b = state.get(prefix + ".bias")
    b = None if b is None else b.to(x.dtype)
    ad = state.get(LORA_KEY)
    if ad and prefix in ad:
        return lora_linear(x, w, b, [(A.to(x.dtype), Bm.to(x.dtype), sc) for A, Bm, sc in ad[prefix]])
    return F.linear(x, w, b)


def timestep_sinusoid(t: torch.Tensor, dim: int = 256) -> torch.Tensor:
    """get_timestep_embedding(flip_sin_to_cos=True, downscale_freq_shift=0, scale=1, max_period=10000) -> fp32 [B, dim]."""
    half = dim // 2
    exponent = -math.log(10000) * torch.arange(0, half, dtype=torch.float32) / half
    emb = t[:, None].float() * torch.exp(exponent)[None, :]
    emb = torch.cat([torch.sin(emb), torch.cos(emb)], dim=-1)
    return torch.cat([emb[:, half:], emb[:, :half]], dim=-1)  # flip -> [cos | sin]


def time_text_embed(state: State, prefix: str, timestep: torch.Tensor, pooled: torch.Tensor,
                    guidance: Optional[torch.Tensor] = None) -> torch.Tensor:
    """CombinedTimestep(Guidance)TextProjEmbeddings: timestep_embedder(time_proj(t)) [+ guidance_embedder] + text_embedder(pooled)."""
    dt = pooled.dtype
    tp = timestep_sinusoid(timestep).to(dt)
    temb = linear(state, prefix + ".timestep_embedder.linear_2", F.silu(linear(state, prefix + ".timestep_embedder.linear_1", tp)))
    if guidance is not None:
        gp = timestep_sinusoid(guidance).to(dt)
        temb = temb + linear(state, prefix + ".guidance_embedder.linear_2", F.silu(linear(state, prefix + ".guidance_embedder.linear_1", gp)))
    pe = linear(state, prefix + ".text_embedder.linear_2", F.silu(linear(state, prefix + ".text_embedder.linear_1", pooled)))
    return temb + pe


def flux_pos_embed(ids: torch.Tensor, axes_dim: Sequence[int], theta: float = 10000.0) -> Tuple[torch.Tensor, torch.Tensor]:
    """FluxPosEmbed.forward: per axis get_1d_rotary_pos_embed(repeat_interleave_real=True, use_real=True, float64 freqs) -> fp32 cos, sin [S, sum(axes)]."""
    cos_out, sin_out = [], []
    pos = ids.float()
    for i, d in enumerate(axes_dim):
        freqs = 1.0 / (theta ** (torch.arange(0, d, 2, dtype=torch.float64) / d))
        ang = torch.outer(pos[:, i].to(torch.float64), freqs)
        cos_out.append(ang.cos().repeat_interleave(2, dim=1).float())
        sin_out.append(ang.sin().repeat_interleave(2, dim=1).float())
    return torch.cat(cos_out, dim=-1), torch.cat(sin_out, dim=-1)


def apply_rotary_emb(x: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor) -> torch.Tensor:
    """diffusers apply_rotary_emb(use_real=True, use_real_unbind_dim=-1) on [B, H, S, dh]."""
    cos, sin = cos[None, None], sin[None, None]
    xr, xi = x.reshape(*x.shape[:-1], -1, 2).unbind(-1)
    x_rot = torch.stack([-xi, xr], dim=-1).flatten(3)
    return (x.float() * cos + x_rot.float() * sin).to(x.dtype)


def rms_norm(x: torch.Tensor, weight: Optional[torch.Tensor], eps: float = 1e-6) -> torch.Tensor:
    """diffusers RMSNorm.forward: fp32 statistics, x * rsqrt(var + eps) in fp32, cast to the weight dtype, * weight."""
    dt = x.dtype
    var = x.to(torch.float32).pow(2).mean(-1, keepdim=True)
    x = x * torch.rsqrt(var + eps)
    if weight is not None:
        w = weight.to(dt)
        if w.dtype in (torch.float16, torch.bfloat16):
            x = x.to(w.dtype)
        x = x * w
    else:
        x = x.to(dt)
    return x


def layer_norm(x: torch.Tensor, eps: float = 1e-6) -> torch.Tensor:
    return F.layer_norm(x, (x.shape[-1],), None, None, eps)


def adaln_zero(state: State, prefix: str, x: torch.Tensor, emb: torch.Tensor):
    """AdaLayerNormZero.forward -> (x_mod, gate_msa, shift_mlp, scale_mlp, gate_mlp)."""
    e = linear(state, prefix + ".linear", F.silu(emb))
    shift_msa, scale_msa, gate_msa, shift_mlp, scale_mlp, gate_mlp = e.chunk(6, dim=1)
    x = layer_norm(x) * (1 + scale_msa[:, None]) + shift_msa[:, None]
    return x, gate_msa, shift_mlp, scale_mlp, gate_mlp


def adaln_zero_single(state: State, prefix: str, x: torch.Tensor, emb: torch.Tensor):
    e = linear(state, prefix + ".linear", F.silu(emb))
    shift_msa, scale_msa, gate_msa = e.chunk(3, dim=1)
    x = layer_norm(x) * (1 + scale_msa[:, None]) + shift_msa[:, None]
    return x, gate_msa


def adaln_continuous(state: State, prefix: str, x: torch.Tensor, emb: torch.Tensor) -> torch.Tensor:
    e = linear(state, prefix + ".linear", F.silu(emb).to(x.dtype))
    scale, shift = torch.chunk(e, 2, dim=1)
    return layer_norm(x) * (1 + scale)[:, None, :] + shift[:, None, :]


def feed_forward(state: State, prefix: str, x: torch.Tensor) -> torch.Tensor:
    """FeedForward(activation_fn='gelu-approximate'): net.0.proj -> gelu(tanh) -> net.2."""
    return linear(state, prefix + ".net.2", F.gelu(linear(state, prefix + ".net.0.proj", x), approximate="tanh"))


def _heads(x: torch.Tensor, H: int) -> torch.Tensor:
    B, L, D = x.shape
    return x.view(B, L, H, D // H).transpose(1, 2)


def _sdpa(q, k, v):
    return F.scaled_dot_product_attention(q, k, v, dropout_p=0.0, is_causal=False)


def joint_attention(state: State, prefix: str, cfg: FluxConfig, x: torch.Tensor, enc: Optional[torch.Tensor],
                    rope: Optional[Tuple[torch.Tensor, torch.Tensor]], text_first: bool):
    """Attention module of a Flux block.
    text_first=True : FluxAttnProcessor2_0 (cat([context, sample]); rope rows follow that order).
    text_first=False: JointAttnRopeProcessor src/UniGenUtils.py:532-622 (cat([sample, context])).
    Returns (sample_out, context_out) after to_out[0] / to_add_out; for enc=None returns the raw attention (single block)."""
    H = cfg.num_attention_heads
    q = rms_norm(_heads(linear(state, prefix + ".to_q", x), H), state[prefix + ".norm_q.weight"])
    k = rms_norm(_heads(linear(state, prefix + ".to_k", x), H), state[prefix + ".norm_k.weight"])
    v = _heads(linear(state, prefix + ".to_v", x), H)
    if enc is not None:
        eq = rms_norm(_heads(linear(state, prefix + ".add_q_proj", enc), H), state[prefix + ".norm_added_q.weight"])
        ek = rms_norm(_heads(linear(state, prefix + ".add_k_proj", enc), H), state[prefix + ".norm_added_k.weight"])
        ev = _heads(linear(state, prefix + ".add_v_proj", enc), H)
        if text_first:
            q, k, v = torch.cat([eq, q], 2), torch.cat([ek, k], 2), torch.cat([ev, v], 2)
        else:
            q, k, v = torch.cat([q, eq], 2), torch.cat([k, ek], 2), torch.cat([v, ev], 2)
    if rope is not None:
        q, k = apply_rotary_emb(q, *rope), apply_rotary_emb(k, *rope)
    o = _sdpa(q, k, v)
    B = x.shape[0]
    o = o.transpose(1, 2).reshape(B, -1, cfg.inner_dim).to(q.dtype)
    if enc is None:
        return o, None
    T, N = enc.shape[1], x.shape[1]
    if text_first:
        eo, xo = o[:, :T], o[:, T:]
    else:
        xo, eo = o[:, :N], o[:, N:]
    return linear(state, prefix + ".to_out.0", xo), linear(state, prefix + ".to_add_out", eo)


def flux_double_block(state: State, prefix: str, cfg: FluxConfig, x, enc, temb, rope, text_first: bool = True):
    """diffusers FluxTransformerBlock.forward. Returns (enc, x)."""
    n, g, sh, sc, gm = adaln_zero(state, prefix + ".norm1", x, temb)
    nc, cg, csh, csc, cgm = adaln_zero(state, prefix + ".norm1_context", enc, temb)
    a, ca = joint_attention(state, prefix + ".attn", cfg, n, nc, rope, text_first)
    x = x + g.unsqueeze(1) * a
    n2 = layer_norm(x) * (1 + sc[:, None]) + sh[:, None]
    x = x + gm.unsqueeze(1) * feed_forward(state, prefix + ".ff", n2)
    enc = enc + cg.unsqueeze(1) * ca
    nc2 = layer_norm(enc) * (1 + csc[:, None]) + csh[:, None]
    enc = enc + cgm.unsqueeze(1) * feed_forward(state, prefix + ".ff_context", nc2)
    return enc, x


def flux_single_block(state: State, prefix: str, cfg: FluxConfig, h, temb, rope):
    """diffusers FluxSingleTransformerBlock.forward."""
    n, gate = adaln_zero_single(state, prefix + ".norm", h, temb)
    mlp = F.gelu(linear(state, prefix + ".proj_mlp", n), approximate="tanh")
    a, _ = joint_attention(state, prefix + ".attn", cfg, n, None, rope, True)
    out = linear(state, prefix + ".proj_out", torch.cat([a, mlp], dim=2))
    return h + gate.unsqueeze(1) * out


def control_rope(cfg: FluxConfig, ids: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """rope_embed(ids) cast to the ids dtype as JointAttnRopeProcessor does (src/UniGenUtils.py:597); the pipeline builds
    ids in the latent dtype (src/UniGenPipeline.py:640) so the control-path tables are rounded to that dtype."""
    cos, sin = flux_pos_embed(ids, cfg.axes_dims_rope)
    return cos.to(ids.dtype), sin.to(ids.dtype)


# ---------------------------------------------------------------------------------------------------------------------
# CoMoE: deepspeed 0.16.5 top1gating + src/UniGenUtils.py:74-191 + src/UniGenTransformer.py:925-1026
# ---------------------------------------------------------------------------------------------------------------------

def moe_capacity(S: int, E: int, capacity_factor: float = 1.0, min_capacity: int = 4) -> int:
    return max(int(math.ceil((S / E) * capacity_factor)), min_capacity)


def top1gating(logits: torch.Tensor, uniform: torch.Tensor, capacity: int):
    """deepspeed.moe.sharded_moe.top1gating(use_rts=True, drop_tokens=True), with the Uniform(0,1) sample passed in.
    Returns l_aux, combine_weights [S,E,C] fp32, dispatch_mask [S,E,C] bool, exp_counts [E] int64."""
    S, E = logits.shape
    gates = F.softmax(logits, dim=1)
    idx = torch.argmax(gates, dim=1)
    mask1 = F.one_hot(idx, num_classes=E)
    exp_counts = torch.sum(mask1, dim=0).detach()
    me = torch.mean(gates, dim=0)
    ce = torch.mean(mask1.float(), dim=0)
    l_aux = torch.sum(me * ce) * E
    mask1_rand = mask1 * uniform
    top_idx = torch.topk(mask1_rand, k=capacity, dim=0)[1]
    mask1 = mask1 * torch.zeros_like(mask1).scatter_(0, top_idx, 1)
    locations1 = torch.cumsum(mask1, dim=0) - 1
    locations1_s = torch.sum(locations1 * mask1, dim=1)
    mask1_float = mask1.float()
    gates = gates * mask1_float
    locations1_sc = F.one_hot(locations1_s, num_classes=capacity).float()
    combine_weights = torch.einsum("se,sc->sec", gates, locations1_sc)
    dispatch_mask = combine_weights.bool()
    return l_aux, combine_weights, dispatch_mask, exp_counts


def routing_from_gates(gates: torch.Tensor, uniform: torch.Tensor, capacity: int):
    """Index form of top1gating: per token (expert idx, slot or -1). Used to check the dense form and the HIP kernels."""
    S, E = gates.shape
    idx = torch.argmax(gates, dim=1)
    slot = torch.full((S,), -1, dtype=torch.int64)
    token_of_slot = torch.full((E, capacity), -1, dtype=torch.int64)
    for e in range(E):
        toks = torch.nonzero(idx == e).flatten()
        if toks.numel() > capacity:
            keys = uniform[toks, e]
            keep = torch.topk(keys, k=capacity)[1]
            toks = toks[torch.sort(keep)[0]]
        slot[toks] = torch.arange(toks.numel())
        token_of_slot[e, : toks.numel()] = toks
    return idx, slot, token_of_slot


def top2gating(logits: torch.Tensor, noise: Optional[torch.Tensor], capacity: int):
    """deepspeed 0.16.5 deepspeed.moe.sharded_moe.top2gating(drop_tokens=True, top2_2nd_expert_sampling=True) - what TopKGate calls for k = 2,
    i.e. control_params.top_num = 2 (src/UniGenTransformer.py:162,197 / :808,857 / :1565,1650; src/UniGenUtils.py:33-36 passes k and the
    defaults capacity_factor = 1, min_capacity = 4, drop_tokens, top2_2nd_expert_sampling). deepspeed is not in /root/reference: restated
    from its published source, statement by statement; the Gumbel(0, 1) sample it draws from the device RNG (gumbel_rsample) is passed in
    (None = top2_2nd_expert_sampling off). `capacity` = _capacity(gates, capacity_factor * 2, min_capacity) = moe_capacity(S, E, 2.0).
    Returns l_aux, combine_weights [S,E,C] fp32, dispatch_mask [S,E,C] bool, exp_counts [E] int64. PARITY UNPINNED."""
    S, E = logits.shape
    gates = F.softmax(logits, dim=1)
    indices1_s = torch.argmax(gates, dim=1)
    mask1 = F.one_hot(indices1_s, num_classes=E)
    if noise is not None:
        logits = logits + noise                                        # logits += gumbel_rsample(...)
    logits_except1 = logits.masked_fill(mask1.bool(), float("-inf"))
    indices2_s = torch.argmax(logits_except1, dim=1)
    mask2 = F.one_hot(indices2_s, num_classes=E)
    locations1 = torch.cumsum(mask1, dim=0) - 1
    locations2 = torch.cumsum(mask2, dim=0) - 1
    locations2 = locations2 + torch.sum(mask1, dim=0, keepdim=True)    # second choices queue behind every first choice of the expert
    me = torch.mean(gates, dim=0)
    ce = torch.mean(mask1.float(), dim=0)
    l_aux = torch.mean(me * ce) * E * E
    exp_counts = torch.sum(mask1 + mask2, dim=0).detach()
    mask1 = mask1 * torch.lt(locations1, capacity)
    mask2 = mask2 * torch.lt(locations2, capacity)
    locations1_s = torch.sum(locations1 * mask1, dim=1)
    locations2_s = torch.sum(locations2 * mask2, dim=1)
    mask1_float, mask2_float = mask1.float(), mask2.float()
    gates1_s = torch.einsum("se,se->s", gates, mask1_float)
    gates2_s = torch.einsum("se,se->s", gates, mask2_float)
    denom_s = torch.clamp(gates1_s + gates2_s, min=torch.finfo(gates.dtype).eps)
    gates1_s = gates1_s / denom_s
    gates2_s = gates2_s / denom_s
    gates1 = torch.einsum("s,se->se", gates1_s, mask1_float)
    gates2 = torch.einsum("s,se->se", gates2_s, mask2_float)
    locations1_sc = F.one_hot(locations1_s, num_classes=capacity).float()
    locations2_sc = F.one_hot(locations2_s, num_classes=capacity).float()
    combine_weights = torch.einsum("se,sc->sec", gates1, locations1_sc) + torch.einsum("se,sc->sec", gates2, locations2_sc)
    dispatch_mask = combine_weights.bool()
    return l_aux, combine_weights, dispatch_mask, exp_counts


def routing_top2(gates: torch.Tensor, logits: torch.Tensor, noise: Optional[torch.Tensor], capacity: int, idx: Optional[torch.Tensor] = None):
    """Index form of top2gating: idx [2, S] (first / second choice), slot [2, S] (-1 = dropped), token_of_slot [E, capacity], weights [2, S] fp32
    (the normalised gate probabilities, 0 for a dropped choice). Used to check the dense form and the HIP kernels. `idx`: take the two choices
    as given (a device's, which may differ from the host's on floating-point near-ties) and restate only what follows from them."""
    S, E = gates.shape
    if idx is None:
        i1 = torch.argmax(gates, dim=1)
        noisy = logits if noise is None else logits + noise
        i2 = torch.argmax(noisy.masked_fill(F.one_hot(i1, num_classes=E).bool(), float("-inf")), dim=1)
        idx = torch.stack([i1, i2])
    i1, i2 = idx[0], idx[1]
    slot = torch.full((2, S), -1, dtype=torch.int64)
    token_of_slot = torch.full((E, capacity), -1, dtype=torch.int64)
    for e in range(E):
        fill = 0
        for k in range(2):
            toks = torch.nonzero(idx[k] == e).flatten()
            loc = fill + torch.arange(toks.numel())
            keep = loc < capacity
            slot[k, toks[keep]] = loc[keep]
            token_of_slot[e, loc[keep]] = toks[keep]
            fill += toks.numel()
    ar = torch.arange(S)
    g = torch.stack([gates[ar, i1], gates[ar, i2]]) * (slot >= 0).float()
    weights = g / torch.clamp(g.sum(0, keepdim=True), min=torch.finfo(gates.dtype).eps)
    return idx, slot, token_of_slot, weights


def topkgating(logits: torch.Tensor, k: int, capacity: int):
    """deepspeed 0.16.5 deepspeed.moe.sharded_moe.topkgating(drop_tokens=True, drop_policy="probs") - what TopKGate calls for k > 2, i.e.
    control_params.top_num > 2 (src/UniGenTransformer.py:808 -> src/UniGenUtils.py:33-36). deepspeed is not in /root/reference: restated from its
    published source, statement by statement. `capacity` = _capacity(gates, capacity_factor * k, min_capacity) = moe_capacity(S, E, k). No random
    draw. Two properties of the published code that the restatement keeps: the capacity rule ranks the top-k LOGITS (not probabilities) of an
    expert's column against ZEROS for the tokens that did not choose it, so a chosen logit below zero loses to every non-chooser and is dropped
    whenever choosers-above-zero + non-choosers fill the capacity; and the kept probabilities are renormalised by their sum.
    Returns l_aux, combine_weights [S,E,C] fp32, dispatch_mask [S,E,C] bool, exp_counts [E] int64. PARITY UNPINNED."""
    S, E = logits.shape
    top_gate, top_idx = torch.topk(logits, k=k, dim=1)
    gates = F.softmax(logits, dim=1)
    topk_masked_gates = torch.zeros_like(logits).scatter(1, top_idx, top_gate)
    mask = torch.zeros_like(gates, dtype=torch.bool).scatter_(1, top_idx, 1)
    exp_counts = torch.sum(mask, dim=0).detach()
    me = torch.mean(gates, dim=0)
    ce = torch.mean(mask.float(), dim=0)
    l_aux = torch.mean(me * ce) * E * E / k
    capacity_indices = torch.topk(topk_masked_gates, k=capacity, dim=0, sorted=False)[1]
    capacity_mask = torch.zeros_like(logits).scatter(0, capacity_indices, 1)
    mask = torch.logical_and(mask, capacity_mask)
    locations = torch.cumsum(mask, dim=0) - 1
    gates_masked = gates * mask
    gates_s = torch.sum(gates_masked, dim=-1, keepdim=True)
    denom_s = torch.clamp(gates_s, min=torch.finfo(gates_masked.dtype).eps)
    gates_masked = gates_masked / denom_s
    locations_sc = F.one_hot(locations * mask, num_classes=capacity).float()          # _one_hot_to_float
    combine_weights = torch.einsum("se,sec->sec", gates_masked, locations_sc)
    dispatch_mask = combine_weights.bool()
    return l_aux, combine_weights, dispatch_mask, exp_counts


def routing_topk(gates: torch.Tensor, logits: torch.Tensor, k: int, capacity: int, idx: Optional[torch.Tensor] = None):
    """Index form of topkgating: idx [k, S] (choices by descending logit), slot [k, S] (-1 = dropped), token_of_slot [E, capacity], weights [k, S]
    fp32. An expert keeps the `capacity` largest entries of its column (chosen logit, or 0 for a non-chooser), ties in token order (torch.topk's
    own choice among exact ties is unspecified; between a chosen logit and the zeros of non-choosers a tie needs a logit of exactly 0);
    slots in token order among the kept. `idx`: take the choices as given (a device's) and restate only what follows from them."""
    S, E = gates.shape
    if idx is None:
        idx = torch.topk(logits, k=k, dim=1)[1].t().contiguous()
    slot = torch.full((k, S), -1, dtype=torch.int64)
    token_of_slot = torch.full((E, capacity), -1, dtype=torch.int64)
    ar = torch.arange(S)
    for e in range(E):
        chosen = (idx == e).any(0)
        col = torch.where(chosen, logits[:, e], torch.zeros(()))
        order = torch.sort(col, descending=True, stable=True)[1]                       # value descending, token ascending among equals
        in_cap = torch.zeros(S, dtype=torch.bool)
        in_cap[order[:capacity]] = True
        kept = chosen & in_cap
        toks = ar[kept]
        token_of_slot[e, : toks.numel()] = toks
        loc = torch.cumsum(kept.long(), 0) - 1
        for kk in range(k):
            sel = (idx[kk] == e) & kept
            slot[kk, sel] = loc[sel]
    g = torch.stack([gates[ar, idx[kk]] for kk in range(k)]) * (slot >= 0).float()
    weights = g / torch.clamp(g.sum(0, keepdim=True), min=torch.finfo(gates.dtype).eps)
    return idx, slot, token_of_slot, weights


def gate_route(logits: torch.Tensor, draw: Optional[torch.Tensor], top_num: int = 1):
    """TopKGate.forward's dispatch on k (deepspeed 0.16.5: k = 1 -> top1gating, k = 2 -> top2gating, k > 2 -> topkgating) with the capacity each computes for the
    reference's settings. `draw`: the random sample the gating function takes from the device RNG - the Uniform(0,1) of Random Token
    Selection for k = 1, the Gumbel(0,1) added to the logits for k = 2. Returns (l_aux, combine_weights, dispatch_mask, exp_counts, routing)."""
    S, E = logits.shape
    gates = F.softmax(logits, dim=1)
    if top_num == 1:
        C = moe_capacity(S, E)
        l_aux, cw, dm, exp_counts = top1gating(logits, draw, C)
        idx, slot, tos = routing_from_gates(gates, draw, C)
        routing = dict(gates=gates, idx=idx, slot=slot, token_of_slot=tos, capacity=C, logits=logits)
    elif top_num == 2:
        C = moe_capacity(S, E, capacity_factor=2.0)
        l_aux, cw, dm, exp_counts = top2gating(logits, draw, C)
        idx, slot, tos, weights = routing_top2(gates, logits, draw, C)
        routing = dict(gates=gates, idx=idx, slot=slot, token_of_slot=tos, capacity=C, logits=logits, weights=weights)
    else:                                                                               # k > 2: topkgating, no random draw
        if not 2 < top_num <= E:
            raise ValueError(f"top_num {top_num} with {E} experts")
        C = moe_capacity(S, E, capacity_factor=float(top_num))
        l_aux, cw, dm, exp_counts = topkgating(logits, top_num, C)
        idx, slot, tos, weights = routing_topk(gates, logits, top_num, C)
        routing = dict(gates=gates, idx=idx, slot=slot, token_of_slot=tos, capacity=C, logits=logits, weights=weights)
    return l_aux, cw, dm, exp_counts, routing


def modulated_flatten_literal(x: torch.Tensor, w: torch.Tensor, s: torch.Tensor) -> torch.Tensor:
    """src/UniGenUtils.py:204-228, 3-D `s` branch, verbatim semantics (materialises b x n x o x i): tiny sizes only."""
    w = w.unsqueeze(0).unsqueeze(1) * s.unsqueeze(2)
    return torch.einsum("bnoi,bni->bno", w, x)


def modulated_linear(x: torch.Tensor, w: torch.Tensor, s: torch.Tensor) -> torch.Tensor:
    """Algebraically identical restatement used at scale: Linear_W(s * x)."""
    return F.linear(s * x, w)


def expert_forward(state: State, xd: torch.Tensor, cd: torch.Tensor, pd: torch.Tensor, cpd: torch.Tensor, literal: bool = False,
                   prefix: str = "moe.moe_layer.experts.deepspeed_experts."):
    """UniGenFlux.expert_forward (src/UniGenTransformer.py:925-967; UniGenBase.expert_forward :225-267 is the same text), modulated experts:
    per expert e on its dispatched slots  c' = mf(c, W_c, Lin_c(cond_pooled)) + b_c ;  h' = mf(h + c', W_h, Lin_h(pooled)) + b_h.
    xd, cd [E, C, D]; pd, cpd [E, C, P] -> (h' [E, C, D], c' [E, C, D]). REFERENCE-PINNED: tests/test_ref_leaf_cpu.py holds it to the
    outputs of the reference's own function (tests/golden/ref_leaf.safetensors)."""
    dt = xd.dtype
    mf = modulated_flatten_literal if literal else modulated_linear
    yh, yc = [], []
    for e in range(xd.shape[0]):
        p = f"{prefix}{e}"
        wc, bc = state[p + ".0.0.weight"].to(dt), state[p + ".0.0.bias"].to(dt)
        wh, bh = state[p + ".1.0.weight"].to(dt), state[p + ".1.0.bias"].to(dt)
        s_c = linear(state, p + ".0.1", cpd[e][None])
        s_h = linear(state, p + ".1.1", pd[e][None])
        c_e = mf(cd[e][None], wc, s_c) + bc.unsqueeze(0)
        h_e = mf(xd[e][None] + c_e, wh, s_h) + bh.unsqueeze(0)
        yh.append(h_e[0]); yc.append(c_e[0])
    return torch.stack(yh), torch.stack(yc)


def comoe_experts(state: State, cfg: FluxConfig, x, c, pooled, cond_pooled, gates_in: Optional[torch.Tensor],
                  uniform: torch.Tensor, literal: bool = False, choice: Optional[torch.Tensor] = None):
    """MOELayer.forward + UniGenFlux.expert_forward with modulated experts. x, c: [B, N, D]. Returns
    (expert_h [B,N,D], expert_c [B,N,D], l_aux, exp_counts, routing dict)."""
    B, N, D = x.shape
    E = cfg.expert_nums
    S = B * N
    dt = x.dtype
    choice = (x + c if choice is None else choice).reshape(S, D)        # choice_expert_input (src/UniGenTransformer.py:979)
    wg = state["moe.moe_layer.gate.wg.weight"]
    logits = F.linear(choice.float(), wg.float())           # TopKGate.forward: fp32 input and weight
    l_aux, combine_weights, dispatch_mask, exp_counts, routing = gate_route(logits, uniform, cfg.top_num)
    C, token_of_slot = routing["capacity"], routing["token_of_slot"]

    def dispatch(t2d):  # einsum("sec,sm->ecm") with a one-hot mask == row gather, zeros for empty slots
        out = torch.zeros(E, C, t2d.shape[-1], dtype=t2d.dtype)
        valid = token_of_slot >= 0
        out[valid] = t2d[token_of_slot[valid]]
        return out

    if literal:
        dm = dispatch_mask.to(dt)
        xd = torch.einsum("sec,sm->ecm", dm, x.reshape(S, D))
        cd = torch.einsum("sec,sm->ecm", dm, c.reshape(S, D))
        pd = torch.einsum("sec,sm->ecm", dm, pooled.unsqueeze(1).expand(-1, N, -1).reshape(S, -1))
        cpd = torch.einsum("sec,sm->ecm", dm, cond_pooled.unsqueeze(1).expand(-1, N, -1).reshape(S, -1))
    else:
        xd, cd = dispatch(x.reshape(S, D)), dispatch(c.reshape(S, D))
        pd = dispatch(pooled.unsqueeze(1).expand(-1, N, -1).reshape(S, -1))
        cpd = dispatch(cond_pooled.unsqueeze(1).expand(-1, N, -1).reshape(S, -1))
    yh, yc = expert_forward(state, xd, cd, pd, cpd, literal=literal)    # [E, C, D]
    cw = combine_weights.to(dt)
    eh = torch.einsum("sec,ecm->sm", cw, yh).reshape(B, N, D)
    ec = torch.einsum("sec,ecm->sm", cw, yc).reshape(B, N, D)
    return eh, ec, l_aux, exp_counts, routing


# ---------------------------------------------------------------------------------------------------------------------
# Wiring of UniGenFlux / MultiCondtionUniGenFlux: the reference's own methods (moe_forward, preprocess_moe_forward, control_forward,
# base_forward), written against INJECTED module callables `m` - the attributes of `self` those methods call. REFERENCE-PINNED:
# tests/test_ref_wiring_cpu.py runs these very functions on the deterministic stand-in modules of tests/wiring_cases.py and holds them, bit for
# bit, to the outputs of the reference's methods executed from /root/reference on the same stand-ins (tests/golden/ref_wiring.safetensors,
# written by tests/golden/make_ref_wiring_golden.py). unigen_flux_forward below calls the same functions with modules built from a state dict.
#
#   m.double(i, x, enc, temb) -> (enc, x)                                transformer_blocks[i] (base RoPE table inside)
#   m.single(j, h, temb) -> h                                            single_transformer_blocks[j]
#   m.control_joint(k, z, enc, temb, hd_ids, encoder_hd_ids) -> (enc, z) control_joint_trans_blocks[k]; ids None = no RoPE ({} kwargs)
#   m.control_single(k, h, temb, hd_ids) -> h                            control_single_trans_blocks[k]
#   m.add_joint(k, z), m.add_single(k, z)                                controlnet_add_*_blocks[k] (zero-res projections)
#   m.control_x_embedder(c), m.control_context_embedder(enc)
#   m.control_time_text_embed(timestep, pooled, guidance), m.control_condition_embed(timestep, cond_pooled, guidance)
#   m.moe_layer(choice_expert_input=, hidden_states=, condition_hidden_states=, encoder_hidden_states=, temb=, condition_temb=,
#               condition_pooled_projections=, pooled_projections=) -> (expert_h, expert_c, l_aux, exp_counts)      self.moe.moe_layer
#   m.shared_expert(k, x, enc, temb, hd_ids, encoder_hd_ids) -> (enc, x);  m.consis_module(k, ...) likewise
# ---------------------------------------------------------------------------------------------------------------------

def _cat_ids(*ids):
    return torch.cat(list(ids), dim=0)


def flux_moe_forward(m, cfg: FluxConfig, x, c, *, ctrl_enc, control_temb, condition_temb, pooled, cond_pooled, ids: Optional[Dict[str, torch.Tensor]]):
    """UniGenFlux.moe_forward (src/UniGenTransformer.py:969-1026; UniGenBase.moe_forward :269-296 is the same text without the consistency
    module). x = hidden states, c = embedded condition tokens, kwargs['temb'] = control_temb (:1055). `ids` None <=> use_rope False.
    Returns ((hidden + expert_h, condition + expert_c), l_aux, exp_counts)."""
    rope = ids is not None
    eh, ec, l_aux, exp_counts = m.moe_layer(choice_expert_input=x + c, hidden_states=x, condition_hidden_states=c, encoder_hidden_states=ctrl_enc,
                                            temb=control_temb, condition_temb=condition_temb, condition_pooled_projections=cond_pooled,
                                            pooled_projections=pooled)
    out = (eh, ec)
    if getattr(cfg, "use_consis_module", False):
        # "V2" (:984-1004): consis_module[0] is called TWICE, consis_module[1] is built (:909-920) but never called; both keep the sample stream
        #   1. sample = expert condition states, context = condition tokens c, temb = condition_temb, ids (condition_ids | condition_ids)
        #   2. sample = [expert hidden | result of 1], context = hidden states x, temb = kwargs['temb'], ids ([img | cond] | img)
        _, cech = m.consis_module(0, ec, c, condition_temb, ids["condition_ids"] if rope else None, ids["condition_ids"] if rope else None)
        _, y = m.consis_module(0, torch.cat([eh, cech], dim=1), x, control_temb,
                               _cat_ids(ids["img_ids"], ids["condition_ids"]) if rope else None, ids["img_ids"] if rope else None)
        n0 = eh.shape[1]
        eh, ec = eh + y[:, :n0, :], ec + y[:, n0:, :]
        # reference quirk (found by the wiring fixture, case flux_consis_rope_no_single): the consistency terms are added to the local
        # variables only; the returned tuple `expert_output` is rebuilt from them inside `if self.use_shared_expert` (:1024) and nowhere else, so
        # with use_consis_module and WITHOUT use_shared_expert the reference returns the raw MoE output and the module's work is discarded.
    if cfg.use_shared_expert:
        # shared_expert[0](hidden = x, encoder = c, temb = condition_temb), sample rows first: [img | cond]            (:1013-1015)
        cond_s, x_s = m.shared_expert(0, x, c, condition_temb, ids["img_ids"] if rope else None, ids["condition_ids"] if rope else None)
        # shared_expert[1](hidden = [x_s | cond_s], encoder = ctrl_enc, temb = kwargs['temb'])                        (:1017-1022)
        _, hc = m.shared_expert(1, torch.cat([x_s, cond_s], dim=1), ctrl_enc, control_temb,
                                _cat_ids(ids["img_ids"], ids["condition_ids"]) if rope else None, ids["prompt_ids"] if rope else None)
        n = x_s.shape[1]
        x_s, cond_s = hc[:, :n, :], hc[:, n:, :]
        out = (x_s + eh, cond_s + ec)
    return out, l_aux, exp_counts


def flux_preprocess_moe_forward(m, cfg: FluxConfig, x, condition_hidden_states, enc, pooled, condition_pooled_projections, timestep, *,
                                guidance, img_ids, prompt_ids, condition_ids, trace: Optional[dict] = None):
    """UniGenFlux.preprocess_moe_forward (src/UniGenTransformer.py:1028-1068) and, when the condition arguments are lists,
    MultiCondtionUniGenFlux.preprocess_moe_forward (:1275-1322: the whole MoE per condition; sum of (expert_h + expert_c) and of the
    condition tembs; l_aux / exp_counts of the LAST condition). Returns dict(z0, ctrl_enc, control_temb, condition_temb, l_aux, exp_counts)."""
    multi = isinstance(condition_hidden_states, (list, tuple))
    control_pooled = pooled if cfg.use_pooled_prompt_embeds else torch.zeros_like(pooled)
    control_temb = m.control_time_text_embed(timestep, control_pooled, guidance)
    if multi:
        ctrl_enc = m.control_context_embedder(enc)
        merged, merged_temb = [], []
        for k, (cid, ct, cp) in enumerate(zip(condition_ids, condition_hidden_states, condition_pooled_projections)):
            if cp.dim() == 1:
                cp = cp.unsqueeze(0)
            if ct.dim() == 2:
                ct = ct.unsqueeze(0)
            c = m.control_x_embedder(ct)
            condition_temb = m.control_condition_embed(timestep, cp, guidance)
            ids = dict(img_ids=img_ids, prompt_ids=prompt_ids, condition_ids=cid) if cfg.use_rope else None
            if hasattr(m, "select_condition"):
                m.select_condition(k)
            (oh, oc), l_aux, exp_counts = flux_moe_forward(m, cfg, x, c, ctrl_enc=ctrl_enc, control_temb=control_temb, condition_temb=condition_temb,
                                                           pooled=pooled, cond_pooled=cp, ids=ids)
            merged.append(oh + oc)
            merged_temb.append(condition_temb)
        z0, condition_temb = sum(merged), sum(merged_temb)
    else:
        c = m.control_x_embedder(condition_hidden_states)
        condition_temb = m.control_condition_embed(timestep, condition_pooled_projections, guidance)
        ctrl_enc = m.control_context_embedder(enc)
        ids = dict(img_ids=img_ids, prompt_ids=prompt_ids, condition_ids=condition_ids) if cfg.use_rope else None
        (oh, oc), l_aux, exp_counts = flux_moe_forward(m, cfg, x, c, ctrl_enc=ctrl_enc, control_temb=control_temb, condition_temb=condition_temb,
                                                       pooled=pooled, cond_pooled=condition_pooled_projections, ids=ids)
        z0 = oh + oc                                                                     # control_forward :1089
    if trace is not None:
        trace["z0"] = z0
    return dict(z0=z0, ctrl_enc=ctrl_enc, control_temb=control_temb, condition_temb=condition_temb, l_aux=l_aux, exp_counts=exp_counts)


def flux_base_forward(m, cfg: FluxConfig, x, condition_hidden_states, enc, pooled, condition_pooled_projections, timestep, *,
                      conditioning_scale, temb, guidance, img_ids, prompt_ids, condition_ids, trace: Optional[dict] = None):
    """UniGenFlux.base_forward (src/UniGenTransformer.py:1106-1180) with control_forward (:1070-1104; MultiCondtionUniGenFlux's :1324-1357 is the
    same with the merged keys) inlined: loop A (base double block -> control block on the BASE stream -> zero-res add), concat [text | image],
    loop B (base single block -> control single block -> zero-res add), slice. The control blocks are not chained: each reads the base stream
    (only the first reads the CoMoE output); ctrl_enc / condition_temb are fixed after base block 0. Returns (x, enc, moe_out)."""
    moe_out = None
    n_d, n_cj = cfg.num_layers, cfg.cn_joint_layers
    for i in range(n_d):
        k = int(i / (n_d / n_cj))                                                        # :1126-1127
        enc, x = m.double(i, x, enc, temb)
        if moe_out is None:                                                              # :1085-1089, once, with the text stream AFTER base block 0
            moe_out = flux_preprocess_moe_forward(m, cfg, x, condition_hidden_states, enc, pooled, condition_pooled_projections, timestep,
                                                  guidance=guidance, img_ids=img_ids, prompt_ids=prompt_ids, condition_ids=condition_ids, trace=trace)
            z_in = moe_out["z0"]
        else:
            z_in = x
        _, z = m.control_joint(k, z_in, moe_out["ctrl_enc"], moe_out["condition_temb"],
                               img_ids if cfg.use_rope else None, prompt_ids if cfg.use_rope else None)          # :1095-1097
        x = x + m.add_joint(k, z) * conditioning_scale                                   # :1104, :1141
        if trace is not None:
            trace.setdefault("x_after_double", []).append(x)
    T = enc.shape[1]
    h = torch.cat([enc, x], dim=1)
    n_s, n_cs = cfg.num_single_layers, cfg.cn_single_layers
    for j in range(n_s):
        h = m.single(j, h, temb)
        if cfg.use_single_trans_blocks:                                                  # hasattr(self, 'control_single_trans_blocks') :1159
            k = int(j / (n_s / n_cs))
            z = m.control_single(k, h, moe_out["condition_temb"], _cat_ids(prompt_ids, img_ids) if cfg.use_rope else None)   # :1099-1102
            y = m.add_single(k, z) * conditioning_scale
            if cfg.single_block_control_method == "overall_add":
                h = h + y
            else:                                                                        # "in flux controlnet": image rows only (:1168-1172)
                h = torch.cat([h[:, :T, ...], h[:, T:, ...] + y[:, T:, ...]], dim=1)
    return h[:, T:, ...], enc, moe_out


def flux_modules(state: State, cfg: FluxConfig, rope, gate_uniform, n_cond: int = 1, trace: Optional[dict] = None):
    """The injected callables of the wiring above, built from a flat state dict (the reference's module tree by parameter name)."""
    from types import SimpleNamespace
    m = SimpleNamespace()
    unis = gate_uniform if isinstance(gate_uniform, (list, tuple)) else [gate_uniform] * n_cond
    sel = dict(k=0)

    def crope(hd_ids, encoder_hd_ids=None):
        # JointAttnRopeProcessor (src/UniGenUtils.py:590-597): table over cat([hd_ids, encoder_hd_ids]), cast to the ids' dtype
        if hd_ids is None:
            return None
        return control_rope(cfg, hd_ids if encoder_hd_ids is None else torch.cat([hd_ids, encoder_hd_ids], dim=0))

    def joint(prefix):
        return lambda k, x, enc, temb, hd_ids, encoder_hd_ids: flux_double_block(state, f"{prefix}.{k}", cfg, x, enc, temb, crope(hd_ids, encoder_hd_ids), text_first=False)

    def moe_layer(*, choice_expert_input, hidden_states, condition_hidden_states, encoder_hidden_states, temb, condition_temb,
                  condition_pooled_projections, pooled_projections):
        eh, ec, l_aux, exp_counts, routing = comoe_experts(state, cfg, hidden_states, condition_hidden_states, pooled_projections,
                                                           condition_pooled_projections, None, unis[sel["k"]], choice=choice_expert_input)
        if trace is not None:
            trace.setdefault("routing", []).append(routing)
        return eh, ec, l_aux, exp_counts

    m.select_condition = lambda k: sel.update(k=k)
    m.double = lambda i, x, enc, temb: flux_double_block(state, f"transformer_blocks.{i}", cfg, x, enc, temb, rope, text_first=True)
    m.single = lambda j, h, temb: flux_single_block(state, f"single_transformer_blocks.{j}", cfg, h, temb, rope)
    m.control_joint = joint("control_joint_trans_blocks")
    m.control_single = lambda k, h, temb, hd_ids: flux_single_block(state, f"control_single_trans_blocks.{k}", cfg, h, temb, crope(hd_ids))
    m.add_joint = lambda k, z: linear(state, f"controlnet_add_joint_blocks.{k}", z)
    m.add_single = lambda k, z: linear(state, f"controlnet_add_single_blocks.{k}", z)
    m.control_x_embedder = lambda c: linear(state, "control_x_embedder", c)
    m.control_context_embedder = lambda e: linear(state, "control_context_embedder", e)
    m.control_time_text_embed = lambda t, p, g: time_text_embed(state, "control_time_text_embed", t, p, g)
    m.control_condition_embed = lambda t, p, g: time_text_embed(state, "control_condition_embed", t, p, g)
    m.moe_layer = moe_layer
    m.shared_expert = joint("shared_expert")
    m.consis_module = joint("consis_module")
    return m


def comoe(state: State, cfg: FluxConfig, x, cond_tokens, ctrl_enc, control_temb, condition_temb, pooled, cond_pooled,
          ids: Dict[str, torch.Tensor], uniform: torch.Tensor):
    """One condition's CoMoE: control_x_embedder + MoE experts + shared experts (src/UniGenTransformer.py:969-1026,1040) through the pinned
    wiring. Returns (expert_hidden_states, expert_condition_hidden_states, l_aux, exp_counts, routing)."""
    trace: dict = {}
    m = flux_modules(state, cfg, None, uniform, trace=trace)
    c = m.control_x_embedder(cond_tokens)
    (oh, oc), l_aux, exp_counts = flux_moe_forward(m, cfg, x, c, ctrl_enc=ctrl_enc, control_temb=control_temb, condition_temb=condition_temb,
                                                   pooled=pooled, cond_pooled=cond_pooled, ids=ids if cfg.use_rope else None)
    return oh, oc, l_aux, exp_counts, trace["routing"][0]


# ---------------------------------------------------------------------------------------------------------------------
# UniGenFlux.forward / MultiCondtionUniGenFlux.forward
# ---------------------------------------------------------------------------------------------------------------------

def unigen_flux_forward(state: State, cfg: FluxConfig, *, hidden_states, condition_hidden_states, encoder_hidden_states,
                        pooled_projections, condition_pooled_projections, timestep, img_ids, txt_ids, condition_ids,
                        guidance=None, conditioning_scale: float = 1.0, gate_uniform=None, dtype=torch.bfloat16,
                        io_dtype=torch.bfloat16, trace: Optional[dict] = None):
    """src/UniGenTransformer.py:1182-1271 (single condition) and :1360-1450 (lists => MultiCondtionUniGenFlux): embeds, the base RoPE table,
    base_forward through the reference-pinned wiring above, norm_out + proj_out.
    gate_uniform: [S, E] fp32 uniforms (one tensor, or a list per condition) standing in for the RTS draw."""
    multi = isinstance(condition_hidden_states, (list, tuple))
    dt = dtype
    cast = lambda t: t.to(dt)
    x = linear(state, "x_embedder", cast(hidden_states))
    # `timestep.to(hidden_states.dtype) * 1000` (:1217-1220): evaluated in the model's I/O dtype (bf16) in both modes, so the
    # fp32 mode sees the same effective timestep (e.g. 752, not 750) as the reference's bf16 run.
    if guidance is not None:
        guidance = (guidance.to(io_dtype) * 1000).to(dt)
    timestep = (timestep.to(io_dtype) * 1000).to(dt)
    pooled = cast(pooled_projections)
    temb = time_text_embed(state, "time_text_embed", timestep, pooled, guidance)
    enc = linear(state, "context_embedder", cast(encoder_hidden_states))
    rope = flux_pos_embed(torch.cat((txt_ids, img_ids), dim=0), cfg.axes_dims_rope)   # base path: fp32 tables
    if multi:
        cond, cpool = [cast(t) for t in condition_hidden_states], [cast(t) for t in condition_pooled_projections]
    else:
        cond, cpool = cast(condition_hidden_states), cast(condition_pooled_projections)
    m = flux_modules(state, cfg, rope, gate_uniform, n_cond=len(cond) if multi else 1, trace=trace)
    x, enc, moe_out = flux_base_forward(m, cfg, x, cond, enc, pooled, cpool, timestep, conditioning_scale=conditioning_scale, temb=temb,
                                        guidance=guidance, img_ids=img_ids, prompt_ids=txt_ids, condition_ids=condition_ids, trace=trace)
    x = adaln_continuous(state, "norm_out", x, temb)
    out = linear(state, "proj_out", x)
    return out, dict(moe_loss=moe_out["l_aux"] * 0.1), dict(expert_counts=moe_out["exp_counts"])


# ---------------------------------------------------------------------------------------------------------------------
# denoise loop (src/UniGenPipeline.py:662-677, 721-789) and FlowMatchEulerDiscreteScheduler
# ---------------------------------------------------------------------------------------------------------------------

def schnell_sigmas(num_steps: int, shift: float = 1.0) -> torch.Tensor:
    sig = torch.linspace(1.0, 1.0 / num_steps, num_steps, dtype=torch.float32)
    sig = shift * sig / (1 + (shift - 1) * sig)
    return torch.cat([sig, torch.zeros(1)])


def euler_step(latents: torch.Tensor, model_output: torch.Tensor, sigma: float, sigma_next: float) -> torch.Tensor:
    """FlowMatchEulerDiscreteScheduler.step (diffusers 0.32.2; called at src/UniGenPipeline.py:768 / :411), executed as torch executes it:
    the scheduler's sigmas are an fp32 tensor on the model's device, `sigma_next - sigma` a 0-dim fp32 tensor, and `step * model_output` therefore
    takes model_output's dtype (type promotion ignores 0-dim operands of the same category) with BOTH operands cast to it: in bf16 the step and the
    product are rounded before the product meets the upcast sample. Written as the scheduler writes it; torch does the rest."""
    sig = torch.tensor([sigma, sigma_next], dtype=torch.float32)
    prev = latents.to(torch.float32) + (sig[1] - sig[0]) * model_output
    return prev.to(model_output.dtype)


def denoise(state: State, cfg: FluxConfig, *, latents, num_steps: int, dtype=torch.bfloat16, gate_uniforms=None, true_cfg_scale: float = 1.0,
            negative_encoder_hidden_states=None, negative_pooled_projections=None, negative_gate_uniforms=None, negative_txt_ids=None, **fwd_kwargs):
    """UniGenFLUXPipeline.__call__ loop (src/UniGenPipeline.py:721-789): timestep = t.expand(B).to(latents.dtype); forward(timestep / 1000);
    with true CFG (:748-763) a second forward on the negative prompt - without conditioning_scale, i.e. the forward's default - and
    neg + true_cfg_scale * (pred - neg) in the working dtype; Euler step."""
    sig = schnell_sigmas(num_steps)
    latents = latents.to(dtype)
    B = latents.shape[0]
    do_true_cfg = true_cfg_scale > 1 and negative_encoder_hidden_states is not None and negative_pooled_projections is not None
    for i in range(num_steps):
        t = (sig[i] * 1000).expand(B).to(latents.dtype)
        uni = None if gate_uniforms is None else gate_uniforms[i]
        pred = unigen_flux_forward(state, cfg, hidden_states=latents, timestep=t / 1000, gate_uniform=uni, dtype=dtype, **fwd_kwargs)[0]
        if do_true_cfg:
            kw = {k: v for k, v in fwd_kwargs.items() if k not in ("encoder_hidden_states", "pooled_projections", "conditioning_scale")}
            nuni = None if negative_gate_uniforms is None else negative_gate_uniforms[i]
            if negative_txt_ids is not None:
                kw["txt_ids"] = negative_txt_ids
            neg = unigen_flux_forward(state, cfg, hidden_states=latents, timestep=t / 1000, gate_uniform=nuni, dtype=dtype,
                                      encoder_hidden_states=negative_encoder_hidden_states, pooled_projections=negative_pooled_projections, **kw)[0]
            pred = neg + true_cfg_scale * (pred - neg)
        latents = euler_step(latents, pred, float(sig[i]), float(sig[i + 1]))
    return latents


# ---------------------------------------------------------------------------------------------------------------------
# LoRA (peft 0.15 Linear; src/lora_switching_module.py:4-38 — never invoked by the reference)
# ---------------------------------------------------------------------------------------------------------------------

def lora_linear(x, w, b, adapters: List[Tuple[torch.Tensor, torch.Tensor, float]]):
    """y = base(x) + sum_a B_a(A_a(x)) * scaling_a, scaling = lora_alpha / r (adapters: (A [r,K], B [N,r], scaling))."""
    y = F.linear(x, w, b)
    for A, Bm, sc in adapters:
        y = y + F.linear(F.linear(x, A), Bm) * sc
    return y


# ---------------------------------------------------------------------------------------------------------------------
# synthetic weights / inputs: N(0, 0.02^2) weights, zero biases, unit RMSNorm weights, zero-res
# projections re-initialised so the control path contributes.
# ---------------------------------------------------------------------------------------------------------------------

def _double_block_shapes(p: str, D: int, dh: int) -> Dict[str, Tuple[int, ...]]:
    s = {}
    for n in ("norm1", "norm1_context"):
        s[f"{p}.{n}.linear.weight"] = (6 * D, D); s[f"{p}.{n}.linear.bias"] = (6 * D,)
    for n in ("to_q", "to_k", "to_v", "add_q_proj", "add_k_proj", "add_v_proj", "to_out.0", "to_add_out"):
        s[f"{p}.attn.{n}.weight"] = (D, D); s[f"{p}.attn.{n}.bias"] = (D,)
    for n in ("norm_q", "norm_k", "norm_added_q", "norm_added_k"):
        s[f"{p}.attn.{n}.weight"] = (dh,)
    for n in ("ff", "ff_context"):
        s[f"{p}.{n}.net.0.proj.weight"] = (4 * D, D); s[f"{p}.{n}.net.0.proj.bias"] = (4 * D,)
        s[f"{p}.{n}.net.2.weight"] = (D, 4 * D); s[f"{p}.{n}.net.2.bias"] = (D,)
    return s


def _single_block_shapes(p: str, D: int, dh: int) -> Dict[str, Tuple[int, ...]]:
    s = {f"{p}.norm.linear.weight": (3 * D, D), f"{p}.norm.linear.bias": (3 * D,),
         f"{p}.proj_mlp.weight": (4 * D, D), f"{p}.proj_mlp.bias": (4 * D,),
         f"{p}.proj_out.weight": (D, 5 * D), f"{p}.proj_out.bias": (D,)}
    for n in ("to_q", "to_k", "to_v"):
        s[f"{p}.attn.{n}.weight"] = (D, D); s[f"{p}.attn.{n}.bias"] = (D,)
    for n in ("norm_q", "norm_k"):
        s[f"{p}.attn.{n}.weight"] = (dh,)
    return s


def _tte_shapes(p: str, D: int, pooled: int, guidance: bool) -> Dict[str, Tuple[int, ...]]:
    s = {f"{p}.timestep_embedder.linear_1.weight": (D, 256), f"{p}.timestep_embedder.linear_1.bias": (D,),
         f"{p}.timestep_embedder.linear_2.weight": (D, D), f"{p}.timestep_embedder.linear_2.bias": (D,),
         f"{p}.text_embedder.linear_1.weight": (D, pooled), f"{p}.text_embedder.linear_1.bias": (D,),
         f"{p}.text_embedder.linear_2.weight": (D, D), f"{p}.text_embedder.linear_2.bias": (D,)}
    if guidance:
        s.update({f"{p}.guidance_embedder.linear_1.weight": (D, 256), f"{p}.guidance_embedder.linear_1.bias": (D,),
                  f"{p}.guidance_embedder.linear_2.weight": (D, D), f"{p}.guidance_embedder.linear_2.bias": (D,)})
    return s


def state_shapes(cfg: FluxConfig) -> Dict[str, Tuple[int, ...]]:
    """Every parameter of UniGenFlux after init_condition_block, under the reference's state-dict key names."""
    D, dh = cfg.inner_dim, cfg.attention_head_dim
    s: Dict[str, Tuple[int, ...]] = {}
    s["x_embedder.weight"] = (D, cfg.in_channels); s["x_embedder.bias"] = (D,)
    s["context_embedder.weight"] = (D, cfg.joint_attention_dim); s["context_embedder.bias"] = (D,)
    s.update(_tte_shapes("time_text_embed", D, cfg.pooled_projection_dim, cfg.guidance_embeds))
    for i in range(cfg.num_layers):
        s.update(_double_block_shapes(f"transformer_blocks.{i}", D, dh))
    for j in range(cfg.num_single_layers):
        s.update(_single_block_shapes(f"single_transformer_blocks.{j}", D, dh))
    s["norm_out.linear.weight"] = (2 * D, D); s["norm_out.linear.bias"] = (2 * D,)
    s["proj_out.weight"] = (cfg.in_channels, D); s["proj_out.bias"] = (cfg.in_channels,)
    # control modules (src/UniGenTransformer.py:727-773)
    s.update(_tte_shapes("control_time_text_embed", D, cfg.pooled_projection_dim, cfg.guidance_embeds))
    s.update(_tte_shapes("control_condition_embed", D, cfg.pooled_projection_dim, cfg.guidance_embeds))
    s["control_context_embedder.weight"] = (D, D); s["control_context_embedder.bias"] = (D,)
    s["control_x_embedder.weight"] = (D, cfg.in_channels); s["control_x_embedder.bias"] = (D,)
    for m in range(cfg.cn_joint_layers):
        s.update(_double_block_shapes(f"control_joint_trans_blocks.{m}", D, dh))
        s[f"controlnet_add_joint_blocks.{m}.weight"] = (D, D); s[f"controlnet_add_joint_blocks.{m}.bias"] = (D,)
    if cfg.use_single_trans_blocks:
        for m in range(cfg.cn_single_layers):
            s.update(_single_block_shapes(f"control_single_trans_blocks.{m}", D, dh))
            s[f"controlnet_add_single_blocks.{m}.weight"] = (D, D); s[f"controlnet_add_single_blocks.{m}.bias"] = (D,)
    # CoMoE (:833-842, 857)
    s["moe.moe_layer.gate.wg.weight"] = (cfg.expert_nums, D)
    for e in range(cfg.expert_nums):
        p = f"moe.moe_layer.experts.deepspeed_experts.{e}"
        for k in (0, 1):
            s[f"{p}.{k}.0.weight"] = (D, D); s[f"{p}.{k}.0.bias"] = (D,)
            s[f"{p}.{k}.1.weight"] = (D, cfg.pooled_projection_dim); s[f"{p}.{k}.1.bias"] = (D,)
    if cfg.use_shared_expert:
        for k in (0, 1):
            s.update(_double_block_shapes(f"shared_expert.{k}", D, dh))
    if cfg.use_consis_module:
        for k in (0, 1):
            s.update(_double_block_shapes(f"consis_module.{k}", D, dh))
    return s


def make_state(cfg: FluxConfig, seed: int = 0, std: float = 0.02, bias_std: float = 0.0, dtype=torch.bfloat16,
               device: str = "cpu", fast: bool = False) -> State:
    """fast=True tiles one 1M-element random block instead of drawing every weight (timing runs at FLUX width only)."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    st: State = {}
    block = (std * torch.randn(1 << 20, generator=g)).to(dtype) if fast else None
    for name, shape in state_shapes(cfg).items():
        if fast and not name.endswith(".bias") and ".norm_" not in name:
            n = int(torch.Size(shape).numel())
            st[name] = block.repeat((n + block.numel() - 1) // block.numel())[:n].view(shape).to(device)
            continue
        if ".norm_q." in name or ".norm_k." in name or ".norm_added_q." in name or ".norm_added_k." in name:
            t = 1.0 + 0.1 * torch.randn(shape, generator=g) if bias_std > 0 else torch.ones(shape)
        elif name.endswith(".bias"):
            t = bias_std * torch.randn(shape, generator=g) if bias_std > 0 else torch.zeros(shape)
        else:
            t = std * torch.randn(shape, generator=g)
        st[name] = t.to(dtype).to(device)
    return st


def make_ids(h: int, w: int, dtype=torch.bfloat16) -> torch.Tensor:
    """_prepare_latent_image_ids: [h*w, 3] with [:,1]=row, [:,2]=col (diffusers FluxPipeline)."""
    ids = torch.zeros(h, w, 3)
    ids[..., 1] = ids[..., 1] + torch.arange(h)[:, None]
    ids[..., 2] = ids[..., 2] + torch.arange(w)[None, :]
    return ids.reshape(h * w, 3).to(dtype)


def make_inputs(cfg: FluxConfig, B: int, grid: int, T: int, seed: int = 12443, n_cond: int = 1, dtype=torch.bfloat16):
    """Synthetic inputs of SURVEY 8(d): latents, condition tokens ~ N(0,1); prompt embeds ~ 0.1 N(0,1); pooled ~ N(0,1)."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    N = grid * grid
    E = cfg.expert_nums
    inp = dict(
        hidden_states=torch.randn(B, N, cfg.in_channels, generator=g).to(dtype),
        encoder_hidden_states=(0.1 * torch.randn(B, T, cfg.joint_attention_dim, generator=g)).to(dtype),
        pooled_projections=torch.randn(B, cfg.pooled_projection_dim, generator=g).to(dtype),
        img_ids=make_ids(grid, grid, dtype), txt_ids=torch.zeros(T, 3, dtype=dtype),
    )
    conds = [torch.randn(B, N, cfg.in_channels, generator=g).to(dtype) for _ in range(n_cond)]
    cpools = [torch.randn(B, cfg.pooled_projection_dim, generator=g).to(dtype) for _ in range(n_cond)]
    unis = [torch.rand(B * N, E, generator=g) for _ in range(n_cond)]
    if n_cond == 1:
        inp.update(condition_hidden_states=conds[0], condition_pooled_projections=cpools[0], condition_ids=make_ids(grid, grid, dtype))
        inp["gate_uniform"] = unis[0]
    else:
        inp.update(condition_hidden_states=conds, condition_pooled_projections=cpools,
                   condition_ids=[make_ids(grid, grid, dtype) for _ in range(n_cond)])
        inp["gate_uniform"] = unis
    return inp


# =====================================================================================================================
# UniGenSD3 (SD3.5-medium backbone): src/UniGenTransformer.py:21-710 (UniGenBase / UniGenSD3), src/UniGenUtils.py:340-530,
# diffusers 0.32.2 SD3Transformer2DModel / JointTransformerBlock / SD3SingleTransformerBlock / PatchEmbed.
# =====================================================================================================================

@dataclass
class SD3Config:
    """diffusers SD3Transformer2DModel config (SD3.5-medium defaults) + UniGen control params."""
    sample_size: int = 128
    patch_size: int = 2
    in_channels: int = 16
    out_channels: int = 16
    num_layers: int = 24
    attention_head_dim: int = 64
    num_attention_heads: int = 24
    joint_attention_dim: int = 4096
    caption_projection_dim: int = 1536
    pooled_projection_dim: int = 2048
    pos_embed_max_size: int = 384
    dual_attention_layers: Tuple[int, ...] = tuple(range(13))
    qk_norm: Optional[str] = "rms_norm"
    # control params (config/unigen.yaml defaults: transformer-block experts, shared experts, no rope)
    condition_nums: int = 1
    use_modulate: bool = False
    use_pooled_prompt_embeds: bool = True
    use_shared_expert: bool = True
    expert_num_each_condition: int = 3
    expert_num: Optional[int] = None
    top_num: int = 1
    use_rope: bool = False          # only the wiring restates it (ids handed to the injected blocks); the SD3 blocks here have no RoPE

    @property
    def inner_dim(self) -> int:
        return self.num_attention_heads * self.attention_head_dim

    @property
    def expert_nums(self) -> int:
        return self.expert_num if self.expert_num is not None else (self.condition_nums + 1) * self.expert_num_each_condition


def sincos_pos_embed_2d(embed_dim: int, grid_size: int, base_size: int, interpolation_scale: float = 1.0) -> torch.Tensor:
    """diffusers get_2d_sincos_pos_embed(output_type='pt') -> [grid_size^2, embed_dim] fp32 (the PatchEmbed `pos_embed` buffer)."""
    grid_h = torch.arange(grid_size, dtype=torch.float32) / (grid_size / base_size) / interpolation_scale
    grid_w = torch.arange(grid_size, dtype=torch.float32) / (grid_size / base_size) / interpolation_scale
    gw, gh = torch.meshgrid(grid_w, grid_h, indexing="xy")
    grid = torch.stack([gw, gh], dim=0).reshape(2, 1, grid_size, grid_size)

    def one_d(dim, pos):
        omega = torch.arange(dim // 2, dtype=torch.float64) / (dim / 2.0)
        omega = 1.0 / 10000 ** omega
        out = torch.outer(pos.reshape(-1).double(), omega)
        return torch.cat([torch.sin(out), torch.cos(out)], dim=1)

    emb = torch.cat([one_d(embed_dim // 2, grid[0]), one_d(embed_dim // 2, grid[1])], dim=1)
    return emb.float()


def patch_embed(state: State, prefix: str, cfg: SD3Config, latent: torch.Tensor) -> torch.Tensor:
    """diffusers PatchEmbed.forward with pos_embed_max_size: Conv2d(k = s = patch) -> flatten -> + centre-cropped sincos table."""
    p = cfg.patch_size
    h, w = latent.shape[-2] // p, latent.shape[-1] // p
    x = F.conv2d(latent, state[prefix + ".proj.weight"].to(latent.dtype), state[prefix + ".proj.bias"].to(latent.dtype), stride=p)
    x = x.flatten(2).transpose(1, 2)
    mx = cfg.pos_embed_max_size
    pe = state[prefix + ".pos_embed"].reshape(1, mx, mx, -1)
    top, left = (mx - h) // 2, (mx - w) // 2
    pe = pe[:, top:top + h, left:left + w, :].reshape(1, h * w, -1)
    return (x + pe).to(x.dtype)


def adaln_zero_x(state: State, prefix: str, x: torch.Tensor, emb: torch.Tensor):
    """SD35AdaLayerNormZeroX (reference restatement src/UniGenUtils.py:340-352); emb [B, D] or per-token [B, L, D]."""
    e = linear(state, prefix + ".linear", F.silu(emb))
    sh, sc, g, shm, scm, gm, sh2, sc2, g2 = e.chunk(9, dim=-1)
    n = layer_norm(x)
    if e.dim() == x.dim():
        return n * (1 + sc) + sh, g, shm, scm, gm, n * (1 + sc2) + sh2, g2
    return n * (1 + sc[:, None]) + sh[:, None], g, shm, scm, gm, n * (1 + sc2[:, None]) + sh2[:, None], g2


def adaln_zero_any(state: State, prefix: str, x: torch.Tensor, emb: torch.Tensor):
    """AdaLayerNormZero with the reference's per-token extension (src/UniGenUtils.py:354-363)."""
    e = linear(state, prefix + ".linear", F.silu(emb))
    sh, sc, g, shm, scm, gm = e.chunk(6, dim=-1)
    if e.dim() == x.dim():
        return layer_norm(x) * (1 + sc) + sh, g, shm, scm, gm
    return layer_norm(x) * (1 + sc[:, None]) + sh[:, None], g, shm, scm, gm


def _gate(g: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
    return g * y if g.dim() == y.dim() else g.unsqueeze(1) * y


def _mod(n: torch.Tensor, scale: torch.Tensor, shift: torch.Tensor) -> torch.Tensor:
    if scale.dim() == n.dim():
        return n * (1 + scale) + shift
    return n * (1 + scale[:, None]) + shift[:, None]


def sd3_attention(state: State, prefix: str, H: int, x: torch.Tensor, enc: Optional[torch.Tensor], context_pre_only: bool = False):
    """Attention + JointAttnProcessor2_0 (sample-first concat, no RoPE); qk RMSNorm when the weights exist."""
    def nrm(t, name):
        w = state.get(f"{prefix}.{name}.weight")
        return rms_norm(t, w) if w is not None else t
    q = nrm(_heads(linear(state, prefix + ".to_q", x), H), "norm_q")
    k = nrm(_heads(linear(state, prefix + ".to_k", x), H), "norm_k")
    v = _heads(linear(state, prefix + ".to_v", x), H)
    N = x.shape[1]
    if enc is not None:
        eq = nrm(_heads(linear(state, prefix + ".add_q_proj", enc), H), "norm_added_q")
        ek = nrm(_heads(linear(state, prefix + ".add_k_proj", enc), H), "norm_added_k")
        ev = _heads(linear(state, prefix + ".add_v_proj", enc), H)
        q, k, v = torch.cat([q, eq], 2), torch.cat([k, ek], 2), torch.cat([v, ev], 2)
    o = _sdpa(q, k, v).transpose(1, 2).reshape(x.shape[0], -1, x.shape[2]).to(q.dtype)
    if enc is None:
        return linear(state, prefix + ".to_out.0", o), None
    xo, eo = o[:, :N], o[:, N:]
    xo = linear(state, prefix + ".to_out.0", xo)
    eo = None if context_pre_only else linear(state, prefix + ".to_add_out", eo)
    return xo, eo


def sd3_joint_block(state: State, prefix: str, H: int, x, enc, temb, context_pre_only: bool = False, dual: bool = False):
    """JointTransformerBlock.forward as restated by the reference (src/UniGenUtils.py:440-522). Returns (enc or None, x)."""
    if dual:
        n, g, shm, scm, gm, n2, g2 = adaln_zero_x(state, prefix + ".norm1", x, temb)
    else:
        n, g, shm, scm, gm = adaln_zero_any(state, prefix + ".norm1", x, temb)
    if context_pre_only:
        nc = adaln_continuous(state, prefix + ".norm1_context", enc, temb)
    else:
        nc, cg, cshm, cscm, cgm = adaln_zero_any(state, prefix + ".norm1_context", enc, temb)
    a, ca = sd3_attention(state, prefix + ".attn", H, n, nc, context_pre_only)
    x = x + _gate(g, a)
    if dual:
        a2, _ = sd3_attention(state, prefix + ".attn2", H, n2, None)
        x = x + _gate(g2, a2)
    x = x + _gate(gm, feed_forward(state, prefix + ".ff", _mod(layer_norm(x), scm, shm)))
    if context_pre_only:
        return None, x
    enc = enc + _gate(cg, ca)
    enc = enc + _gate(cgm, feed_forward(state, prefix + ".ff_context", _mod(layer_norm(enc), cscm, cshm)))
    return enc, x


def sd3_single_block(state: State, prefix: str, H: int, x, temb):
    """SD3SingleTransformerBlock.forward (src/UniGenUtils.py:386-414); temb per sample [B, D] or per token [B, L, D]."""
    n, g, shm, scm, gm = adaln_zero_any(state, prefix + ".norm1", x, temb)
    a, _ = sd3_attention(state, prefix + ".attn", H, n, None)
    x = x + _gate(g, a)
    return x + _gate(gm, feed_forward(state, prefix + ".ff", _mod(layer_norm(x), scm, shm)))


def _sd3_experts(state: State, cfg: SD3Config, x, c, control_temb, condition_temb, pooled, cond_pooled, uniform, choice=None):
    """MOELayer.forward + UniGenBase.expert_forward (src/UniGenTransformer.py:225-267): modulated linears, or two SD3SingleTransformerBlocks per
    expert fed per-token tembs, attending over the expert's capacity slots. Returns (expert_h, expert_c, l_aux, exp_counts, routing)."""
    B, N, D = x.shape
    E, S, H, dt = cfg.expert_nums, B * N, cfg.num_attention_heads, x.dtype
    if cfg.use_modulate:
        fcfg = FluxConfig(attention_head_dim=cfg.attention_head_dim, num_attention_heads=H, condition_nums=cfg.condition_nums,
                          expert_num_each_condition=cfg.expert_num_each_condition, expert_num=cfg.expert_num, top_num=cfg.top_num)
        return comoe_experts(state, fcfg, x, c, pooled, cond_pooled, None, uniform, choice=choice)
    logits = F.linear((x + c if choice is None else choice).reshape(S, D).float(), state["moe.moe_layer.gate.wg.weight"].float())
    l_aux, combine_weights, _, exp_counts, routing = gate_route(logits, uniform, cfg.top_num)
    C, tos = routing["capacity"], routing["token_of_slot"]

    def dispatch(t2d):
        out = torch.zeros(E, C, t2d.shape[-1], dtype=t2d.dtype)
        valid = tos >= 0
        out[valid] = t2d[tos[valid]]
        return out

    xd, cd = dispatch(x.reshape(S, D)), dispatch(c.reshape(S, D))
    td = dispatch(control_temb.unsqueeze(1).expand(-1, N, -1).reshape(S, D))
    ctd = dispatch(condition_temb.unsqueeze(1).expand(-1, N, -1).reshape(S, D))
    yh, yc = [], []
    for e in range(E):
        p = f"moe.moe_layer.experts.deepspeed_experts.{e}"
        yh.append(sd3_single_block(state, p + ".0", H, xd[e][None], td[e][None])[0])        # expert[0](hidden, temb)        (:261)
        yc.append(sd3_single_block(state, p + ".1", H, cd[e][None], ctd[e][None])[0])       # expert[1](condition, cond_temb) (:262)
    cw = combine_weights.to(dt)
    eh = torch.einsum("sec,ecm->sm", cw, torch.stack(yh)).reshape(B, N, D)
    ec = torch.einsum("sec,ecm->sm", cw, torch.stack(yc)).reshape(B, N, D)
    return eh, ec, l_aux, exp_counts, routing


# ---------------------------------------------------------------------------------------------------------------------
# Wiring of UniGenSD3 against injected module callables `m` (REFERENCE-PINNED like the FLUX wiring above: tests/test_ref_wiring_cpu.py).
# UniGenBase.moe_forward (src/UniGenTransformer.py:269-296) is flux_moe_forward's text without the consistency module - the same function.
#   m.block(i, x, enc, temb) -> (enc | None, x)          transformer_blocks[i]
#   m.control_block(k, z, enc, temb, hd_ids, encoder_hd_ids) -> (enc, z);   m.add(k, z)
#   m.control_pos_embed_input(latent), m.control_context_embedder(enc), m.control_time_text_embed(t, pooled), m.control_condition_embed(t, pooled)
#   m.moe_layer(...), m.shared_expert(k, x, enc, temb, hd_ids, encoder_hd_ids)  as for FLUX
# ---------------------------------------------------------------------------------------------------------------------

def sd3_preprocess_moe_forward(m, cfg: SD3Config, x, condition_hidden_states, enc, pooled, cond_pooled, timestep, *, ids=None, trace=None):
    """UniGenSD3.preprocess_moe_forward (src/UniGenTransformer.py:498-537)."""
    c = m.control_pos_embed_input(condition_hidden_states)
    control_pooled = pooled if cfg.use_pooled_prompt_embeds else torch.zeros_like(pooled)
    control_temb = m.control_time_text_embed(timestep, control_pooled)
    condition_temb = m.control_condition_embed(timestep, cond_pooled)
    ctrl_enc = m.control_context_embedder(enc)
    (oh, oc), l_aux, exp_counts = flux_moe_forward(m, cfg, x, c, ctrl_enc=ctrl_enc, control_temb=control_temb, condition_temb=condition_temb,
                                                   pooled=pooled, cond_pooled=cond_pooled, ids=ids if cfg.use_rope else None)
    z0 = oh + oc                                                                         # control_forward :563
    if trace is not None:
        trace["z0"] = z0
    return dict(z0=z0, ctrl_enc=ctrl_enc, control_temb=control_temb, condition_temb=condition_temb, l_aux=l_aux, exp_counts=exp_counts)


def sd3_base_forward(m, cfg: SD3Config, x, condition_hidden_states, enc, pooled, cond_pooled, timestep, *, conditioning_scale, temb,
                     n_control: Optional[int] = None, ids=None, trace=None):
    """UniGenSD3.base_forward (src/UniGenTransformer.py:581-623) with control_forward (:539-579, the use_encoder_hidden_states branch - the only one
    UniGenSD3.init_control_block admits, :496) inlined: base block i -> control block int(i / (L / n_control)) on the BASE stream (the first one on
    the CoMoE output) with the fixed ctrl_enc / condition_temb -> zero-res add. Returns (x, enc, moe_out)."""
    L = cfg.num_layers
    n_control = L if n_control is None else n_control
    rope = cfg.use_rope and ids is not None
    moe_out = None
    for i in range(L):
        enc, x = m.block(i, x, enc, temb)
        k = int(i / (L / n_control))                                                     # :552-553
        if i == 0:                                                                       # :559-563, text stream AFTER base block 0
            moe_out = sd3_preprocess_moe_forward(m, cfg, x, condition_hidden_states, enc, pooled, cond_pooled, timestep, ids=ids, trace=trace)
            z_in = moe_out["z0"]
        else:
            z_in = x
        _, z = m.control_block(k, z_in, moe_out["ctrl_enc"], moe_out["condition_temb"], ids["img_ids"] if rope else None, ids["prompt_ids"] if rope else None)
        x = x + m.add(k, z) * conditioning_scale                                         # :579, :614
    return x, enc, moe_out


def sd3_modules(state: State, cfg: SD3Config, gate_uniform, trace: Optional[dict] = None):
    """The injected callables of the SD3 wiring, built from a flat state dict. (use_rope is not restated for SD3: the shipped yaml has none.)"""
    from types import SimpleNamespace
    m, H, L = SimpleNamespace(), cfg.num_attention_heads, cfg.num_layers

    def moe_layer(*, choice_expert_input, hidden_states, condition_hidden_states, encoder_hidden_states, temb, condition_temb,
                  condition_pooled_projections, pooled_projections):
        eh, ec, l_aux, exp_counts, routing = _sd3_experts(state, cfg, hidden_states, condition_hidden_states, temb, condition_temb,
                                                          pooled_projections, condition_pooled_projections, gate_uniform, choice=choice_expert_input)
        if trace is not None:
            trace["routing"] = routing
        return eh, ec, l_aux, exp_counts

    m.block = lambda i, x, enc, temb: sd3_joint_block(state, f"transformer_blocks.{i}", H, x, enc, temb, context_pre_only=i == L - 1,
                                                      dual=i in cfg.dual_attention_layers)
    m.control_block = lambda k, z, enc, temb, hd_ids, ehd_ids: sd3_joint_block(state, f"control_transformer_blocks.{k}", H, z, enc, temb,
                                                                               context_pre_only=False, dual=k in cfg.dual_attention_layers)
    m.add = lambda k, z: linear(state, f"controlnet_add_blocks.{k}", z)
    m.control_pos_embed_input = lambda lat: patch_embed(state, "control_pos_embed_input", cfg, lat)
    m.control_context_embedder = lambda e: linear(state, "control_context_embedder", e)
    m.control_time_text_embed = lambda t, p: time_text_embed(state, "control_time_text_embed", t, p)
    m.control_condition_embed = lambda t, p: time_text_embed(state, "control_condition_embed", t, p)
    m.moe_layer = moe_layer
    # shared experts (:205-222): JointTransformerBlock(context_pre_only=False) then JointTransformerBlock(context_pre_only=True, dual attention)
    m.shared_expert = lambda k, x, enc, temb, hd_ids, ehd_ids: sd3_joint_block(state, f"shared_expert.{k}", H, x, enc, temb,
                                                                               context_pre_only=k == 1, dual=k == 1)
    return m


def sd3_comoe(state: State, cfg: SD3Config, x, c, ctrl_enc, control_temb, condition_temb, pooled, cond_pooled, uniform):
    """UniGenBase.moe_forward + expert_forward (src/UniGenTransformer.py:225-296) through the pinned wiring; c = embedded condition tokens."""
    trace: dict = {}
    m = sd3_modules(state, cfg, uniform, trace=trace)
    (oh, oc), l_aux, exp_counts = flux_moe_forward(m, cfg, x, c, ctrl_enc=ctrl_enc, control_temb=control_temb, condition_temb=condition_temb,
                                                   pooled=pooled, cond_pooled=cond_pooled, ids=None)
    return oh, oc, l_aux, exp_counts, trace["routing"]


def unigen_sd3_forward(state: State, cfg: SD3Config, *, hidden_states, condition_hidden_states, encoder_hidden_states, pooled_projections,
                       condition_pooled_projections, timestep, conditioning_scale: float = 1.0, gate_uniform=None, dtype=torch.bfloat16,
                       io_dtype=torch.bfloat16, trace: Optional[dict] = None):
    """UniGenSD3.forward (src/UniGenTransformer.py:625-710): latents NCHW in, NCHW out. The timestep is used as given (no x1000)."""
    dt = dtype
    cast = lambda t: t.to(dt)
    height, width = hidden_states.shape[-2:]
    x = patch_embed(state, "pos_embed", cfg, cast(hidden_states))
    pooled = cast(pooled_projections)
    cpooled = cast(condition_pooled_projections)
    tstep = timestep.float()      # SD3 passes the scheduler timestep unscaled; time_proj works on timesteps.float()
    temb = time_text_embed(state, "time_text_embed", tstep, pooled)
    enc = linear(state, "context_embedder", cast(encoder_hidden_states))
    m = sd3_modules(state, cfg, gate_uniform, trace=trace)
    x, _, moe_out = sd3_base_forward(m, cfg, x, cast(condition_hidden_states), enc, pooled, cpooled, tstep, conditioning_scale=conditioning_scale, temb=temb,
                                     trace=trace)
    x = adaln_continuous(state, "norm_out", x, temb)
    x = linear(state, "proj_out", x)
    p = cfg.patch_size
    h, w = height // p, width // p
    x = x.reshape(x.shape[0], h, w, p, p, cfg.out_channels)
    x = torch.einsum("nhwpqc->nchpwq", x)
    out = x.reshape(x.shape[0], cfg.out_channels, h * p, w * p)
    return out, dict(moe_loss=moe_out["l_aux"] * 0.1), dict(expert_counts=moe_out["exp_counts"])


def _sd3_attn_shapes(s, p, D, dh, qk_norm, ctx: bool, context_pre_only: bool = False):
    for n in ("to_q", "to_k", "to_v", "to_out.0"):
        s[f"{p}.{n}.weight"] = (D, D); s[f"{p}.{n}.bias"] = (D,)
    if qk_norm:
        s[f"{p}.norm_q.weight"] = (dh,); s[f"{p}.norm_k.weight"] = (dh,)
    if ctx:
        for n in ("add_q_proj", "add_k_proj", "add_v_proj"):
            s[f"{p}.{n}.weight"] = (D, D); s[f"{p}.{n}.bias"] = (D,)
        if not context_pre_only:
            s[f"{p}.to_add_out.weight"] = (D, D); s[f"{p}.to_add_out.bias"] = (D,)
        if qk_norm:
            s[f"{p}.norm_added_q.weight"] = (dh,); s[f"{p}.norm_added_k.weight"] = (dh,)


def _sd3_ff_shapes(s, p, D):
    s[f"{p}.net.0.proj.weight"] = (4 * D, D); s[f"{p}.net.0.proj.bias"] = (4 * D,)
    s[f"{p}.net.2.weight"] = (D, 4 * D); s[f"{p}.net.2.bias"] = (D,)


def _sd3_joint_shapes(s, p, D, dh, qk_norm, context_pre_only, dual):
    s[f"{p}.norm1.linear.weight"] = ((9 if dual else 6) * D, D); s[f"{p}.norm1.linear.bias"] = ((9 if dual else 6) * D,)
    nctx = 2 if context_pre_only else 6
    s[f"{p}.norm1_context.linear.weight"] = (nctx * D, D); s[f"{p}.norm1_context.linear.bias"] = (nctx * D,)
    _sd3_attn_shapes(s, p + ".attn", D, dh, qk_norm, True, context_pre_only)
    if dual:
        _sd3_attn_shapes(s, p + ".attn2", D, dh, qk_norm, False)
    _sd3_ff_shapes(s, p + ".ff", D)
    if not context_pre_only:
        _sd3_ff_shapes(s, p + ".ff_context", D)


def sd3_state_shapes(cfg: SD3Config) -> Dict[str, Tuple[int, ...]]:
    D, dh, p = cfg.inner_dim, cfg.attention_head_dim, cfg.patch_size
    s: Dict[str, Tuple[int, ...]] = {}
    for pe in ("pos_embed", "control_pos_embed_input"):
        s[pe + ".proj.weight"] = (D, cfg.in_channels, p, p); s[pe + ".proj.bias"] = (D,)
        s[pe + ".pos_embed"] = (1, cfg.pos_embed_max_size ** 2, D)
    for t in ("time_text_embed", "control_time_text_embed", "control_condition_embed"):
        s.update(_tte_shapes(t, D, cfg.pooled_projection_dim, False))
    s["context_embedder.weight"] = (cfg.caption_projection_dim, cfg.joint_attention_dim); s["context_embedder.bias"] = (cfg.caption_projection_dim,)
    s["control_context_embedder.weight"] = (D, D); s["control_context_embedder.bias"] = (D,)
    for i in range(cfg.num_layers):
        dual = i in cfg.dual_attention_layers
        _sd3_joint_shapes(s, f"transformer_blocks.{i}", D, dh, cfg.qk_norm, i == cfg.num_layers - 1, dual)
        _sd3_joint_shapes(s, f"control_transformer_blocks.{i}", D, dh, cfg.qk_norm, False, dual)
        s[f"controlnet_add_blocks.{i}.weight"] = (D, D); s[f"controlnet_add_blocks.{i}.bias"] = (D,)
    s["norm_out.linear.weight"] = (2 * D, D); s["norm_out.linear.bias"] = (2 * D,)
    s["proj_out.weight"] = (p * p * cfg.out_channels, D); s["proj_out.bias"] = (p * p * cfg.out_channels,)
    s["moe.moe_layer.gate.wg.weight"] = (cfg.expert_nums, D)
    for e in range(cfg.expert_nums):
        pe = f"moe.moe_layer.experts.deepspeed_experts.{e}"
        for k in (0, 1):
            if cfg.use_modulate:
                s[f"{pe}.{k}.0.weight"] = (D, D); s[f"{pe}.{k}.0.bias"] = (D,)
                s[f"{pe}.{k}.1.weight"] = (D, cfg.pooled_projection_dim); s[f"{pe}.{k}.1.bias"] = (D,)
            else:
                s[f"{pe}.{k}.norm1.linear.weight"] = (6 * D, D); s[f"{pe}.{k}.norm1.linear.bias"] = (6 * D,)
                _sd3_attn_shapes(s, f"{pe}.{k}.attn", D, dh, None, False)
                _sd3_ff_shapes(s, f"{pe}.{k}.ff", D)
    if cfg.use_shared_expert:
        _sd3_joint_shapes(s, "shared_expert.0", D, dh, cfg.qk_norm, False, False)
        _sd3_joint_shapes(s, "shared_expert.1", D, dh, cfg.qk_norm, True, True)
    return s


def make_sd3_state(cfg: SD3Config, seed: int = 0, std: float = 0.02, bias_std: float = 0.0, dtype=torch.bfloat16) -> State:
    g = torch.Generator(device="cpu").manual_seed(seed)
    st: State = {}
    base = cfg.sample_size // cfg.patch_size
    for name, shape in sd3_state_shapes(cfg).items():
        if name.endswith(".pos_embed"):
            st[name] = sincos_pos_embed_2d(cfg.inner_dim, cfg.pos_embed_max_size, base).unsqueeze(0)      # fp32 buffer
            continue
        if ".norm_q." in name or ".norm_k." in name or ".norm_added_q." in name or ".norm_added_k." in name:
            t = 1.0 + 0.1 * torch.randn(shape, generator=g) if bias_std > 0 else torch.ones(shape)
        elif name.endswith(".bias"):
            t = bias_std * torch.randn(shape, generator=g) if bias_std > 0 else torch.zeros(shape)
        else:
            t = std * torch.randn(shape, generator=g)
        st[name] = t.to(dtype)
    return st


def make_sd3_inputs(cfg: SD3Config, B: int, hw: int, T: int, seed: int = 12443, dtype=torch.bfloat16):
    """hw = latent height = width (pixels / 8); N = (hw / patch)^2 tokens."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    N = (hw // cfg.patch_size) ** 2
    return dict(
        hidden_states=torch.randn(B, cfg.in_channels, hw, hw, generator=g).to(dtype),
        condition_hidden_states=torch.randn(B, cfg.in_channels, hw, hw, generator=g).to(dtype),
        encoder_hidden_states=(0.1 * torch.randn(B, T, cfg.joint_attention_dim, generator=g)).to(dtype),
        pooled_projections=torch.randn(B, cfg.pooled_projection_dim, generator=g).to(dtype),
        condition_pooled_projections=torch.randn(B, cfg.pooled_projection_dim, generator=g).to(dtype),
        gate_uniform=torch.rand(B * N, cfg.expert_nums, generator=g),
    )
