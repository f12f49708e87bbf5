"""Host side of the UniGenFlux / MultiCondtionUniGenFlux forward on MI355X.

Mirrors the reference's call surface (src/UniGenTransformer.py:712-1450 of gavin-gqzhang/UniGen): same class names, constructor
flow (`from_pretrained` / `from_config` -> `.to()` -> `init_condition_block(condition_nums, condition_types, control_params)`),
forward kwargs, 3-tuple return and state-dict key names. The arithmetic is NOT torch: every projection, norm, attention and
routing step is a call through the C ABI of libunigen_hip.so (unigen_amd/ops.py). torch supplies device memory, the
parameter containers (so `state_dict` / `load_state_dict` / `.to` behave as callers expect) and the stream.

What the reference leaves undefined (FluxJointRoPETransformerBlock / FluxSingleRoPETransformerBlock) is defined here
as: parameters and math of the diffusers Flux double / single block, RoPE tables from FluxPosEmbed rounded to the ids dtype as
JointAttnRopeProcessor does (src/UniGenUtils.py:597). Work whose result every caller discards (text-stream output of control
joint blocks and of shared_expert[1]) is skipped, and a control block's text K/V are computed once per step, not once per use.
"""
from __future__ import annotations

import json
import math
import os
from types import SimpleNamespace
from typing import Any, Dict, List, Optional, Sequence, Tuple

import torch
from torch import nn

from . import lib as L
from . import ops
from .engine import BF, HipModule, _Holder, _register, _Stream, _Workspace, reference_init_
from .ops import RowMap


FLUX_SCHNELL_CONFIG = dict(patch_size=1, in_channels=64, num_layers=19, num_single_layers=38, attention_head_dim=128,
                           num_attention_heads=24, joint_attention_dim=4096, pooled_projection_dim=768, guidance_embeds=False,
                           axes_dims_rope=(16, 56, 56))


# ----------------------------------------------------------------------------------------------------------------------
# parameter shapes under the reference's state-dict key names
# ----------------------------------------------------------------------------------------------------------------------

def _lin(s: dict, name: str, out_f: int, in_f: int) -> None:
    s[name + ".weight"] = (out_f, in_f)
    s[name + ".bias"] = (out_f,)


def _double_block_shapes(s: dict, p: str, D: int, dh: int) -> None:
    for n in ("norm1", "norm1_context"):
        _lin(s, f"{p}.{n}.linear", 6 * D, D)
    for n in ("to_q", "to_k", "to_v", "add_q_proj", "add_k_proj", "add_v_proj", "to_out.0", "to_add_out"):
        _lin(s, f"{p}.attn.{n}", D, D)
    for n in ("norm_q", "norm_k", "norm_added_q", "norm_added_k"):
        s[f"{p}.attn.{n}.weight"] = (dh,)
    for n in ("ff", "ff_context"):
        _lin(s, f"{p}.{n}.net.0.proj", 4 * D, D)
        _lin(s, f"{p}.{n}.net.2", D, 4 * D)


class _RopeTab(tuple):
    """(cos, sin) [S, dh] fp32 as the callers unpack it, plus `.cs` [S, dh/2, 2]: the same values once per rotation pair."""
    cs = None


def _single_block_shapes(s: dict, p: str, D: int, dh: int) -> None:
    _lin(s, f"{p}.norm.linear", 3 * D, D)
    _lin(s, f"{p}.proj_mlp", 4 * D, D)
    _lin(s, f"{p}.proj_out", D, 5 * D)
    for n in ("to_q", "to_k", "to_v"):
        _lin(s, f"{p}.attn.{n}", D, D)
    for n in ("norm_q", "norm_k"):
        s[f"{p}.attn.{n}.weight"] = (dh,)


def _time_text_embed_shapes(s: dict, p: str, D: int, pooled: int, guidance: bool) -> None:
    _lin(s, f"{p}.timestep_embedder.linear_1", D, 256)
    _lin(s, f"{p}.timestep_embedder.linear_2", D, D)
    if guidance:
        _lin(s, f"{p}.guidance_embedder.linear_1", D, 256)
        _lin(s, f"{p}.guidance_embedder.linear_2", D, D)
    _lin(s, f"{p}.text_embedder.linear_1", D, pooled)
    _lin(s, f"{p}.text_embedder.linear_2", D, D)


def base_param_shapes(cfg) -> Dict[str, Tuple[int, ...]]:
    """diffusers FluxTransformer2DModel parameters."""
    D, dh = cfg.num_attention_heads * cfg.attention_head_dim, cfg.attention_head_dim
    s: Dict[str, Tuple[int, ...]] = {}
    _lin(s, "x_embedder", D, cfg.in_channels)
    _lin(s, "context_embedder", D, cfg.joint_attention_dim)
    _time_text_embed_shapes(s, "time_text_embed", D, cfg.pooled_projection_dim, cfg.guidance_embeds)
    for i in range(cfg.num_layers):
        _double_block_shapes(s, f"transformer_blocks.{i}", D, dh)
    for j in range(cfg.num_single_layers):
        _single_block_shapes(s, f"single_transformer_blocks.{j}", D, dh)
    _lin(s, "norm_out.linear", 2 * D, D)
    _lin(s, "proj_out", cfg.in_channels, D)
    return s


def control_param_shapes(cfg, ctl) -> Dict[str, Tuple[int, ...]]:
    """Modules created by init_control_block / init_moe_block (src/UniGenTransformer.py:717-923)."""
    D, dh = cfg.num_attention_heads * cfg.attention_head_dim, cfg.attention_head_dim
    s: Dict[str, Tuple[int, ...]] = {}
    _time_text_embed_shapes(s, "control_time_text_embed", D, cfg.pooled_projection_dim, cfg.guidance_embeds)
    _time_text_embed_shapes(s, "control_condition_embed", D, cfg.pooled_projection_dim, cfg.guidance_embeds)
    _lin(s, "control_context_embedder", D, D)
    _lin(s, "control_x_embedder", D, cfg.in_channels)
    for m in range(ctl.cn_joint_layers):
        _double_block_shapes(s, f"control_joint_trans_blocks.{m}", D, dh)
        _lin(s, f"controlnet_add_joint_blocks.{m}", D, D)
    if ctl.use_single_trans_blocks:
        for m in range(ctl.cn_single_layers):
            _single_block_shapes(s, f"control_single_trans_blocks.{m}", D, dh)
            _lin(s, f"controlnet_add_single_blocks.{m}", D, D)
    s["moe.moe_layer.gate.wg.weight"] = (ctl.expert_nums, D)
    for e in range(ctl.expert_nums):
        p = f"moe.moe_layer.experts.deepspeed_experts.{e}"
        for k in (0, 1):        # [0] = condition-modulate, [1] = hidden-modulate (:833-842, 956)
            _lin(s, f"{p}.{k}.0", D, D)
            _lin(s, f"{p}.{k}.1", D, cfg.pooled_projection_dim)
    if ctl.use_shared_expert:
        for k in (0, 1):
            _double_block_shapes(s, f"shared_expert.{k}", D, dh)
    if getattr(ctl, "use_consis_module", False):
        for k in (0, 1):          # both are built (:909-920); only consis_module[0] is ever called (:992,996)
            _double_block_shapes(s, f"consis_module.{k}", D, dh)
    return s


class UniGenFlux(HipModule):
    """Drop-in for the reference `UniGenFlux(FluxTransformer2DModel)`."""

    multi_condition = False

    # ------------------------------------------------------------------ construction ---------------------------------
    def __init__(self, config: Optional[dict] = None, device=None, dtype=BF, _init: bool = True, **kwargs):
        super().__init__()
        c = dict(FLUX_SCHNELL_CONFIG)
        c.update(config or {})
        c.update(kwargs)
        c["axes_dims_rope"] = tuple(c["axes_dims_rope"])
        self.config = SimpleNamespace(**c)
        self.inner_dim = self.config.num_attention_heads * self.config.attention_head_dim
        self.out_channels = self.config.in_channels
        if sum(self.config.axes_dims_rope) != self.config.attention_head_dim:
            raise ValueError("axes_dims_rope must sum to attention_head_dim")
        self._heads, self._head_dim = self.config.num_attention_heads, self.config.attention_head_dim
        self._ctl = None
        self._rope_cache: Dict[Tuple, Tuple[torch.Tensor, torch.Tensor]] = {}
        for name, shape in base_param_shapes(self.config).items():
            _register(self, name, shape, device, dtype)
        if _init:      # from_config = a fresh model: torch's default initial values (from_pretrained overwrites, so it skips this)
            reference_init_(self, list(base_param_shapes(self.config)), copies={}, zero_prefixes=(), expert_prefix="\0")

    @classmethod
    def from_config(cls, config: dict, **kw) -> "UniGenFlux":
        return cls(config, **kw)

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path, subfolder: Optional[str] = None, revision=None, variant=None,
                        torch_dtype=BF, device=None, **kw) -> "UniGenFlux":
        """Local directory only (no hub access): config.json + diffusion_pytorch_model*.safetensors (infer.py:115-119)."""
        path = os.fspath(pretrained_model_name_or_path)
        if subfolder:
            path = os.path.join(path, subfolder)
        if not os.path.isdir(path):
            raise OSError(f"{path} is not a local directory (unigen_amd loads checkpoints from disk only)")
        with open(os.path.join(path, "config.json")) as f:
            raw = json.load(f)
        cfg = {k: raw[k] for k in FLUX_SCHNELL_CONFIG if k in raw}
        model = cls(cfg, device=device, dtype=torch_dtype, _init=False)
        from safetensors.torch import load_file
        files = sorted(f for f in os.listdir(path) if f.endswith(".safetensors"))
        if not files:
            raise OSError(f"no *.safetensors weights under {path}")
        sd = {}
        for fn in files:
            sd.update(load_file(os.path.join(path, fn)))
        res = model.load_state_dict(sd, strict=False)
        if res.missing_keys:
            raise RuntimeError(f"checkpoint misses base parameters: {res.missing_keys[:8]} ...")
        return model

    def init_condition_block(self, condition_nums: int = 1, condition_types: Optional[List[str]] = None, **kwargs):
        """src/UniGenTransformer.py:713-715 -> init_control_block(control_params) -> init_moe_block."""
        self.condition_nums = condition_nums
        self.condition_types = condition_types
        control_params = kwargs.get("control_params", None)
        assert control_params is not None, ValueError("Please provice control net model parameter")
        get = control_params.get
        use_rope, use_modulate = bool(get("use_rope", False)), bool(get("use_modulate", False))
        if not (use_rope or use_modulate):
            # the shipped yaml yields stock single blocks fed per-token temb, which cannot run.
            raise ValueError("UniGenFlux needs control_params.use_rope or control_params.use_modulate (modulated experts); "
                             "the transformer-block expert variant of config/unigen.yaml is not executable in the reference either")
        if get("cn2base_method", "add") != "add":
            raise ValueError("only cn2base_method='add' is supported (the CrossAttn variant is dead code in the reference)")
        dev = int(get("single_control_dev", 2))
        expert_num = get("expert_num", None)
        ctl = SimpleNamespace(
            use_rope=use_rope, use_modulate=use_modulate,
            use_pooled_prompt_embeds=bool(get("use_pooled_prompt_embeds", True)),
            use_encoder_hidden_states=bool(get("use_encoder_hidden_states", True)),
            use_shared_expert=bool(get("use_shared_expert", False)),
            use_consis_module=bool(get("use_consis_module", False)),
            use_single_trans_blocks=bool(get("use_single_trans_blocks", True)),
            single_block_control_method=get("single_block_control_method", "overall_add"),
            cn_joint_layers=self.config.num_layers // dev, cn_single_layers=self.config.num_single_layers // dev,
            expert_nums=int(expert_num) if expert_num is not None else (condition_nums + 1) * int(get("expert_num_each_condition", 3)),
            top_k=int(get("top_num", 1)),
        )
        if not 1 <= ctl.top_k <= max(ctl.expert_nums, 1):
            raise ValueError(f"top_num {ctl.top_k} must be in 1 .. expert count {ctl.expert_nums} (deepspeed top1gating / top2gating / topkgating)")
        if ctl.top_k == 2 and ctl.expert_nums < 2:
            raise ValueError("top_num = 2 needs at least two experts")
        if ctl.expert_nums > 16:
            raise ValueError("at most 16 experts are supported")
        self._ctl = ctl
        dev_, dt_ = self.device, self.dtype
        shapes = control_param_shapes(self.config, ctl)
        for name, shape in shapes.items():
            _register(self, name, shape, dev_, dt_)
        reference_init_(self, list(shapes), copies={"control_time_text_embed.": "time_text_embed.", "control_condition_embed.": "time_text_embed.",
                                                    "control_x_embedder.": "x_embedder."},
                        zero_prefixes=("controlnet_add_joint_blocks.", "controlnet_add_single_blocks."), expert_prefix="moe.moe_layer.experts.deepspeed_experts.")
        names = ["control_time_text_embed", "control_condition_embed", "control_context_embedder", "control_x_embedder",
                 "control_joint_trans_blocks", "controlnet_add_joint_blocks", "moe"]
        if ctl.use_single_trans_blocks:
            names += ["control_single_trans_blocks", "controlnet_add_single_blocks"]
        if ctl.use_shared_expert:
            names.append("shared_expert")
        if ctl.use_consis_module:
            names.append("consis_module")
        self.trainable_control_modules = {n: getattr(self, n) for n in names}
        if get("use_transformer_params", False):
            self.init_control_param()
        self._packed.clear()
        self._pname_cache = None

    def init_control_param(self):
        """src/UniGenTransformer.py:789-804: start the control blocks from the base weights."""
        sd = self.state_dict()
        with torch.no_grad():
            for k, v in sd.items():
                for src, dst in (("time_text_embed.", "control_time_text_embed."), ("time_text_embed.", "control_condition_embed.")):
                    if k.startswith(src):
                        sd[dst + k[len(src):]].copy_(v)
                if k.startswith("transformer_blocks."):
                    i = int(k.split(".")[1])
                    if i < self._ctl.cn_joint_layers:
                        sd["control_joint_trans_blocks." + k[len("transformer_blocks."):]].copy_(v)
                if k.startswith("single_transformer_blocks.") and self._ctl.use_single_trans_blocks:
                    j = int(k.split(".")[1])
                    if j < self._ctl.cn_single_layers:
                        sd["control_single_trans_blocks." + k[len("single_transformer_blocks."):]].copy_(v)

    # ------------------------------------------------------------------ flux-specific pieces --------------------------
    def _rope(self, ids_list: Sequence[torch.Tensor], round_to: Optional[torch.dtype]) -> Tuple[torch.Tensor, torch.Tensor]:
        """FluxPosEmbed(theta=10000, axes_dims_rope) on cat(ids): fp32 cos/sin [S, dh]; `round_to` applies the control path's
        cast to the ids dtype (src/UniGenUtils.py:597). Tables are tiny and cached per ids identity."""
        # Keyed on tensor identity AND kept alive by the entry: while an ids tensor is referenced here its storage cannot be recycled for
        # another grid's ids of the same shape, and `_version` catches in-place edits (a cache keyed on data_ptr alone returned the tables
        # of a freed 1024x512 grid for a fresh 512x1024 one of equal N).
        key = tuple((id(t), t.data_ptr(), tuple(t.shape), t.dtype, t._version) for t in ids_list) + (round_to,)
        hit = self._rope_cache.get(key)
        if hit is not None:
            return hit[0]
        ids = torch.cat([t.to(self.device) for t in ids_list], dim=0).float()
        cos_out, sin_out = [], []
        for i, d in enumerate(self.config.axes_dims_rope):
            freqs = 1.0 / (10000.0 ** (torch.arange(0, d, 2, dtype=torch.float64, device=self.device) / d))
            ang = torch.outer(ids[:, i].to(torch.float64), freqs)
            cos_out.append(ang.cos().repeat_interleave(2, dim=1).float())
            sin_out.append(ang.sin().repeat_interleave(2, dim=1).float())
        cos, sin = torch.cat(cos_out, -1), torch.cat(sin_out, -1)
        if round_to is not None and round_to != torch.float32:
            cos, sin = cos.to(round_to).float(), sin.to(round_to).float()
        out = _RopeTab((cos.contiguous(), sin.contiguous()))
        # one (cos, sin) per rotation pair, interleaved: the table the fused QKV epilogue reads (UG_EPI_QKV_ROPE)
        out.cs = torch.stack([out[0][:, 0::2], out[1][:, 0::2]], dim=-1).contiguous()
        if len(self._rope_cache) > 64:
            self._rope_cache.clear()
        self._rope_cache[key] = (out, tuple(ids_list))      # the ids tensors stay referenced (see the key)
        return out

    def _single_block(self, p: str, B: int, h_in: _Stream, h_out: _Stream, temb: torch.Tensor,
                      rope: Optional[Tuple[torch.Tensor, torch.Tensor]]) -> None:
        """diffusers FluxSingleTransformerBlock (control twin A7): h = h + gate * proj_out(cat[attn(n), gelu(proj_mlp(n))])."""
        D, H, dh = self.inner_dim, self.config.num_attention_heads, self.config.attention_head_dim
        Lj = h_in.Ls
        a = p + ".attn"
        cos, sin = rope if rope is not None else (None, None)
        emb = self._adaln_emb(p + ".norm", temb, 3, "1")
        n = self._modulate(h_in, emb, 0, 1, B, "j")
        sb = self._w("single", (B * Lj, 8 * D))       # [q | k | v | attn | mlp(4D)]
        # to_q / to_k / to_v and proj_mlp read the same input: one launch over their concatenated weights [7D, D]; GELU from column 3D on,
        # and those columns land behind the attention slot (column shift D). One launch of 84 column tiles instead of 36 + 48 also
        # saves a partially filled round of tiles (M = B * Lj = 72 row tiles: 23.6 rounds instead of 10.1 + 13.5).
        wq, wk = self._P(a + ".norm_q.weight"), self._P(a + ".norm_k.weight")
        cs = getattr(rope, "cs", None)
        fused = False
        packed = self._single_qkv_mlp(p)               # None unless the split falls on a tile boundary (3 D % 256 == 0)
        if packed is not None:
            w7, b7 = packed
            # q / k RMSNorm + RoPE ride in the same launch's epilogue when the shapes allow (whole 256^2 tiles)
            # live LoRA adapters of the four projections: ONE T = n A_cat^T for all of them, then the K-segment of this launch (engine._lora_operands)
            lt, lb = self._lora_operands([a + ".to_q", a + ".to_k", a + ".to_v", p + ".proj_mlp"], n, B * Lj, "j")
            fused = lt is None and cs is not None and ops.qk_rope_fusable(B * Lj, 7 * D, 2 * D, dh, n.dtype)
            ops.gemm(n, w7, b7, sb, M=B * Lj, ldc=8 * D, epilogue=L.EPI_BIAS_GELU, gelu_from_n=3 * D, c_shift_from_n=3 * D, c_shift=D,
                     qk_rope=ops.QkRope(wq, wk, cs, Lj, 0, 2 * D, dh=dh) if fused else None, lora_t=lt, lora_b=lb)
        else:
            w_qkv, b_qkv = self._attn_qkv(a)
            lt, lb = self._lora_operands([a + ".to_q", a + ".to_k", a + ".to_v"], n, B * Lj, "j")
            ops.gemm(n, w_qkv, b_qkv, sb, M=B * Lj, ldc=8 * D, lora_t=lt, lora_b=lb)
            lt, lb = self._lora_operands([p + ".proj_mlp"], n, B * Lj, "jm")
            ops.gemm(n, self._P(p + ".proj_mlp.weight"), self._P(p + ".proj_mlp.bias"), sb[0, 4 * D:], M=B * Lj, epilogue=L.EPI_BIAS_GELU, ldc=8 * D,
                     lora_t=lt, lora_b=lb)
        if not fused:
            ops.qk_rmsnorm_rope(sb, batches=B, rows_per_batch=Lj, ld=8 * D, q_off=0, k_off=D, heads=H, dh=dh, wq_b=wq, wk_b=wk, split=0,
                                cos=cos, sin=sin)
        st = (8 * D, Lj * 8 * D)
        ops.flash_attn(sb, sb[0, D:], sb[0, 2 * D:], sb[0, 3 * D:], batches=B, heads=H, dh=dh, Lq=Lj, Lkv=Lj, q_strides=st, k_strides=st,
                       v_strides=st, o_strides=st)
        lt, lb = self._lora_operands([p + ".proj_out"], sb[0, 3 * D:], B * Lj, "jo", lda=8 * D)
        ops.gemm(sb[0, 3 * D:], self._P(p + ".proj_out.weight"), self._P(p + ".proj_out.bias"), h_out.base, M=B * Lj, epilogue=L.EPI_RES_GATE,
                 lda=8 * D, ldc=h_out.ld, c_map=h_out.map, residual=h_in.base, ldr=h_in.ld, r_map=h_in.map, gate=emb[:, 2 * D:],
                 gate_ld=emb.stride(0), rows_per_sample=Lj, lora_t=lt, lora_b=lb)

    # ------------------------------------------------------------------ CoMoE -----------------------------------------
    def _comoe(self, B: int, N: int, T: int, x: torch.Tensor, cond_tokens: torch.Tensor, ctrl_enc: torch.Tensor, control_temb: torch.Tensor,
               condition_temb: torch.Tensor, pooled: torch.Tensor, cond_pooled: torch.Tensor, img_ids, txt_ids, cond_ids,
               uniform: Optional[torch.Tensor], z0: torch.Tensor, accumulate: bool):
        """preprocess_moe_forward + moe_forward for one condition (src/UniGenTransformer.py:969-1068): returns (l_aux, exp_counts)
        and writes / accumulates expert_hidden + expert_condition into z0 [B*N, D]."""
        ctl, D, dev = self._ctl, self.inner_dim, self.device
        E, S = ctl.expert_nums, B * N
        # condition tokens -> D   (control_x_embedder, :1040)
        c = self._w("moe_c", (S, D))
        ops.gemm(cond_tokens.reshape(S, -1), self._P("control_x_embedder.weight"), self._P("control_x_embedder.bias"), c, M=S)
        # gate on (x + c): top-1 with Random Token Selection or top-2 (control_params.top_num); `uniform` is the gate's random draw
        rt = self._route(x, c, E, ctl.top_k, uniform)
        C, tos, l_aux, exp_counts = rt.C, rt.tos, rt.l_aux, rt.exp_counts
        # expert modulation: s = Linear(768 -> D)(pooled) per (expert, sample): all E experts' linears as one launch over their stacked
        # weights [E*D, 768] -> [B, E*D] (row of (expert e, sample b) at e*D + b*E*D)
        pe = "moe.moe_layer.experts.deepspeed_experts."
        wmc = self._pack("moe.modc.w", [f"{pe}{e}.0.1.weight" for e in range(E)])
        bmc = self._pack("moe.modc.b", [f"{pe}{e}.0.1.bias" for e in range(E)])
        wmh = self._pack("moe.modh.w", [f"{pe}{e}.1.1.weight" for e in range(E)])
        bmh = self._pack("moe.modh.b", [f"{pe}{e}.1.1.bias" for e in range(E)])
        mod_c = ops.small_linear(cond_pooled, wmc, bmc, self._w("moe_modc", (B, E * D)))
        mod_h = ops.small_linear(pooled, wmh, bmh, self._w("moe_modh", (B, E * D)))
        w_c = self._pack_stack("moe.wc", [f"{pe}{e}.0.0.weight" for e in range(E)])
        b_c = self._pack_stack("moe.bc", [f"{pe}{e}.0.0.bias" for e in range(E)])
        w_h = self._pack_stack("moe.wh", [f"{pe}{e}.1.0.weight" for e in range(E)])
        b_h = self._pack_stack("moe.bh", [f"{pe}{e}.1.0.bias" for e in range(E)])
        xd, yc, yh = self._w("moe_xd", (E, C, D)), self._w("moe_yc", (E, C, D)), self._w("moe_yh", (E, C, D))
        # c' = W_c (s_c * c) + b_c ;  h' = W_h (s_h * (h + c')) + b_h      (expert_forward :957-959)
        mk = dict(E=E, capacity=C, tokens_per_sample=N, mod_estride=D, mod_bstride=E * D)
        ops.moe_dispatch_modulate(c, None, mod_c, tos, xd, **mk)
        ops.gemm(xd, w_c, b_c, yc, M=C, groups=E, a_gstride=C * D, w_gstride=D * D, bias_gstride=D, c_gstride=C * D)
        ops.moe_dispatch_modulate(x, yc, mod_h, tos, xd, **mk)
        ops.gemm(xd, w_h, b_h, yh, M=C, groups=E, a_gstride=C * D, w_gstride=D * D, bias_gstride=D, c_gstride=C * D)
        consis = None
        # Reference quirk, pinned by tests/test_ref_wiring_cpu.py[flux_consis_rope_no_single]: moe_forward adds the consistency terms to locals and
        # rebuilds its return value from them only inside `if self.use_shared_expert` (:1024) - without shared experts the module's work is discarded.
        if ctl.use_consis_module and ctl.use_shared_expert:
            # "V2" consistency module (src/UniGenTransformer.py:984-1004; off in every shipped configuration, round 3): needs the experts' outputs in
            # TOKEN order and separately - ug_moe_combine with the other operand zero gives exactly bf16(p * y). consis_module[0] runs twice
            # (consis_module[1] only holds parameters), both times keeping the sample stream only (context = K / V).
            zero = self._w("moe_zero", (E, C, D)); zero.zero_()
            ehc = self._w("moe_ehc", (B, 2 * N, D))            # per sample [expert hidden | consistency condition states]
            ehc2 = ehc.view(B * 2 * N, D)
            ect = self._w("moe_ect", (S, D))                   # expert condition states, token order
            self._combine(rt, zero, yc, ect)
            for b in range(B):                                 # token rows of one sample are contiguous in ehc[b, :N]
                self._combine(rt, yh, zero, ehc[b, :N], rows=slice(b * N, (b + 1) * N))
            round_to = img_ids.dtype if ctl.use_rope else None
            rope_a = self._rope([cond_ids, cond_ids], round_to) if ctl.use_rope else None
            self._double_block("consis_module.0", B, _Stream(ect, N), _Stream(ehc2[N:], N, 2 * N), _Stream(c, N), None, condition_temb, rope_a, "cs0")
            rope_b = self._rope([img_ids, img_ids, cond_ids], round_to) if ctl.use_rope else None
            ycons = self._w("moe_consis", (B, 2 * N, D))
            self._double_block("consis_module.0", B, _Stream(ehc2, 2 * N), _Stream(ycons.view(B * 2 * N, D), 2 * N), _Stream(x, N), None, control_temb, rope_b, "cs1")
            eh2, ec2 = self._w("moe_eh2", (B, N, D)), self._w("moe_ec2", (B, N, D))
            for b in range(B):
                ops.add(ehc[b, :N], ycons[b, :N], eh2[b])                       # expert_hidden_states + consis_expert_hidden
                ops.add(ect.view(B, N, D)[b], ycons[b, N:], ec2[b])             # expert_condition_states + consis_expert_condition_hidden
            consis = (eh2, ec2)
        xs = cs = None
        s_map = ops.IDENT
        if ctl.use_shared_expert:
            xc = self._w("moe_xc", (B, 2 * N, D))
            xc2 = xc.view(B * 2 * N, D)
            round_to = img_ids.dtype if ctl.use_rope else None
            # shared_expert[0]: sample = image tokens, context = condition tokens, temb = condition_temb   (:1013-1015)
            rope0 = self._rope([cond_ids, img_ids], round_to) if ctl.use_rope else None
            self._double_block("shared_expert.0", B, _Stream(x, N), _Stream(xc2, N, 2 * N), _Stream(c, N), _Stream(xc2[N:], N, 2 * N),
                               condition_temb, rope0, "se0")
            # shared_expert[1]: sample = [image | condition], context = control text (K/V only), temb = control_temb   (:1017-1022)
            rope1 = self._rope([txt_ids, img_ids, cond_ids], round_to) if ctl.use_rope else None
            xcs = _Stream(xc2, 2 * N)
            self._double_block("shared_expert.1", B, xcs, xcs, _Stream(ctrl_enc, T), None, control_temb, rope1, "se1")
            xs, cs, s_map = xc2, xc2[N:], RowMap(N, 2 * N)      # token (b, n): image half at row b*2N + n, condition half N rows further
        if consis is not None:
            # (x_s + expert_h') + (cond_s + expert_c'), each sum a bf16 tensor (:1024, 1089): token-order operands, per sample
            eh2, ec2 = consis
            t1, t2 = self._w("moe_t1", (N, D)), self._w("moe_t2", (N, D))
            z3 = z0.view(B, N, D)
            for b in range(B):
                if xs is not None:
                    xsb, csb = xs.view(B, 2 * N, D)[b, :N], xs.view(B, 2 * N, D)[b, N:]
                    ops.add(xsb, eh2[b], t1); ops.add(csb, ec2[b], t2)
                    a_, b_ = t1, t2
                else:
                    a_, b_ = eh2[b], ec2[b]
                if accumulate:
                    ops.add(a_, b_, t1); ops.add(z3[b], t1, z3[b])
                else:
                    ops.add(a_, b_, z3[b])
            return l_aux, exp_counts
        # combine + CoMoE residual sums for all B samples in one launch (:1024, 1089)
        self._combine(rt, yh, yc, z0, xs=xs, cs=cs, s_map=s_map, accumulate=accumulate)
        return l_aux, exp_counts

    # ------------------------------------------------------------------ forward ---------------------------------------
    def forward(self, *args, **kwargs):
        """Inference (no autograd: the in-place HIP engine below) or, when autograd is on and some parameter requires a gradient - the reference's
        train.py:622-662 after `init_trainable_param()` - the differentiable forward of unigen_amd/training.py. Same arguments, same 3-tuple."""
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            from . import training
            self._refuse_lora_in_training()
            return training.flux_forward(self, *args, **kwargs)
        with torch.no_grad(), self._lora_scaled(kwargs.get("joint_attention_kwargs")):
            return self._forward_inference(*args, **kwargs)

    def _forward_inference(self, hidden_states: torch.Tensor, condition_hidden_states=None, conditioning_scale: float = 1.0,
                encoder_hidden_states: torch.Tensor = None, pooled_projections: torch.Tensor = None, condition_pooled_projections=None,
                timestep: torch.Tensor = None, img_ids: torch.Tensor = None, txt_ids: torch.Tensor = None, guidance: torch.Tensor = None,
                condition_ids=None, joint_attention_kwargs: Optional[Dict[str, Any]] = None, skip_layers=None, gate_uniform=None, **kwargs):
        """src/UniGenTransformer.py:1182-1271 (lists of conditions: :1360-1450). Returns (pred [B, N, C_in], {'moe_loss'}, {'expert_counts'}).
        `gate_uniform` (extra): the Uniform(0,1) [S, E] draw of deepspeed's Random Token Selection, for reproducible parity runs."""
        if self._ctl is None:
            raise RuntimeError("call init_condition_block(...) before forward")
        if joint_attention_kwargs and any(k != "scale" for k in joint_attention_kwargs):
            raise ValueError("joint_attention_kwargs other than 'scale' (e.g. IP-adapter embeds) are not supported")
        ctl, cfg, D, dev = self._ctl, self.config, self.inner_dim, self.device
        multi = isinstance(condition_hidden_states, (list, tuple))
        if multi != self.multi_condition:
            raise ValueError(f"{type(self).__name__} expects {'lists of' if self.multi_condition else 'single'} condition tensors")
        dt = self._check_dtype(hidden_states, encoder_hidden_states)       # bf16, or fp32 = verification mode
        self._emb_tab.clear()
        if txt_ids.ndim == 3:
            txt_ids = txt_ids[0]
        if img_ids.ndim == 3:
            img_ids = img_ids[0]
        B, N, _ = hidden_states.shape
        T = encoder_hidden_states.shape[1]
        Lj = T + N
        # `timestep.to(hidden_states.dtype) * 1000` (:1217-1220) - bf16 scalar path, then fp32 for the sinusoid
        t_f32 = (timestep.to(BF) * 1000).float().contiguous()
        g_f32 = (guidance.to(BF) * 1000).float().contiguous() if guidance is not None else None
        if cfg.guidance_embeds and g_f32 is None:
            raise ValueError("guidance is required when config.guidance_embeds is True")
        if not cfg.guidance_embeds:
            g_f32 = None
        pooled = pooled_projections.to(dt).contiguous()

        x = self._w("x", (B * N, D))
        ops.gemm(hidden_states.to(dt).reshape(B * N, -1), self._P("x_embedder.weight"), self._P("x_embedder.bias"), x, M=B * N)
        temb = self._time_text_embed("time_text_embed", t_f32, pooled, g_f32, "base")
        enc = self._w("enc", (B * T, D))
        ops.gemm(encoder_hidden_states.to(dt).reshape(B * T, -1), self._P("context_embedder.weight"), self._P("context_embedder.bias"), enc, M=B * T)
        n_d, n_cj, n_s, n_cs = cfg.num_layers, ctl.cn_joint_layers, cfg.num_single_layers, ctl.cn_single_layers
        # every AdaLN linear driven by temb, in one launch (engine._adaln_group)
        self._adaln_group("base", temb, [f"transformer_blocks.{i}.{n}" for i in range(n_d) for n in ("norm1", "norm1_context")]
                          + [f"single_transformer_blocks.{j}.norm" for j in range(n_s)] + ["norm_out"])
        rope_base = self._rope([txt_ids, img_ids], None)                         # base path: fp32 tables (:1238-1239)
        rope_ctl = self._rope([txt_ids, img_ids], img_ids.dtype) if ctl.use_rope else None

        xs, es = _Stream(x, N), _Stream(enc, T)
        z = self._w("z", (B * N, D))
        zs = _Stream(z, N)
        moe = None
        last_m = -1
        for i in range(n_d):
            self._double_block(f"transformer_blocks.{i}", B, xs, xs, es, es, temb, rope_base, "base")
            m = int(i / (n_d / n_cj))                                             # (:1126-1127)
            if moe is None:
                # preprocess_moe_forward, once per step, on the text stream AFTER base block 0 (:1137 -> :1051)
                control_pooled = pooled if ctl.use_pooled_prompt_embeds else torch.zeros_like(pooled)
                control_temb = self._time_text_embed("control_time_text_embed", t_f32, control_pooled, g_f32, "ctl")
                if ctl.use_shared_expert:
                    self._adaln_group("ctl", control_temb, ["shared_expert.1.norm1", "shared_expert.1.norm1_context"])
                ctrl_enc = self._w("ctrl_enc", (B * T, D))
                ops.gemm(enc, self._P("control_context_embedder.weight"), self._P("control_context_embedder.bias"), ctrl_enc, M=B * T)
                if multi:
                    conds = list(zip(condition_hidden_states, condition_pooled_projections, condition_ids))
                else:
                    conds = [(condition_hidden_states, condition_pooled_projections, condition_ids)]
                unis = gate_uniform if isinstance(gate_uniform, (list, tuple)) else [gate_uniform] * len(conds)
                z0 = self._w("z0", (B * N, D))
                cond_temb_sum = None
                for k, ((ct, cp, cid), uni) in enumerate(zip(conds, unis)):
                    if cp.ndim == 1:
                        cp = cp.unsqueeze(0)
                    if ct.ndim == 2:
                        ct = ct.unsqueeze(0)
                    cp = cp.to(dt).contiguous()
                    cid = cid[0] if cid.ndim == 3 else cid
                    cond_temb = self._time_text_embed("control_condition_embed", t_f32, cp, g_f32, f"cond{k}")
                    if ctl.use_shared_expert:       # shared_expert[0] is driven by THIS condition's temb (:1015), the control blocks by the sum
                        self._adaln_group("se0", cond_temb, ["shared_expert.0.norm1", "shared_expert.0.norm1_context"])
                    l_aux, exp_counts = self._comoe(B, N, T, x, ct.to(dt), ctrl_enc, control_temb, cond_temb, pooled, cp, img_ids, txt_ids, cid,
                                                    uni, z0, accumulate=k > 0)
                    if cond_temb_sum is None:
                        cond_temb_sum = cond_temb
                    else:                                                         # sum(merge_condition_temb) (:1319)
                        cond_temb_sum = ops.add(cond_temb_sum, cond_temb, self._w("cond_temb_sum", cond_temb.shape))
                moe = dict(ctrl_enc=_Stream(ctrl_enc, T), condition_temb=cond_temb_sum, l_aux=l_aux, exp_counts=exp_counts)
                # every control block's AdaLN linear reads the (summed) condition temb (:1092, 1319): one launch
                self._adaln_group("cond", cond_temb_sum, [f"control_joint_trans_blocks.{m_}.{n}" for m_ in range(n_cj) for n in ("norm1", "norm1_context")]
                                  + ([f"control_single_trans_blocks.{m_}.norm" for m_ in range(n_cs)] if ctl.use_single_trans_blocks else []))
                z_in = _Stream(z0, N)
            else:
                z_in = xs                                                         # control blocks read the BASE stream (:1085-1097)
            self._double_block(f"control_joint_trans_blocks.{m}", B, z_in, zs, moe["ctrl_enc"], None, moe["condition_temb"], rope_ctl, "ctl",
                               ctx_cached=(m == last_m))
            last_m = m
            # x = x + controlnet_add_joint_blocks[m](z) * conditioning_scale   (:1104,1141)
            ops.gemm(z, self._P(f"controlnet_add_joint_blocks.{m}.weight"), self._P(f"controlnet_add_joint_blocks.{m}.bias"), x, M=B * N,
                     epilogue=L.EPI_RES_SCALE, residual=x, alpha=float(conditioning_scale))

        # hidden_states = cat([encoder_hidden_states, hidden_states], dim=1)   (:1146)
        h = self._w("h", (B, Lj, D))
        h[:, :T].copy_(enc.view(B, T, D))
        h[:, T:].copy_(x.view(B, N, D))
        h2 = h.view(B * Lj, D)
        hs = _Stream(h2, Lj)
        zj = self._w("zj", (B * Lj, D))
        zjs = _Stream(zj, Lj)
        for j in range(n_s):
            self._single_block(f"single_transformer_blocks.{j}", B, hs, hs, temb, rope_base)
            if ctl.use_single_trans_blocks:
                m = int(j / (n_s / n_cs))                                         # (:1159-1160)
                self._single_block(f"control_single_trans_blocks.{m}", B, hs, zjs, moe["condition_temb"], rope_ctl)
                wz, bz = self._P(f"controlnet_add_single_blocks.{m}.weight"), self._P(f"controlnet_add_single_blocks.{m}.bias")
                if ctl.single_block_control_method == "overall_add":              # (:1166-1167)
                    ops.gemm(zj, wz, bz, h2, M=B * Lj, epilogue=L.EPI_RES_SCALE, residual=h2, alpha=float(conditioning_scale))
                else:                                                             # image tokens only (:1168-1172)
                    mp = RowMap(N, Lj)
                    ops.gemm(zj[T:], wz, bz, h2[T:], M=B * N, epilogue=L.EPI_RES_SCALE, lda=D, a_map=mp, ldc=D, c_map=mp, residual=h2[T:],
                             ldr=D, r_map=mp, alpha=float(conditioning_scale))
        # norm_out (AdaLayerNormContinuous: scale first) + proj_out on the image tokens   (:1174, 1264-1265)
        emb_o = self._adaln_emb("norm_out", temb, 2, "o")
        img = _Stream(h2[T:], N, Lj)
        no = self._w("norm_s", (B * N, D))
        ops.adaln_modulate(img.base, emb_o[:, D:], emb_o, no, rows=B * N, D=D, rows_per_sample=N, mod_ld=emb_o.stride(0), ldx=D, x_map=img.map)
        out = torch.empty(B, N, self.out_channels, device=dev, dtype=dt)
        ops.gemm(no, self._P("proj_out.weight"), self._P("proj_out.bias"), out.view(B * N, -1), M=B * N)
        return out, dict(moe_loss=moe["l_aux"][0] * 0.1), dict(expert_counts=moe["exp_counts"])


class MultiCondtionUniGenFlux(UniGenFlux):
    """Drop-in for the reference `MultiCondtionUniGenFlux` (sic): lists of per-condition tensors; per-condition CoMoE outputs and
    condition tembs are summed (src/UniGenTransformer.py:1275-1357)."""

    multi_condition = True
