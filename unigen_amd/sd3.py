"""Host side of the UniGenSD3 forward (SD3.5-medium backbone) on MI355X.

Mirrors the reference `UniGenSD3(UniGenBase(SD3Transformer2DModel))` (src/UniGenTransformer.py:21-710): 24 base
JointTransformerBlocks (dual attention in layers 0-12, last block context_pre_only), one control JointTransformerBlock + zero-res
Linear per base block, CoMoE once per step whose experts are either the modulated linears (use_modulate / use_rope) or - the
shipped config/unigen.yaml - two SD3SingleTransformerBlocks per expert that are fed PER-TOKEN time embeddings and self-attend over
the expert's capacity slots (empty slots included, src/UniGenUtils.py:386-414), plus the two shared joint blocks. Latents are NCHW
in and out; the timestep is used unscaled. All arithmetic goes through the C ABI (unigen_amd/ops.py).
"""
from __future__ import annotations

import json
import math
import os
from types import SimpleNamespace
from typing import Any, Dict, List, Optional, Tuple

import torch

from . import lib as L
from . import ops
from .engine import BF, HipModule, _register, _Stream, reference_init_
from .flux import _lin, _time_text_embed_shapes
from .ops import RowMap

SD35_MEDIUM_CONFIG = dict(sample_size=128, patch_size=2, in_channels=16, out_channels=16, num_layers=24, attention_head_dim=64,
                          num_attention_heads=24, joint_attention_dim=4096, caption_projection_dim=1536, pooled_projection_dim=2048,
                          pos_embed_max_size=384, dual_attention_layers=tuple(range(13)), qk_norm="rms_norm")


def _attn_shapes(s, p, D, dh, qk_norm, ctx, context_pre_only=False):
    for n in ("to_q", "to_k", "to_v", "to_out.0"):
        _lin(s, f"{p}.{n}", D, D)
    if qk_norm:
        s[f"{p}.norm_q.weight"] = (dh,); s[f"{p}.norm_k.weight"] = (dh,)
    if ctx:
        for n in ("add_q_proj", "add_k_proj", "add_v_proj"):
            _lin(s, f"{p}.{n}", D, D)
        if not context_pre_only:
            _lin(s, f"{p}.to_add_out", D, D)
        if qk_norm:
            s[f"{p}.norm_added_q.weight"] = (dh,); s[f"{p}.norm_added_k.weight"] = (dh,)


def _ff_shapes(s, p, D):
    _lin(s, f"{p}.net.0.proj", 4 * D, D)
    _lin(s, f"{p}.net.2", D, 4 * D)


def _joint_shapes(s, p, D, dh, qk_norm, context_pre_only, dual):
    _lin(s, f"{p}.norm1.linear", (9 if dual else 6) * D, D)
    _lin(s, f"{p}.norm1_context.linear", (2 if context_pre_only else 6) * D, D)
    _attn_shapes(s, p + ".attn", D, dh, qk_norm, True, context_pre_only)
    if dual:
        _attn_shapes(s, p + ".attn2", D, dh, qk_norm, False)
    _ff_shapes(s, p + ".ff", D)
    if not context_pre_only:
        _ff_shapes(s, p + ".ff_context", D)


def sd3_base_param_shapes(cfg) -> Dict[str, Tuple[int, ...]]:
    D, dh, p = cfg.num_attention_heads * cfg.attention_head_dim, cfg.attention_head_dim, cfg.patch_size
    s: Dict[str, Tuple[int, ...]] = {}
    s["pos_embed.proj.weight"] = (D, cfg.in_channels, p, p); s["pos_embed.proj.bias"] = (D,)
    _time_text_embed_shapes(s, "time_text_embed", D, cfg.pooled_projection_dim, False)
    _lin(s, "context_embedder", cfg.caption_projection_dim, cfg.joint_attention_dim)
    for i in range(cfg.num_layers):
        _joint_shapes(s, f"transformer_blocks.{i}", D, dh, cfg.qk_norm, i == cfg.num_layers - 1, i in cfg.dual_attention_layers)
    _lin(s, "norm_out.linear", 2 * D, D)
    _lin(s, "proj_out", p * p * cfg.out_channels, D)
    return s


def sd3_control_param_shapes(cfg, ctl) -> Dict[str, Tuple[int, ...]]:
    D, dh, p = cfg.num_attention_heads * cfg.attention_head_dim, cfg.attention_head_dim, cfg.patch_size
    s: Dict[str, Tuple[int, ...]] = {}
    s["control_pos_embed_input.proj.weight"] = (D, cfg.in_channels, p, p); s["control_pos_embed_input.proj.bias"] = (D,)
    _time_text_embed_shapes(s, "control_time_text_embed", D, cfg.pooled_projection_dim, False)
    _time_text_embed_shapes(s, "control_condition_embed", D, cfg.pooled_projection_dim, False)
    _lin(s, "control_context_embedder", D, D)                                     # UniGenSD3 override (:493)
    for i in range(cfg.num_layers):
        _joint_shapes(s, f"control_transformer_blocks.{i}", D, dh, cfg.qk_norm, False, i in cfg.dual_attention_layers)
        _lin(s, f"controlnet_add_blocks.{i}", D, D)
    s["moe.moe_layer.gate.wg.weight"] = (ctl.expert_nums, D)
    for e in range(ctl.expert_nums):
        pe = f"moe.moe_layer.experts.deepspeed_experts.{e}"
        for k in (0, 1):
            if ctl.modulated:
                _lin(s, f"{pe}.{k}.0", D, D)
                _lin(s, f"{pe}.{k}.1", D, cfg.pooled_projection_dim)
            else:                                                                  # SD3SingleTransformerBlock (no qk-norm)
                _lin(s, f"{pe}.{k}.norm1.linear", 6 * D, D)
                _attn_shapes(s, f"{pe}.{k}.attn", D, dh, None, False)
                _ff_shapes(s, f"{pe}.{k}.ff", D)
    if ctl.use_shared_expert:
        _joint_shapes(s, "shared_expert.0", D, dh, cfg.qk_norm, False, False)
        _joint_shapes(s, "shared_expert.1", D, dh, cfg.qk_norm, True, True)
    return s


def sincos_pos_embed_2d(embed_dim: int, grid_size: int, base_size: int) -> torch.Tensor:
    """diffusers get_2d_sincos_pos_embed (interpolation_scale 1) -> [grid_size^2, embed_dim] fp32: PatchEmbed's `pos_embed` buffer."""
    g = torch.arange(grid_size, dtype=torch.float32) / (grid_size / base_size)
    gw, gh = torch.meshgrid(g, g, indexing="xy")

    def one_d(dim, pos):
        omega = 1.0 / 10000 ** (torch.arange(dim // 2, dtype=torch.float64) / (dim / 2.0))
        out = torch.outer(pos.reshape(-1).double(), omega)
        return torch.cat([torch.sin(out), torch.cos(out)], dim=1)

    return torch.cat([one_d(embed_dim // 2, gw), one_d(embed_dim // 2, gh)], dim=1).float()


class UniGenSD3(HipModule):
    """Drop-in for the reference `UniGenSD3`."""

    def __init__(self, config: Optional[dict] = None, device=None, dtype=BF, _init: bool = True, **kwargs):
        super().__init__()
        c = dict(SD35_MEDIUM_CONFIG)
        c.update(config or {})
        c.update(kwargs)
        c["dual_attention_layers"] = tuple(c["dual_attention_layers"])
        self.config = SimpleNamespace(**c)
        self.inner_dim = self.config.num_attention_heads * self.config.attention_head_dim
        if self.config.caption_projection_dim != self.inner_dim:
            raise ValueError("caption_projection_dim must equal the inner dim")
        self.out_channels = self.config.out_channels
        self._heads, self._head_dim = self.config.num_attention_heads, self.config.attention_head_dim
        if self._head_dim not in (64, 128):
            raise ValueError("attention_head_dim must be 64 or 128")
        self._ctl = None
        self._pos_crop: Dict[Tuple, torch.Tensor] = {}
        for name, shape in sd3_base_param_shapes(self.config).items():
            _register(self, name, shape, device, dtype)
        if _init:      # from_config = a fresh model: torch's default initial values (from_pretrained overwrites, so it skips this)
            reference_init_(self, list(sd3_base_param_shapes(self.config)), copies={}, zero_prefixes=(), expert_prefix="\0")
        mx = self.config.pos_embed_max_size
        base = self.config.sample_size // self.config.patch_size
        table = sincos_pos_embed_2d(self.inner_dim, mx, base).unsqueeze(0)
        self.pos_embed.register_buffer("pos_embed", table.to(device) if device is not None else table, persistent=True)

    @classmethod
    def from_config(cls, config: dict, **kw) -> "UniGenSD3":
        return cls(config, **kw)

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path, subfolder: Optional[str] = None, revision=None, variant=None, torch_dtype=BF,
                        device=None, **kw) -> "UniGenSD3":
        path = os.fspath(pretrained_model_name_or_path)
        if subfolder:
            path = os.path.join(path, subfolder)
        if not os.path.isdir(path):
            raise OSError(f"{path} is not a local directory (unigen_amd loads checkpoints from disk only)")
        with open(os.path.join(path, "config.json")) as f:
            raw = json.load(f)
        model = cls({k: raw[k] for k in SD35_MEDIUM_CONFIG if k in raw}, device=device, dtype=torch_dtype, _init=False)
        from safetensors.torch import load_file
        sd = {}
        for fn in sorted(f for f in os.listdir(path) if f.endswith(".safetensors")):
            sd.update(load_file(os.path.join(path, fn)))
        if not sd:
            raise OSError(f"no *.safetensors weights under {path}")
        res = model.load_state_dict(sd, strict=False)
        if res.missing_keys:
            raise RuntimeError(f"checkpoint misses base parameters: {res.missing_keys[:8]} ...")
        return model

    def _probe(self):
        return self.pos_embed.proj.weight

    def init_condition_block(self, condition_nums: int = 1, condition_types: Optional[List[str]] = None, **kwargs):
        """UniGenBase.init_condition_block -> init_control_block -> init_moe_block (src/UniGenTransformer.py:22-223, 491-496)."""
        self.condition_nums = condition_nums
        self.condition_types = condition_types
        control_params = kwargs.get("control_params", None)
        assert control_params is not None, ValueError("Please provice control net model parameter")
        get = control_params.get
        if get("use_rope", False):
            raise ValueError("UniGenSD3 with use_rope needs the reference's undefined prepare_latent_image_ids (SURVEY Q1); not supported")
        assert get("use_encoder_hidden_states", True), ValueError("please use joint transformer block to enhance condition hidden states")
        if get("cn2base_method", "add") != "add" or get("use_pos_embed", False) or get("extra_conditioning_channels", 0):
            raise ValueError("only the shipped control configuration is supported (cn2base_method='add', no extra pos-embed / channels)")
        if int(get("num_layers", self.config.num_layers)) != self.config.num_layers:
            raise ValueError("control num_layers must equal the base depth (interval 1)")
        expert_num = get("expert_num", None)
        ctl = SimpleNamespace(
            modulated=bool(get("use_modulate", False)),
            use_pooled_prompt_embeds=bool(get("use_pooled_prompt_embeds", True)),
            use_shared_expert=bool(get("use_shared_expert", False)),
            expert_nums=int(expert_num) if expert_num is not None else (condition_nums + 1) * int(get("expert_num_each_condition", 3)),
            top_k=int(get("top_num", 1)),
        )
        if ctl.top_k < 1 or ctl.expert_nums > 16 or ctl.expert_nums < ctl.top_k:
            raise ValueError("top_num 1 .. expert count (deepspeed top1gating / top2gating / topkgating) with at most 16 experts is implemented")
        self._ctl = ctl
        dev_, dt_ = self.device, self.dtype
        shapes = sd3_control_param_shapes(self.config, ctl)
        for name, shape in shapes.items():
            _register(self, name, shape, dev_, dt_)
        # UniGenBase.init_control_block builds every control module fresh (no deepcopy of the base), zero only for controlnet_add_blocks (:119-123)
        reference_init_(self, list(shapes), copies={}, zero_prefixes=("controlnet_add_blocks.",), expert_prefix="moe.moe_layer.experts.deepspeed_experts.")
        self.control_pos_embed_input.register_buffer("pos_embed", self.pos_embed.pos_embed.clone(), persistent=True)
        names = ["control_pos_embed_input", "control_time_text_embed", "control_condition_embed", "control_context_embedder",
                 "control_transformer_blocks", "controlnet_add_blocks", "moe"] + (["shared_expert"] if ctl.use_shared_expert else [])
        self.trainable_control_modules = {n: getattr(self, n) for n in names}
        if get("use_transformer_params", False):
            self.init_control_param()
        self._packed.clear()
        self._pname_cache = None

    def init_control_param(self):
        """src/UniGenTransformer.py:145-158."""
        sd = self.state_dict()
        with torch.no_grad():
            for k, v in sd.items():
                if k.startswith("pos_embed."):
                    sd["control_pos_embed_input." + k[len("pos_embed."):]].copy_(v)
                if k.startswith("time_text_embed."):
                    sd["control_time_text_embed." + k[len("time_text_embed."):]].copy_(v)
                    sd["control_condition_embed." + k[len("time_text_embed."):]].copy_(v)
                if k.startswith("transformer_blocks."):
                    tgt = "control_transformer_blocks." + k[len("transformer_blocks."):]
                    if tgt in sd and sd[tgt].shape == v.shape:
                        sd[tgt].copy_(v)

    # ------------------------------------------------------------------ pieces ----------------------------------------
    def _patch_embed(self, prefix: str, latents: torch.Tensor, out_name: str) -> torch.Tensor:
        """PatchEmbed: Conv2d(k = s = patch) as a GEMM over unfolded patches, + the centre-cropped sincos table (fp32 add)."""
        cfg, D = self.config, self.inner_dim
        B, C, Hh, Ww = latents.shape
        p = cfg.patch_size
        h, w = Hh // p, Ww // p
        patches = latents.to(self.dtype).view(B, C, h, p, w, p).permute(0, 2, 4, 1, 3, 5).reshape(B * h * w, C * p * p).contiguous()   # layout only
        K = C * p * p
        wt = self._P(prefix + ".proj.weight").view(D, K)
        x = self._w(out_name, (B * h * w, D))
        ops.gemm(patches, wt, self._P(prefix + ".proj.bias"), x, M=B * h * w)
        key = (prefix, h, w)
        tab = self._pos_crop.get(key)
        buf = getattr(self, prefix).pos_embed
        if tab is None or tab.device != x.device:
            mx = cfg.pos_embed_max_size
            top, left = (mx - h) // 2, (mx - w) // 2
            tab = buf.reshape(mx, mx, D)[top:top + h, left:left + w].reshape(h * w, D).float().contiguous().to(x.device)
            self._pos_crop[key] = tab
        ops.add_rowbcast_f32(x, tab, h * w)
        return x

    def _experts_modulated(self, B, N, x, c, pooled, cond_pooled, tos, C):
        """expert_forward with modulated linears (:252-259): c' = W_c (s_c * c) + b_c ; h' = W_h (s_h * (h + c')) + b_h."""
        E, D = self._ctl.expert_nums, self.inner_dim
        pe = "moe.moe_layer.experts.deepspeed_experts."
        wmc = self._pack("moe.modc.w", [f"{pe}{e}.0.1.weight" for e in range(E)])
        bmc = self._pack("moe.modc.b", [f"{pe}{e}.0.1.bias" for e in range(E)])
        wmh = self._pack("moe.modh.w", [f"{pe}{e}.1.1.weight" for e in range(E)])
        bmh = self._pack("moe.modh.b", [f"{pe}{e}.1.1.bias" for e in range(E)])
        mod_c = ops.small_linear(cond_pooled, wmc, bmc, self._w("moe_modc", (B, E * D)))      # all experts' Linear(pooled) in one launch
        mod_h = ops.small_linear(pooled, wmh, bmh, self._w("moe_modh", (B, E * D)))
        w_c = self._pack_stack("moe.wc", [f"{pe}{e}.0.0.weight" for e in range(E)])
        b_c = self._pack_stack("moe.bc", [f"{pe}{e}.0.0.bias" for e in range(E)])
        w_h = self._pack_stack("moe.wh", [f"{pe}{e}.1.0.weight" for e in range(E)])
        b_h = self._pack_stack("moe.bh", [f"{pe}{e}.1.0.bias" for e in range(E)])
        xd, yc, yh = self._w("moe_xd", (E, C, D)), self._w("moe_yc", (E, C, D)), self._w("moe_yh", (E, C, D))
        gk = dict(M=C, groups=E, a_gstride=C * D, w_gstride=D * D, bias_gstride=D, c_gstride=C * D)
        mk = dict(E=E, capacity=C, tokens_per_sample=N, mod_estride=D, mod_bstride=E * D)
        ops.moe_dispatch_modulate(c, None, mod_c, tos, xd, **mk)
        ops.gemm(xd, w_c, b_c, yc, **gk)
        ops.moe_dispatch_modulate(x, yc, mod_h, tos, xd, **mk)
        ops.gemm(xd, w_h, b_h, yh, **gk)
        return yh, yc

    def _expert_blocks(self, which: int, B: int, N: int, src: torch.Tensor, temb: torch.Tensor, tos: torch.Tensor, sidx: torch.Tensor, C: int,
                       out_name: str) -> torch.Tensor:
        """expert[which](chunk, temb_chunk) for every expert at once (:261-262): SD3SingleTransformerBlock on the expert's C capacity
        slots (batch 1, sequence C, empty slots are zero rows), AdaLN-Zero driven by the DISPATCHED per-token temb: slot s of expert e
        sees emb_e[sample(token)] (emb_e[zero temb] for empty slots). Grouped GEMMs over experts, attention batched over experts."""
        E, D, H, dh = self._ctl.expert_nums, self.inner_dim, self._heads, self._head_dim
        pe = "moe.moe_layer.experts.deepspeed_experts."
        blk = lambda e: f"{pe}{e}.{which}"
        # per-(expert, sample) AdaLN rows, plus one row for the all-zero temb of empty slots: Linear(silu(0)) = bias
        tz = self._w("exp_tz", (B + 1, D))
        tz[:B].copy_(temb)
        tz[B].zero_()
        wn = self._pack(f"exp{which}.n1w", [blk(e) + ".norm1.linear.weight" for e in range(E)])
        bn = self._pack(f"exp{which}.n1b", [blk(e) + ".norm1.linear.bias" for e in range(E)])
        emb_tab = ops.small_linear(tz, wn, bn, self._w("exp_embtab", (B + 1, E * 6 * D)), silu_in=True)     # all experts' AdaLN linears, one launch
        emb = self._w("exp_emb", (E * C, 6 * D))                                  # per-slot modulation rows
        ops.gather_rows(emb_tab.view((B + 1) * E, 6 * D), sidx, emb)
        xd = self._w(out_name, (E, C, D))
        ops.moe_dispatch_modulate(src, None, None, tos, xd, E=E, capacity=C, tokens_per_sample=N)
        x2 = xd.view(E * C, D)
        n = self._w("exp_norm", (E * C, D))
        ops.adaln_modulate(x2, emb, emb[:, D:], n, rows=E * C, D=D, rows_per_sample=1, mod_ld=6 * D)
        w_qkv = self._pack_stack(f"exp{which}.qkvw", [blk(e) + ".attn.to_q.weight" for e in range(E)] + [blk(e) + ".attn.to_k.weight" for e in range(E)]
                                 + [blk(e) + ".attn.to_v.weight" for e in range(E)]).view(3, E, D, D)
        b_qkv = self._pack_stack(f"exp{which}.qkvb", [blk(e) + ".attn.to_q.bias" for e in range(E)] + [blk(e) + ".attn.to_k.bias" for e in range(E)]
                                 + [blk(e) + ".attn.to_v.bias" for e in range(E)]).view(3, E, D)
        qkv = self._w("exp_qkv", (E, C, 3 * D))
        for j in range(3):                                                          # q, k, v: one grouped GEMM each into column block j
            ops.gemm(n, w_qkv[j], b_qkv[j], qkv.view(E * C, 3 * D)[0, j * D:], M=C, ldc=3 * D, groups=E, a_gstride=C * D, w_gstride=D * D,
                     bias_gstride=D, c_gstride=C * 3 * D)
        att = self._w("exp_att", (E, C, D))
        st = (3 * D, C * 3 * D)
        q2 = qkv.view(E * C, 3 * D)
        ops.flash_attn(q2, q2[0, D:], q2[0, 2 * D:], att, batches=E, heads=H, dh=dh, Lq=C, Lkv=C, q_strides=st, k_strides=st, v_strides=st,
                       o_strides=(D, C * D))
        w_o = self._pack_stack(f"exp{which}.ow", [blk(e) + ".attn.to_out.0.weight" for e in range(E)])
        b_o = self._pack_stack(f"exp{which}.ob", [blk(e) + ".attn.to_out.0.bias" for e in range(E)])
        gk = dict(M=C, groups=E, w_gstride=D * D, bias_gstride=D, rows_per_sample=1, gate_ld=6 * D, gate_gstride=C * 6 * D)
        ops.gemm(att, w_o, b_o, xd, epilogue=L.EPI_RES_GATE, a_gstride=C * D, c_gstride=C * D, residual=xd, r_gstride=C * D, gate=emb[:, 2 * D:], **gk)
        ops.adaln_modulate(x2, emb[:, 3 * D:], emb[:, 4 * D:], n, rows=E * C, D=D, rows_per_sample=1, mod_ld=6 * D)
        w1 = self._pack_stack(f"exp{which}.w1", [blk(e) + ".ff.net.0.proj.weight" for e in range(E)])
        b1 = self._pack_stack(f"exp{which}.b1", [blk(e) + ".ff.net.0.proj.bias" for e in range(E)])
        w2 = self._pack_stack(f"exp{which}.w2", [blk(e) + ".ff.net.2.weight" for e in range(E)])
        b2 = self._pack_stack(f"exp{which}.b2", [blk(e) + ".ff.net.2.bias" for e in range(E)])
        hid = self._w("exp_ffh", (E, C, 4 * D))
        ops.gemm(n, w1, b1, hid, M=C, epilogue=L.EPI_BIAS_GELU, groups=E, a_gstride=C * D, w_gstride=4 * D * D, bias_gstride=4 * D, c_gstride=C * 4 * D)
        gk["w_gstride"] = 4 * D * D
        ops.gemm(hid, w2, b2, xd, epilogue=L.EPI_RES_GATE, a_gstride=C * 4 * D, c_gstride=C * D, residual=xd, r_gstride=C * D, gate=emb[:, 5 * D:], **gk)
        return xd

    def _comoe(self, B, N, T, x, c, ctrl_enc, control_temb, condition_temb, pooled, cond_pooled, uniform, z0):
        ctl, D, dev = self._ctl, self.inner_dim, self.device
        E, S = ctl.expert_nums, B * N
        rt = self._route(x, c, E, ctl.top_k, uniform)          # top-1 + RTS or top-2 (control_params.top_num); `uniform` = the gate's random draw
        C, tos, l_aux, exp_counts = rt.C, rt.tos, rt.l_aux, rt.exp_counts
        if ctl.modulated:
            yh, yc = self._experts_modulated(B, N, x, c, pooled, cond_pooled, tos, C)
        else:
            # row of the [B + 1][E] AdaLN table each slot reads: sample(token) * E + e, with the zero-temb row B for empty slots
            e_ar = torch.arange(E, device=dev, dtype=torch.int32).view(E, 1)
            sidx = (e_ar + E * torch.where(tos >= 0, torch.div(tos, N, rounding_mode="floor"), torch.full_like(tos, B))).to(torch.int32).reshape(-1).contiguous()
            yh = self._expert_blocks(0, B, N, x, control_temb, tos, sidx, C, "moe_yh")          # expert[0](hidden, temb)
            yc = self._expert_blocks(1, B, N, c, condition_temb, tos, sidx, C, "moe_yc")        # expert[1](condition, condition_temb)
        xs = cs = None
        s_map = ops.IDENT
        if ctl.use_shared_expert:
            xc = self._w("moe_xc", (B, 2 * N, D))
            xc2 = xc.view(B * 2 * N, D)
            self._double_block("shared_expert.0", B, _Stream(x, N), _Stream(xc2, N, 2 * N), _Stream(c, N), _Stream(xc2[N:], N, 2 * N),
                               condition_temb, None, "se0")
            xcs = _Stream(xc2, 2 * N)
            self._double_block("shared_expert.1", B, xcs, xcs, _Stream(ctrl_enc, T), None, control_temb, None, "se1", dual=True, ctx_continuous=True)
            xs, cs, s_map = xc2, xc2[N:], ops.RowMap(N, 2 * N)
        self._combine(rt, yh, yc, z0, xs=xs, cs=cs, s_map=s_map)      # all B samples in one launch
        return l_aux, exp_counts

    # ------------------------------------------------------------------ forward ---------------------------------------
    def forward(self, *args, **kwargs):
        """Inference (the in-place HIP engine below, no autograd) or - autograd on and some parameter requiring a gradient, train.py:622-662 after
        `init_trainable_param()` - the differentiable forward of unigen_amd/training.py. Same arguments, same 3-tuple."""
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            from . import training
            self._refuse_lora_in_training()
            return training.sd3_forward(self, *args, **kwargs)
        with torch.no_grad(), self._lora_scaled(kwargs.get("joint_attention_kwargs")):
            return self._forward_inference(*args, **kwargs)

    def _forward_inference(self, hidden_states: torch.Tensor, condition_hidden_states: torch.Tensor = None, conditioning_scale: float = 1.0,
                encoder_hidden_states: torch.Tensor = None, pooled_projections: torch.Tensor = None,
                condition_pooled_projections: torch.Tensor = None, timestep: torch.Tensor = None,
                joint_attention_kwargs: Optional[Dict[str, Any]] = None, skip_layers=None, gate_uniform=None, **kwargs):
        """src/UniGenTransformer.py:625-710. hidden_states / condition_hidden_states: NCHW latents; returns (NCHW, {'moe_loss'}, {'expert_counts'})."""
        if self._ctl is None:
            raise RuntimeError("call init_condition_block(...) before forward")
        if joint_attention_kwargs and any(k != "scale" for k in joint_attention_kwargs):
            raise ValueError("joint_attention_kwargs other than 'scale' are not supported")
        dt = self._check_dtype(hidden_states, encoder_hidden_states)       # bf16, or fp32 = verification mode
        self._emb_tab.clear()
        cfg, ctl, D, dev = self.config, self._ctl, self.inner_dim, self.device
        B, _, height, width = hidden_states.shape
        p = cfg.patch_size
        h, w = height // p, width // p
        N, T, Lyr = h * w, encoder_hidden_states.shape[1], cfg.num_layers
        t_f32 = timestep.to(dev).float().expand(B).contiguous()                    # used as given: SD3 does not rescale the timestep
        pooled = pooled_projections.to(dt).contiguous()
        cpooled = condition_pooled_projections.to(dt).contiguous()

        x = self._patch_embed("pos_embed", hidden_states, "x")
        temb = self._time_text_embed("time_text_embed", t_f32, pooled, None, "base")
        enc = self._w("enc", (B * T, D))
        ops.gemm(encoder_hidden_states.to(dt).reshape(B * T, -1), self._P("context_embedder.weight"), self._P("context_embedder.bias"), enc, M=B * T)
        self._adaln_group("base", temb, [f"transformer_blocks.{i}.{n}" for i in range(Lyr) for n in ("norm1", "norm1_context")] + ["norm_out"])
        xs, es = _Stream(x, N), _Stream(enc, T)
        z = self._w("z", (B * N, D))
        zs = _Stream(z, N)
        moe = None
        for i in range(Lyr):
            last = i == Lyr - 1
            dual = i in cfg.dual_attention_layers
            self._double_block(f"transformer_blocks.{i}", B, xs, xs, es, None if last else es, temb, None, "base", dual=dual, ctx_continuous=last)
            if moe is None:          # preprocess_moe_forward at base_block_idx == 0 (:559-563)
                c = self._patch_embed("control_pos_embed_input", condition_hidden_states, "moe_c")
                control_pooled = pooled if ctl.use_pooled_prompt_embeds else torch.zeros_like(pooled)
                control_temb = self._time_text_embed("control_time_text_embed", t_f32, control_pooled, None, "ctl")
                condition_temb = self._time_text_embed("control_condition_embed", t_f32, cpooled, None, "cond")
                self._adaln_group("cond", condition_temb, [f"control_transformer_blocks.{i_}.{n}" for i_ in range(Lyr) for n in ("norm1", "norm1_context")]
                                  + (["shared_expert.0.norm1", "shared_expert.0.norm1_context"] if ctl.use_shared_expert else []))
                if ctl.use_shared_expert:
                    self._adaln_group("ctl", control_temb, ["shared_expert.1.norm1", "shared_expert.1.norm1_context"])
                ctrl_enc = self._w("ctrl_enc", (B * T, D))
                ops.gemm(enc, self._P("control_context_embedder.weight"), self._P("control_context_embedder.bias"), ctrl_enc, M=B * T)
                z0 = self._w("z0", (B * N, D))
                l_aux, exp_counts = self._comoe(B, N, T, x, c, ctrl_enc, control_temb, condition_temb, pooled, cpooled, gate_uniform, z0)
                moe = dict(ctrl_enc=_Stream(ctrl_enc, T), condition_temb=condition_temb, l_aux=l_aux, exp_counts=exp_counts)
                z_in = _Stream(z0, N)
            else:
                z_in = xs
            self._double_block(f"control_transformer_blocks.{i}", B, z_in, zs, moe["ctrl_enc"], None, moe["condition_temb"], None, "ctl", dual=dual)
            ops.gemm(z, self._P(f"controlnet_add_blocks.{i}.weight"), self._P(f"controlnet_add_blocks.{i}.bias"), x, M=B * N,
                     epilogue=L.EPI_RES_SCALE, residual=x, alpha=float(conditioning_scale))
        emb_o = self._adaln_emb("norm_out", temb, 2, "o")
        no = self._w("norm_s", (B * N, D))
        ops.adaln_modulate(x, emb_o[:, D:], emb_o, no, rows=B * N, D=D, rows_per_sample=N, mod_ld=emb_o.stride(0))
        po = self._w("proj_out", (B * N, p * p * self.out_channels))
        ops.gemm(no, self._P("proj_out.weight"), self._P("proj_out.bias"), po, M=B * N)
        # unpatchify (:693-704): layout only
        out = po.view(B, h, w, p, p, self.out_channels).permute(0, 5, 1, 3, 2, 4).reshape(B, self.out_channels, h * p, w * p).contiguous()
        return out, dict(moe_loss=moe["l_aux"][0] * 0.1), dict(expert_counts=moe["exp_counts"])
