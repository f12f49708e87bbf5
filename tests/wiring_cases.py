"""Wiring cases: the reference's own orchestration methods against the oracle's restatement of them, on STAND-IN modules.

The reference's `base_forward` / `control_forward` / `preprocess_moe_forward` / `moe_forward` (UniGenFlux src/UniGenTransformer.py:969-1180,
MultiCondtionUniGenFlux :1275-1357, UniGenSD3 :498-623, UniGenBase.moe_forward :269-296) are pure orchestration: which block runs on which
stream with which temb / ids, what is added to what. Their blocks, embedders and `self.moe.moe_layer` are attributes of `self` - inputs. Here
those attributes are small deterministic torch callables (float64; every one mixes ALL of its arguments, is position dependent along the token
axis and has its own seeded parameters by module name), so a swapped stream, a wrong block index, a stale temb, a different id order or a
different summation moves the result by O(0.1).

Two users, one set of stand-ins:
  * tests/golden/make_ref_wiring_golden.py (build container only) hands them to the reference's methods, compiled from /root/reference by
    tests/golden/ref_harness.py, and writes the inputs and the reference's outputs to tests/golden/ref_wiring.safetensors;
  * tests/test_ref_wiring_cpu.py hands the SAME stand-ins to oracle.unigen_ref.flux_base_forward / sd3_base_forward / flux_moe_forward - the
    functions unigen_flux_forward / unigen_sd3_forward run with state-dict modules - and requires the fixture's outputs.
"""
from __future__ import annotations

import zlib
from types import SimpleNamespace
from typing import Dict, Optional

import torch

F64 = torch.float64
D, P, C_IN, B, GRID, T = 16, 8, 5, 2, (2, 3), 4
N = GRID[0] * GRID[1]
E = 6


class StandIns:
    """Deterministic stand-in modules. Parameters are drawn per (name, shape) from a generator seeded by crc32 - the same on every host."""

    def __init__(self, seed: int = 20251005):
        self.seed = seed
        self._cache: Dict[tuple, torch.Tensor] = {}

    def w(self, name: str, *shape: int, scale: float = 1.0) -> torch.Tensor:
        key = (name, shape)
        if key not in self._cache:
            g = torch.Generator().manual_seed(zlib.crc32(f"{self.seed}:{name}:{shape}".encode()))
            self._cache[key] = torch.randn(*shape, generator=g, dtype=F64) * scale / max(shape[0], 1) ** 0.5
        return self._cache[key]

    def ids_feat(self, ids: Optional[torch.Tensor]):
        if ids is None:
            return 0.0
        return torch.tanh(ids.to(F64) @ self.w("ids", 3, D, scale=0.7))

    @staticmethod
    def _tok(t: torch.Tensor, like: torch.Tensor) -> torch.Tensor:
        return t if t.dim() == like.dim() else t[:, None]

    def joint(self, name, x, enc, temb, hd_ids=None, encoder_hd_ids=None, context_out: bool = True):
        """(enc', x'): both streams see each other's (position-weighted) summary, the temb, their ids and their neighbour token."""
        xs, es = x + self.ids_feat(hd_ids), enc + self.ids_feat(encoder_hd_ids)
        pos_x = torch.linspace(0.5, 1.5, xs.shape[1], dtype=F64)[None, :, None]
        pos_e = torch.linspace(1.5, 0.5, es.shape[1], dtype=F64)[None, :, None]
        m_x, m_e = (xs * pos_x).mean(1), (es * pos_e).mean(1)
        xo = x + 0.5 * torch.tanh(xs @ self.w(name + ".xx", D, D) + torch.roll(xs, 1, 1) @ self.w(name + ".xr", D, D)
                                  + (m_e @ self.w(name + ".xe", D, D))[:, None] + self._tok(temb @ self.w(name + ".xt", D, D), x))
        if not context_out:
            return None, xo
        eo = enc + 0.5 * torch.tanh(es @ self.w(name + ".ee", D, D) + torch.roll(es, 1, 1) @ self.w(name + ".er", D, D)
                                    + (m_x @ self.w(name + ".ex", D, D))[:, None] + self._tok(temb @ self.w(name + ".et", D, D), enc))
        return eo, xo

    def single(self, name, h, temb, hd_ids=None):
        hs = h + self.ids_feat(hd_ids)
        pos = torch.linspace(0.5, 1.5, hs.shape[1], dtype=F64)[None, :, None]
        return h + 0.5 * torch.tanh(hs @ self.w(name + ".hh", D, D) + torch.roll(hs, 1, 1) @ self.w(name + ".hr", D, D)
                                    + ((hs * pos).mean(1) @ self.w(name + ".hm", D, D))[:, None] + self._tok(temb @ self.w(name + ".ht", D, D), h))

    def linear(self, name, x, d_in: int = D, d_out: int = D):
        return x @ self.w(name + ".w", d_in, d_out) + self.w(name + ".b", 1, d_out, scale=0.1)[0]

    def zero_res(self, name, z):
        """stands for a controlnet_add_* projection; small, so the residual stream stays O(1) through 57 layers"""
        return 0.05 * self.linear(name, z)

    def patch_embed(self, name, latent):
        """[B, C, H, W] -> [B, H W, D] (stands for PatchEmbed: the wiring only hands the latent through)."""
        return self.linear(name, latent.flatten(2).transpose(1, 2), C_IN, D)

    def tte(self, name, timestep, pooled, guidance=None):
        y = pooled @ self.w(name + ".p", P, D) + timestep.to(F64)[:, None] * 1e-3 * self.w(name + ".t", 1, D)
        if guidance is not None:
            y = y + guidance.to(F64)[:, None] * 1e-3 * self.w(name + ".g", 1, D)
        return torch.tanh(y)

    def moe_layer(self, name, *, choice_expert_input, hidden_states, condition_hidden_states, encoder_hidden_states, temb, condition_temb,
                  condition_pooled_projections, pooled_projections):
        """-> (expert_h, expert_c, l_aux, exp_counts): stands for MOELayer.forward (gate + dispatch + experts + combine)."""
        ch, x, c = choice_expert_input, hidden_states, condition_hidden_states
        eh = torch.tanh(ch @ self.w(name + ".h_ch", D, D) + x @ self.w(name + ".h_x", D, D) + (encoder_hidden_states.mean(1) @ self.w(name + ".h_e", D, D))[:, None]
                        + (temb @ self.w(name + ".h_t", D, D))[:, None] + (pooled_projections @ self.w(name + ".h_p", P, D))[:, None])
        ec = torch.tanh(c @ self.w(name + ".c_c", D, D) + torch.roll(ch, 1, 1) @ self.w(name + ".c_ch", D, D)
                        + (condition_temb @ self.w(name + ".c_t", D, D))[:, None] + (condition_pooled_projections @ self.w(name + ".c_p", P, D))[:, None])
        l_aux = (ch * c).mean()
        exp_counts = (ch.reshape(-1, D)[:, :E] > 0).sum(0)
        return eh, ec, l_aux, exp_counts


def make_ids(rows: int, cols: int, off: float = 0.0) -> torch.Tensor:
    r, c = torch.meshgrid(torch.arange(rows, dtype=F64), torch.arange(cols, dtype=F64), indexing="ij")
    return torch.stack([torch.full_like(r, off), r + off, c - off], -1).reshape(-1, 3)


def inputs(seed: int, n_cond: int = 1, sd3: bool = False) -> Dict[str, object]:
    g = torch.Generator().manual_seed(seed)
    rn = lambda *s: torch.randn(*s, generator=g, dtype=F64)
    d = dict(x=rn(B, N, D), enc=rn(B, T, D), temb=rn(B, D), pooled=rn(B, P), timestep=torch.tensor([750.0, 250.0], dtype=F64),
             guidance=torch.tensor([3500.0, 1000.0], dtype=F64), img_ids=make_ids(*GRID), prompt_ids=rn(T, 3).round())
    if sd3:
        d.update(cond=rn(B, C_IN, *GRID), cond_pooled=rn(B, P), condition_ids=make_ids(*GRID, off=1.0))
    elif n_cond == 1:
        d.update(cond=rn(B, N, C_IN), cond_pooled=rn(B, P), condition_ids=make_ids(*GRID, off=1.0))
    else:
        d.update(cond=[rn(B, N, C_IN) for _ in range(n_cond)], cond_pooled=[rn(B, P) for _ in range(n_cond)],
                 condition_ids=[make_ids(*GRID, off=1.0 + k) for k in range(n_cond)])
    return d


# name, class, control params, block counts, conditioning_scale, guidance used
FLUX_CASES = [
    dict(name="flux_schnell_depth", n_double=19, n_single=38, n_cj=9, n_cs=19, use_rope=True, use_single_trans_blocks=True,
         single_block_control_method="overall_add", use_shared_expert=True, use_consis_module=False, use_pooled_prompt_embeds=True, scale=0.7, guidance=False, n_cond=1),
    dict(name="flux_single_add_norope_guidance", n_double=19, n_single=38, n_cj=9, n_cs=19, use_rope=False, use_single_trans_blocks=True,
         single_block_control_method="single_add", use_shared_expert=True, use_consis_module=True, use_pooled_prompt_embeds=False, scale=1.3, guidance=True, n_cond=1),
    dict(name="flux_consis_rope_no_single", n_double=5, n_single=6, n_cj=2, n_cs=3, use_rope=True, use_single_trans_blocks=False,
         single_block_control_method="overall_add", use_shared_expert=False, use_consis_module=True, use_pooled_prompt_embeds=True, scale=1.0, guidance=False, n_cond=1),
    dict(name="multi3_schnell_depth", n_double=19, n_single=38, n_cj=9, n_cs=19, use_rope=True, use_single_trans_blocks=True,
         single_block_control_method="overall_add", use_shared_expert=True, use_consis_module=False, use_pooled_prompt_embeds=True, scale=0.9, guidance=False, n_cond=3),
    dict(name="multi3_single_add_norope", n_double=4, n_single=6, n_cj=2, n_cs=3, use_rope=False, use_single_trans_blocks=True,
         single_block_control_method="single_add", use_shared_expert=True, use_consis_module=False, use_pooled_prompt_embeds=False, scale=1.1, guidance=True, n_cond=3),
]
SD3_CASES = [
    dict(name="sd3_medium_depth", n_layers=24, n_control=24, use_rope=False, use_shared_expert=True, use_pooled_prompt_embeds=True, scale=0.8),
    dict(name="sd3_half_control_rope", n_layers=24, n_control=12, use_rope=True, use_shared_expert=True, use_pooled_prompt_embeds=False, scale=1.2),
    dict(name="sd3_no_shared", n_layers=6, n_control=6, use_rope=False, use_shared_expert=False, use_pooled_prompt_embeds=True, scale=1.0),
]
MOE_CASES = [  # moe_forward alone: UniGenFlux's (with the consistency module) and UniGenBase's
    dict(name="moe_flux_rope_consis_shared", cls="flux", use_rope=True, use_consis_module=True, use_shared_expert=True),
    dict(name="moe_flux_norope_shared", cls="flux", use_rope=False, use_consis_module=False, use_shared_expert=True),
    dict(name="moe_base_rope_shared", cls="base", use_rope=True, use_consis_module=False, use_shared_expert=True),
    dict(name="moe_base_norope_noshared", cls="base", use_rope=False, use_consis_module=False, use_shared_expert=False),
]


# ---------------------------------------------------------------------------------------------------------------------
# the stand-ins behind the ORACLE's module interface (oracle/unigen_ref.py: flux_modules / sd3_modules document it)
# ---------------------------------------------------------------------------------------------------------------------

def oracle_flux_modules(si: StandIns):
    m = SimpleNamespace()
    sel = dict(k=0)
    m.select_condition = lambda k: sel.update(k=k)
    m.double = lambda i, x, enc, temb: si.joint(f"transformer_blocks.{i}", x, enc, temb)
    m.single = lambda j, h, temb: si.single(f"single_transformer_blocks.{j}", h, temb)
    m.control_joint = lambda k, z, enc, temb, hd, ehd: si.joint(f"control_joint_trans_blocks.{k}", z, enc, temb, hd, ehd)
    m.control_single = lambda k, h, temb, hd: si.single(f"control_single_trans_blocks.{k}", h, temb, hd)
    m.add_joint = lambda k, z: si.zero_res(f"controlnet_add_joint_blocks.{k}", z)
    m.add_single = lambda k, z: si.zero_res(f"controlnet_add_single_blocks.{k}", z)
    m.control_x_embedder = lambda c: si.linear("control_x_embedder", c, C_IN, D)
    m.control_context_embedder = lambda e: si.linear("control_context_embedder", e)
    m.control_time_text_embed = lambda t, p, g: si.tte("control_time_text_embed", t, p, g)
    m.control_condition_embed = lambda t, p, g: si.tte("control_condition_embed", t, p, g)
    m.moe_layer = lambda **kw: si.moe_layer("moe_layer", **kw)
    m.shared_expert = lambda k, x, enc, temb, hd, ehd: si.joint(f"shared_expert.{k}", x, enc, temb, hd, ehd)
    m.consis_module = lambda k, x, enc, temb, hd, ehd: si.joint(f"consis_module.{k}", x, enc, temb, hd, ehd)
    return m


def oracle_sd3_modules(si: StandIns, n_layers: int):
    m = SimpleNamespace()
    m.block = lambda i, x, enc, temb: si.joint(f"transformer_blocks.{i}", x, enc, temb, context_out=i != n_layers - 1)
    m.control_block = lambda k, z, enc, temb, hd, ehd: si.joint(f"control_transformer_blocks.{k}", z, enc, temb, hd, ehd)
    m.add = lambda k, z: si.zero_res(f"controlnet_add_blocks.{k}", z)
    m.control_pos_embed_input = lambda lat: si.patch_embed("control_pos_embed_input", lat)
    m.control_context_embedder = lambda e: si.linear("control_context_embedder", e)
    m.control_time_text_embed = lambda t, p: si.tte("control_time_text_embed", t, p)
    m.control_condition_embed = lambda t, p: si.tte("control_condition_embed", t, p)
    m.moe_layer = lambda **kw: si.moe_layer("moe_layer", **kw)
    m.shared_expert = lambda k, x, enc, temb, hd, ehd: si.joint(f"shared_expert.{k}", x, enc, temb, hd, ehd)
    return m


def flux_cfg(R, case):
    """oracle FluxConfig carrying the case's flags and block counts (cn_* are properties of num_layers // single_control_dev: 19 // 2 = 9,
    38 // 2 = 19 as in the reference; the small cases pick counts that divide the same way)."""
    dev = 2
    cfg = R.FluxConfig(num_layers=case["n_double"], num_single_layers=case["n_single"], condition_nums=case["n_cond"], use_rope=case["use_rope"],
                       use_pooled_prompt_embeds=case["use_pooled_prompt_embeds"], use_shared_expert=case["use_shared_expert"],
                       use_consis_module=case["use_consis_module"], use_single_trans_blocks=case["use_single_trans_blocks"], single_control_dev=dev,
                       single_block_control_method=case["single_block_control_method"])
    assert cfg.cn_joint_layers == case["n_cj"] and cfg.cn_single_layers == case["n_cs"], (cfg.cn_joint_layers, cfg.cn_single_layers, case)
    return cfg
