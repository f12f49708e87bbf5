"""Fixtures that ORIGINATE IN THE REFERENCE: its block bodies and its orchestration ("wiring") methods, run in the build container.

    python tests/golden/make_ref_wiring_golden.py     # needs /root/reference; writes tests/golden/ref_blocks.safetensors, ref_wiring.safetensors

Compiled from the reference's files by tests/golden/ref_harness.py (sha256-pinned sources, a namespace of torch symbols only, a builtin whitelist,
every free name checked; see its docstring):

  blocks  src/UniGenUtils.py  JointTransformerBlock.forward :438-522 (plain, context_pre_only, use_dual_attention, per-token temb) and
                              SD3SingleTransformerBlock.forward :386-414 (per-sample temb; per-token temb as the SD3 experts feed it).
          `self` is a bag whose attributes are what the reference's constructors put there, built from reference code wherever the reference has any:
          norm1 / norm1_context = functools.partial(adanorm_forward | sd35adanormX_forward | adanormContinuous_forward, module=bag) exactly as
          :379-382, :426-437 install them; attn / attn2 = the reference's JointAttnRopeProcessor.__call__ :533-622 bound to an `attn` bag of
          nn.Linear (no q/k norm, no RoPE: the branches behind diffusers' RMSNorm / apply_rotary_emb are tripwired); norm2 = nn.LayerNorm (no
          affine, eps 1e-6); ff = Linear -> GELU(tanh) -> Linear (diffusers FeedForward's arithmetic). `diffusers_attention` is a tripwire: the
          `_chunk_size is not None` branch is never taken.
  wiring  src/UniGenTransformer.py  UniGenFlux.moe_forward :969, .preprocess_moe_forward :1028, .control_forward :1070, .base_forward :1106;
          MultiCondtionUniGenFlux.preprocess_moe_forward :1275, .control_forward :1324; UniGenSD3.preprocess_moe_forward :498, .control_forward
          :539, .base_forward :581; UniGenBase.moe_forward :269. `self` carries the stand-in modules of tests/wiring_cases.py (deterministic
          float64 torch callables) at the attribute names the reference uses, and the reference's own methods bound to it
          (self.control_forward, self.preprocess_moe_forward, self.moe.forward = self.moe_forward as :199 / :859 do).

The fixtures hold tensors only (inputs, parameters, the reference's outputs). Nothing of the reference's text is stored, nothing of it travels.
"""
from __future__ import annotations

import functools
import os
import sys
from types import SimpleNamespace

import torch
from safetensors.torch import save_file
from torch import nn

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
from ref_harness import REF, bag, compile_reference_function, lin  # noqa: E402
import wiring_cases as W  # noqa: E402

U, T = "src/UniGenUtils.py", "src/UniGenTransformer.py"


# ======================================================================================================================
# (a) block bodies
# ======================================================================================================================

BLOCK_D, BLOCK_H = 128, 2
ATTN_NAMES = ("to_q", "to_k", "to_v", "add_q_proj", "add_k_proj", "add_v_proj", "to_out.0", "to_add_out")


def _attn_bag(mods: dict, prefix: str, ctx: bool, context_pre_only: bool):
    g = lambda n: mods[f"{prefix}.{n}"]
    kw = dict(heads=BLOCK_H, norm_q=None, norm_k=None, norm_added_q=None, norm_added_k=None, context_pre_only=context_pre_only,
              to_q=g("to_q"), to_k=g("to_k"), to_v=g("to_v"), to_out=nn.ModuleList([g("to_out.0"), nn.Dropout(0.0)]))
    if ctx:
        kw.update(add_q_proj=g("add_q_proj"), add_k_proj=g("add_k_proj"), add_v_proj=g("add_v_proj"))
        if not context_pre_only:
            kw.update(to_add_out=g("to_add_out"))
    return bag(**kw)


def block_params(g: torch.Generator, kind: str) -> dict:
    """nn modules of one block by state-dict name (diffusers' names: SURVEY 8(b))."""
    D = BLOCK_D
    m = {}
    dual, cpo, single = "dual" in kind, "cpo" in kind, kind.startswith("single")
    m["norm1.linear"] = lin(D, (9 if dual else 6) * D, g, 1.0)
    for n in ("to_q", "to_k", "to_v", "to_out.0"):
        m[f"attn.{n}"] = lin(D, D, g, 1.0)
    m["ff.net.0.proj"], m["ff.net.2"] = lin(D, 4 * D, g, 1.0), lin(4 * D, D, g, 1.0)
    if not single:
        m["norm1_context.linear"] = lin(D, (2 if cpo else 6) * D, g, 1.0)
        for n in ("add_q_proj", "add_k_proj", "add_v_proj"):
            m[f"attn.{n}"] = lin(D, D, g, 1.0)
        if not cpo:
            m["attn.to_add_out"] = lin(D, D, g, 1.0)
            m["ff_context.net.0.proj"], m["ff_context.net.2"] = lin(D, 4 * D, g, 1.0), lin(4 * D, D, g, 1.0)
    if dual:
        for n in ("to_q", "to_k", "to_v", "to_out.0"):
            m[f"attn2.{n}"] = lin(D, D, g, 1.0)
    return m


def block_self(mods: dict, kind: str, ref: dict):
    """The `self` a reference block forward sees. `ref`: the compiled reference definitions."""
    D = BLOCK_D
    dual, cpo, single = "dual" in kind, "cpo" in kind, kind.startswith("single")
    ln = lambda: nn.LayerNorm(D, elementwise_affine=False, eps=1e-6)
    ada = lambda name, fn: functools.partial(fn, module=bag(emb=None, silu=nn.SiLU(), linear=mods[name + ".linear"], norm=ln()))
    ff = lambda p: nn.Sequential(mods[p + ".net.0.proj"], nn.GELU(approximate="tanh"), mods[p + ".net.2"])

    def processor(attn_bag):      # diffusers Attention.forward hands (self, hidden_states, encoder_hidden_states=..., **kwargs) to its processor
        return lambda hidden_states, encoder_hidden_states=None, **kw: ref["attn_call"](None, attn_bag, hidden_states, encoder_hidden_states=encoder_hidden_states, **kw)

    s = SimpleNamespace(norm1=ada("norm1", ref["adax"] if dual else ref["ada"]), norm2=ln(), ff=ff("ff"), _chunk_size=None, _chunk_dim=0,
                        attn=processor(_attn_bag(mods, "attn", not single, cpo)))
    if not single:
        s.use_dual_attention, s.context_pre_only = dual, cpo
        s.norm1_context = ada("norm1_context", ref["adac"] if cpo else ref["ada"])
        if dual:
            s.attn2 = processor(_attn_bag(mods, "attn2", False, False))
        if not cpo:
            s.norm2_context, s.ff_context = ln(), ff("ff_context")
    return s


def make_blocks(ref: dict) -> dict:
    g = torch.Generator().manual_seed(12443 + 5)
    rnd = lambda *s, scale=1.0: (torch.randn(*s, generator=g) * scale).bfloat16().float()
    fx: dict = {}
    D, B, N, Tc = BLOCK_D, 2, 20, 12
    E, C = 2, 16                                    # single.token: E experts' capacity slots, batch 1 each (src/UniGenTransformer.py:248-262)
    x, enc, temb = (rnd(B, N, D) + rnd(B, N, 1)).bfloat16().float(), (rnd(B, Tc, D) + rnd(B, Tc, 1)).bfloat16().float(), rnd(B, D)   # rows with a non-zero mean
    xq, encq, temb_tok = rnd(B, Tc, D), rnd(B, Tc, D), rnd(B, Tc, D)          # per-token temb needs equal stream lengths (one temb for both AdaLNs)
    for k, v in dict(x=x, enc=enc, temb=temb, xq=xq, encq=encq, temb_tok=temb_tok).items():
        fx[f"in.{k}"] = v.bfloat16()
    # the SD3 experts' call (src/UniGenTransformer.py:261): hidden [1, C, D] = dispatched slots (empty slots zero), temb [1, C, D] = the per-sample
    # temb broadcast per token and dispatched the same way (src/UniGenUtils.py:107-109): rows of one sample share a vector, empty slots zero
    tz = rnd(B, D)
    sample_of_slot = torch.tensor([[0, 1, 0, 0, 1, 1, 0, 1, 1, 0, 0, 1, -1, -1, -1, -1], [1, 1, 0, 1, 0, 0, 0, 1, 0, 1, -1, -1, -1, -1, -1, -1]])
    xs = rnd(E, C, D)
    ts = torch.zeros(E, C, D)
    for e in range(E):
        for c in range(C):
            b = int(sample_of_slot[e, c])
            if b < 0:
                xs[e, c] = 0
            else:
                ts[e, c] = tz[b]
    fx["in.single_tok.x"], fx["in.single_tok.temb_rows"], fx["in.single_tok.sample_of_slot"] = xs.bfloat16(), tz.bfloat16(), sample_of_slot.to(torch.int32)

    kinds = ("joint.plain", "joint.cpo", "joint.dual", "joint.dual_cpo", "single.block", "single.expert0", "single.expert1")
    for kind in kinds:
        mods = block_params(g, kind)
        for n, mod in mods.items():
            fx[f"w.{kind}.{n}.weight"], fx[f"w.{kind}.{n}.bias"] = mod.weight.detach().bfloat16(), mod.bias.detach().bfloat16()
        for tag, dt in (("f32", torch.float32), ("bf16", torch.bfloat16)):
            for mod in mods.values():
                mod.to(dt)
            s = block_self(mods, kind, ref)
            with torch.no_grad():
                if kind.startswith("joint"):
                    eo, xo = ref["joint_fwd"](s, x.to(dt), enc.to(dt), temb.to(dt))
                    fx[f"out.{kind}.sample.{tag}.x"] = xo
                    assert (eo is None) == ("cpo" in kind)
                    if eo is not None:
                        fx[f"out.{kind}.sample.{tag}.enc"] = eo
                    if kind in ("joint.plain", "joint.dual"):        # per-token temb: the expand_gate_dim = False branches (AdaLN-Continuous cannot take one, :368)
                        eo, xo = ref["joint_fwd"](s, xq.to(dt), encq.to(dt), temb_tok.to(dt))
                        fx[f"out.{kind}.token.{tag}.x"], fx[f"out.{kind}.token.{tag}.enc"] = xo, eo
                elif kind == "single.block":
                    fx[f"out.{kind}.sample.{tag}.x"] = ref["single_fwd"](s, x.to(dt), temb.to(dt))
                    fx[f"out.{kind}.token.{tag}.x"] = ref["single_fwd"](s, xq.to(dt), temb_tok.to(dt))
                else:
                    e = int(kind[-1])
                    fx[f"out.{kind}.token.{tag}.x"] = ref["single_fwd"](s, xs[e][None].to(dt), ts[e][None].to(dt))
            for mod in mods.values():
                mod.float()
    return {k: v.detach().clone().contiguous() for k, v in fx.items()}


# ======================================================================================================================
# (b) wiring
# ======================================================================================================================

class _Moe:
    """self.moe: callable through .forward (the reference assigns self.moe.forward = self.moe_forward, :199 / :859), with .moe_layer"""

    def __init__(self, layer):
        self.moe_layer, self.forward = layer, None

    def __call__(self, *a, **k):
        return self.forward(*a, **k)


class _MoeLayer:
    """self.moe.moe_layer: a stand-in for MOELayer.forward; sets .l_aux / .exp_counts as the real one does (src/UniGenUtils.py:99)"""

    def __init__(self, si: W.StandIns):
        self.si, self.l_aux, self.exp_counts = si, None, None

    def __call__(self, choice_expert_input=None, hidden_states=None, condition_hidden_states=None, used_token=None, **kw):
        assert used_token is None and set(kw) == {"encoder_hidden_states", "temb", "condition_temb", "condition_pooled_projections", "pooled_projections"}, sorted(kw)
        eh, ec, self.l_aux, self.exp_counts = self.si.moe_layer("moe_layer", choice_expert_input=choice_expert_input, hidden_states=hidden_states,
                                                                 condition_hidden_states=condition_hidden_states, **kw)
        return eh, ec


def _jkw(kw, rope_token):
    """unpack a control-branch joint_attention_kwargs dict: {} (no RoPE) or {hd_ids, [encoder_hd_ids], rope_embed}"""
    kw = kw or {}
    if not kw:
        return None, None
    assert kw["rope_embed"] is rope_token, "the wiring handed over a different rope_embed module"
    assert set(kw) <= {"hd_ids", "encoder_hd_ids", "rope_embed"}
    return kw["hd_ids"], kw.get("encoder_hd_ids")


def _ctl_joint(si, name, rope_token, context_out=True):
    def f(hidden_states, encoder_hidden_states, temb, joint_attention_kwargs=None):
        hd, ehd = _jkw(joint_attention_kwargs, rope_token)
        return si.joint(name, hidden_states, encoder_hidden_states, temb, hd, ehd, context_out=context_out)
    return f


def _ctl_single(si, name, rope_token):
    def f(hidden_states, temb, joint_attention_kwargs=None):
        hd, ehd = _jkw(joint_attention_kwargs, rope_token)
        assert ehd is None
        return si.single(name, hidden_states, temb, hd)
    return f


def _tte(si, name):
    def f(*a):              # (timestep, pooled) or (timestep, guidance, pooled): CombinedTimestep[Guidance]TextProjEmbeddings' positional order
        t, g, p = (a[0], None, a[1]) if len(a) == 2 else a
        return si.tte(name, t, p, g)
    return f


def flux_self(si, case, ref, multi: bool):
    ROPE_BASE, ROPE_CTL, POS = object(), object(), object()       # image_rotary_emb, control_pos_embed_input, pos_embed: passed through untouched

    def base_double(name):
        def f(hidden_states, encoder_hidden_states, temb, image_rotary_emb=None, joint_attention_kwargs=None):
            assert image_rotary_emb is ROPE_BASE and not joint_attention_kwargs
            return si.joint(name, hidden_states, encoder_hidden_states, temb)
        return f

    def base_single(name):
        def f(hidden_states, temb, image_rotary_emb=None, joint_attention_kwargs=None):
            assert image_rotary_emb is ROPE_BASE and not joint_attention_kwargs
            return si.single(name, hidden_states, temb)
        return f

    s = SimpleNamespace(use_rope=case["use_rope"], use_consis_module=case["use_consis_module"], use_shared_expert=case["use_shared_expert"],
                        use_pooled_prompt_embeds=case["use_pooled_prompt_embeds"], single_block_control_method=case["single_block_control_method"],
                        pos_embed=POS, control_pos_embed_input=ROPE_CTL)
    s.transformer_blocks = [base_double(f"transformer_blocks.{i}") for i in range(case["n_double"])]
    s.single_transformer_blocks = [base_single(f"single_transformer_blocks.{j}") for j in range(case["n_single"])]
    s.control_joint_trans_blocks = [_ctl_joint(si, f"control_joint_trans_blocks.{k}", ROPE_CTL) for k in range(case["n_cj"])]
    s.controlnet_add_joint_blocks = [functools.partial(si.zero_res, f"controlnet_add_joint_blocks.{k}") for k in range(case["n_cj"])]
    if case["use_single_trans_blocks"]:
        s.control_single_trans_blocks = [_ctl_single(si, f"control_single_trans_blocks.{k}", ROPE_CTL) for k in range(case["n_cs"])]
        s.controlnet_add_single_blocks = [functools.partial(si.zero_res, f"controlnet_add_single_blocks.{k}") for k in range(case["n_cs"])]
    s.control_x_embedder = lambda c: si.linear("control_x_embedder", c, W.C_IN, W.D)
    s.control_context_embedder = functools.partial(si.linear, "control_context_embedder")
    s.control_time_text_embed, s.control_condition_embed = _tte(si, "control_time_text_embed"), _tte(si, "control_condition_embed")
    # moe_forward hands rope_embed = self.pos_embed to the shared experts / consistency module (:993-1018)
    s.shared_expert = [_ctl_joint(si, f"shared_expert.{k}", POS) for k in range(2)]
    s.consis_module = [_ctl_joint(si, f"consis_module.{k}", POS) for k in range(2)]
    s.moe = _Moe(_MoeLayer(si))
    s.moe.forward = functools.partial(ref["flux.moe_forward"], s)
    pre = "multi" if multi else "flux"
    s.preprocess_moe_forward = functools.partial(ref[pre + ".preprocess_moe_forward"], s)
    s.control_forward = functools.partial(ref[pre + ".control_forward"], s)
    return s, ROPE_BASE


def sd3_self(si, case, ref):
    ROPE = object()
    L = case["n_layers"]

    def base(name, last):
        def f(hidden_states, encoder_hidden_states, temb, joint_attention_kwargs=None):
            assert not joint_attention_kwargs
            return si.joint(name, hidden_states, encoder_hidden_states, temb, context_out=not last)
        return f

    s = SimpleNamespace(use_rope=case["use_rope"], use_shared_expert=case["use_shared_expert"], use_pooled_prompt_embeds=case["use_pooled_prompt_embeds"],
                        use_encoder_hidden_states=True, cn_method="add", rope_embed=ROPE)
    s.transformer_blocks = [base(f"transformer_blocks.{i}", i == L - 1) for i in range(L)]
    s.control_transformer_blocks = [_ctl_joint(si, f"control_transformer_blocks.{k}", ROPE) for k in range(case["n_control"])]
    s.controlnet_add_blocks = [functools.partial(si.zero_res, f"controlnet_add_blocks.{k}") for k in range(case["n_control"])]
    s.control_pos_embed_input = functools.partial(si.patch_embed, "control_pos_embed_input")
    s.control_context_embedder = functools.partial(si.linear, "control_context_embedder")
    s.control_time_text_embed, s.control_condition_embed = _tte(si, "control_time_text_embed"), _tte(si, "control_condition_embed")
    s.shared_expert = [_ctl_joint(si, f"shared_expert.{k}", ROPE) for k in range(2)]
    s.moe = _Moe(_MoeLayer(si))
    s.moe.forward = functools.partial(ref["base.moe_forward"], s)
    s.preprocess_moe_forward = functools.partial(ref["sd3.preprocess_moe_forward"], s)
    s.control_forward = functools.partial(ref["sd3.control_forward"], s)
    return s


def make_wiring(ref: dict) -> dict:
    fx: dict = {}

    def put(prefix, d):
        for k, v in d.items():
            if isinstance(v, (list, tuple)):
                for i, t in enumerate(v):
                    fx[f"{prefix}.{k}.{i}"] = t.detach().clone().contiguous()
            elif isinstance(v, torch.Tensor):
                fx[f"{prefix}.{k}"] = v.detach().clone().contiguous()

    si = W.StandIns()
    for ci, case in enumerate(W.FLUX_CASES):
        multi = case["n_cond"] > 1
        inp = W.inputs(100 + ci, n_cond=case["n_cond"])
        s, rope_base = flux_self(si, case, ref, multi)
        guidance = inp["guidance"] if case["guidance"] else None
        with torch.no_grad():
            out = ref["flux.base_forward"](s, hidden_states=inp["x"], condition_hidden_states=inp["cond"], encoder_hidden_states=inp["enc"],
                                           pooled_projections=inp["pooled"], condition_pooled_projections=inp["cond_pooled"], timestep=inp["timestep"],
                                           conditioning_scale=case["scale"], temb=inp["temb"], joint_attention_kwargs=None, image_rotary_emb=rope_base,
                                           guidance=guidance, img_ids=inp["img_ids"], prompt_ids=inp["prompt_ids"], condition_ids=inp["condition_ids"])
        put(f"{case['name']}.in", inp)
        put(f"{case['name']}.out", dict(x=out["blocks_hidden_states"], enc=out["block_ctx_hidden_states"], moe_loss=out["moe_loss"].reshape(1),
                                        exp_count=out["exp_count"]))
    for ci, case in enumerate(W.SD3_CASES):
        inp = W.inputs(200 + ci, sd3=True)
        s = sd3_self(si, case, ref)
        with torch.no_grad():
            out = ref["sd3.base_forward"](s, hidden_states=inp["x"], condition_hidden_states=inp["cond"], encoder_hidden_states=inp["enc"],
                                          pooled_projections=inp["pooled"], condition_pooled_projections=inp["cond_pooled"], timestep=inp["timestep"],
                                          conditioning_scale=case["scale"], temb=inp["temb"], joint_attention_kwargs=None,
                                          img_ids=inp["img_ids"], prompt_ids=inp["prompt_ids"], condition_ids=inp["condition_ids"])
        assert out["block_ctx_hidden_states"] is None        # the last base block is context_pre_only
        put(f"{case['name']}.in", inp)
        put(f"{case['name']}.out", dict(x=out["blocks_hidden_states"], moe_loss=out["moe_loss"].reshape(1), exp_count=out["exp_count"]))
    for ci, case in enumerate(W.MOE_CASES):
        inp = W.inputs(300 + ci)
        g = torch.Generator().manual_seed(400 + ci)
        c, ctrl_enc = torch.randn(W.B, W.N, W.D, generator=g, dtype=W.F64), torch.randn(W.B, W.T, W.D, generator=g, dtype=W.F64)
        control_temb, condition_temb = torch.randn(W.B, W.D, generator=g, dtype=W.F64), torch.randn(W.B, W.D, generator=g, dtype=W.F64)
        POS = object()
        s = SimpleNamespace(use_rope=case["use_rope"], use_consis_module=case["use_consis_module"], use_shared_expert=case["use_shared_expert"],
                            pos_embed=POS, rope_embed=POS, moe=_Moe(_MoeLayer(si)))
        s.shared_expert = [_ctl_joint(si, f"shared_expert.{k}", POS) for k in range(2)]
        s.consis_module = [_ctl_joint(si, f"consis_module.{k}", POS) for k in range(2)]
        jk = dict(img_ids=inp["img_ids"], prompt_ids=inp["prompt_ids"], condition_ids=inp["condition_ids"], rope_embed=object()) if case["use_rope"] else dict()
        with torch.no_grad():
            (oh, oc), l_aux, cnt = ref[case["cls"] + ".moe_forward"](s, hidden_states=inp["x"], condition_hidden_states=c, encoder_hidden_states=ctrl_enc,
                                                                     temb=control_temb, condition_temb=condition_temb, condition_pooled_projections=inp["cond_pooled"],
                                                                     pooled_projections=inp["pooled"], joint_attention_kwargs=jk)
        put(f"{case['name']}.in", dict(inp, c=c, ctrl_enc=ctrl_enc, control_temb=control_temb, condition_temb=condition_temb))
        put(f"{case['name']}.out", dict(h=oh, c=oc, l_aux=l_aux.reshape(1), exp_count=cnt))
    return fx


def main() -> None:
    if not os.path.isdir(REF):
        sys.exit(f"{REF} not found: the fixtures can only be regenerated in the build container")
    ref, lines = {}, {}

    def comp(key, path, name, cls=None, **kw):
        ref[key], lines[key] = compile_reference_function(path, name, cls, **kw)

    comp("ada", U, "adanorm_forward"); comp("adax", U, "sd35adanormX_forward"); comp("adac", U, "adanormContinuous_forward")
    comp("attn_call", U, "__call__", "JointAttnRopeProcessor", not_taken=("apply_rotary_emb",))
    comp("joint_fwd", U, "forward", "JointTransformerBlock", not_taken=("diffusers_attention",))
    comp("single_fwd", U, "forward", "SD3SingleTransformerBlock")
    for key, cls in (("flux", "UniGenFlux"), ("multi", "MultiCondtionUniGenFlux"), ("sd3", "UniGenSD3")):
        comp(key + ".preprocess_moe_forward", T, "preprocess_moe_forward", cls)
        comp(key + ".control_forward", T, "control_forward", cls)
    comp("flux.moe_forward", T, "moe_forward", "UniGenFlux"); comp("base.moe_forward", T, "moe_forward", "UniGenBase")
    comp("flux.base_forward", T, "base_forward", "UniGenFlux"); comp("sd3.base_forward", T, "base_forward", "UniGenSD3")
    print("compiled reference definitions:", ", ".join(f"{k} :{v}" for k, v in lines.items()))
    # ONE metadata key: safetensors writes the header's metadata map in an unspecified order, a second key would make the file's bytes vary run to run
    meta = {"origin": "reference methods executed by tests/golden/make_ref_wiring_golden.py; first lines: " + ", ".join(f"{k}:{v}" for k, v in lines.items())}
    for name, fx in (("ref_blocks", make_blocks(ref)), ("ref_wiring", make_wiring(ref))):
        out = os.path.join(os.environ.get("UG_GOLDEN_OUT", HERE), name + ".safetensors")
        save_file(fx, out, metadata=meta)
        print(f"wrote {out}: {len(fx)} tensors, {os.path.getsize(out) / 1024:.0f} KiB")


if __name__ == "__main__":
    main()
