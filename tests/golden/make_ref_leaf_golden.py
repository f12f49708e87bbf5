"""Fixtures that ORIGINATE IN THE REFERENCE: its torch-only leaf functions, run in the build container.

    python tests/golden/make_ref_leaf_golden.py            # needs /root/reference; writes tests/golden/ref_leaf_*.safetensors

The reference cannot be imported (deepspeed / diffusers / peft / ipdb are absent: ordinary ModuleNotFoundError), but a handful of its
functions are self-contained PyTorch. This script parses the reference's files with `ast` AT RUN TIME, compiles ONLY these definitions

    src/UniGenUtils.py        zero_module :194, modulated_flatten :204, sd35adanormX_forward :340, adanorm_forward :354,
                              adanormContinuous_forward :365
                              JointAttnRopeProcessor.__call__ :533 (the control branch's attention: sample-first concatenation, split, to_add_out
                              unless context_pre_only) with rope_embed = None and an `attn` bag without q/k norms - the two branches that need
                              diffusers' apply_rotary_emb / RMSNorm are not taken (a tripwire raises if they were)
    src/UniGenTransformer.py  UniGenFlux.expert_forward :925 and UniGenBase.expert_forward :225 (method bodies, called as plain functions)

into a namespace that holds nothing but `torch`, `torch.nn`, `torch.nn.functional`, `typing.List` and (for expert_forward) the reference's own
`modulated_flatten` compiled the same way, and CHECKS that every global name the compiled code can reach is one of those - so no stand-in for a
third-party symbol is reachable. The `module` / `self` arguments are attribute bags of plain `nn.Linear` / `nn.SiLU` / `nn.LayerNorm` objects
(what diffusers' AdaLayerNormZero* and deepspeed's `Experts.deepspeed_experts` hold at these attribute names); they are inputs, not code.
Nothing of the reference's text is written anywhere: the fixtures hold tensors only (seeded inputs, parameters, the reference's outputs in
bf16 eager and in fp32). /root/reference does not travel to the GPU box; the tests read the fixtures.
"""
from __future__ import annotations

import os
import sys

import torch
from safetensors.torch import save_file
from torch import nn

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from ref_harness import REF, bag, compile_reference_function, lin as _lin  # noqa: E402


def main() -> None:
    if not os.path.isdir(REF):
        sys.exit(f"{REF} not found: the fixtures can only be regenerated in the build container")
    U = "src/UniGenUtils.py"
    zero_module, l0 = compile_reference_function(U, "zero_module")
    modulated_flatten, l1 = compile_reference_function(U, "modulated_flatten")
    adax, l2 = compile_reference_function(U, "sd35adanormX_forward")
    ada, l3 = compile_reference_function(U, "adanorm_forward")
    adac, l4 = compile_reference_function(U, "adanormContinuous_forward")
    T = "src/UniGenTransformer.py"
    exp_flux, l5 = compile_reference_function(T, "expert_forward", "UniGenFlux", {"modulated_flatten": modulated_flatten})
    exp_base, l6 = compile_reference_function(T, "expert_forward", "UniGenBase", {"modulated_flatten": modulated_flatten})
    attn_call, l7 = compile_reference_function(U, "__call__", "JointAttnRopeProcessor", not_taken=("apply_rotary_emb",))
    print("compiled reference definitions at lines", l0, l1, l2, l3, l4, l5, l6, l7)
    g = torch.Generator().manual_seed(12443)
    rnd = lambda *s, scale=1.0: (torch.randn(*s, generator=g) * scale).bfloat16().float()
    fx: dict = {}

    def put(prefix, **t):
        for k, v in t.items():
            fx[f"{prefix}.{k}"] = v.detach().clone().contiguous()

    # ---- modulated_flatten, both branches (src/UniGenUtils.py:204-228) --------------------------------------------------------
    b, n, ci, co = 3, 24, 128, 128
    x, w = rnd(b, n, ci), rnd(co, ci, scale=ci ** -0.5)
    s3, s2 = rnd(b, n, ci, scale=0.5) + 1.0, rnd(b, ci, scale=0.5) + 1.0
    s3, s2 = s3.bfloat16().float(), s2.bfloat16().float()
    put("mf", x=x.bfloat16(), w=w.bfloat16(), s3=s3.bfloat16(), s2=s2.bfloat16())
    for tag, dt in (("f32", torch.float32), ("bf16", torch.bfloat16)):
        put("mf", **{f"y3_{tag}": modulated_flatten(x.to(dt), w.to(dt), s3.to(dt)), f"y2_{tag}": modulated_flatten(x.to(dt), w.to(dt), s2.to(dt))})

    # ---- AdaLayerNormZero / SD35AdaLayerNormZeroX / AdaLayerNormContinuous patches (:340-373) --------------------------------
    # per-sample emb [B, De] and per-token emb [B, L, De]; D = 128 (generic kernel), D = 1536 / 3072 (the 16-byte fast kernel's widths: SD3.5's
    # AdaLN-Zero-X / continuous and FLUX's AdaLN-Zero; a narrow emb keeps the linear small - the reference functions do not care)
    for name, D, De, B, L, kinds in (("d128", 128, 64, 2, 12, ("zero", "zerox", "cont")), ("d1536", 1536, 8, 2, 2, ("zerox", "cont")),
                                     ("d3072", 3072, 8, 2, 2, ("zero",))):
        xs = rnd(B, L, D) + rnd(B, L, 1)      # rows with a non-zero mean
        xs = xs.bfloat16().float()
        e2, e3 = rnd(B, De), rnd(B, L, De)
        norm = nn.LayerNorm(D, elementwise_affine=False, eps=1e-6)
        for kind, fn, k in (("zero", lambda m, h, e: ada(m, h, emb=e), 6), ("zerox", adax, 9), ("cont", adac, 2)):
            if kind not in kinds:
                continue
            lin = _lin(De, k * D, g)
            put(f"ada.{name}.{kind}", w=lin.weight.bfloat16(), b=lin.bias.bfloat16())
            for etag, e in (("sample", e2), ("token", e3)):
                if kind == "cont" and etag == "token":
                    continue       # the reference chunks a 3-D emb on dim 1 (:368): shapes cannot broadcast, it raises - no vector exists
                for tag, dt in (("f32", torch.float32), ("bf16", torch.bfloat16)):
                    m = bag(emb=None, silu=nn.SiLU(), linear=lin, norm=norm).to(dt)
                    with torch.no_grad():
                        outs = fn(m, xs.to(dt), e.to(dt))
                    outs = outs if isinstance(outs, tuple) else (outs,)
                    put(f"ada.{name}.{kind}.{etag}.{tag}", **{f"o{i}": o for i, o in enumerate(outs)})
                    lin.float()
        put(f"ada.{name}", x=xs.bfloat16(), emb_sample=e2.bfloat16(), emb_token=e3.bfloat16())

    # ---- zero_module (:194-197) ----------------------------------------------------------------------------------------------
    zl = zero_module(_lin(8, 8, g))
    put("zero_module", weight=zl.weight, bias=zl.bias)

    # ---- expert_forward (src/UniGenTransformer.py:925-967 and :225-267), the modulated experts of the shipped configs ---------
    E, Cc, D, P = 3, 24, 128, 64
    experts = nn.ModuleList([nn.ModuleList([nn.ModuleList([_lin(D, D, g, 1.0), _lin(P, D, g, 1.0)]), nn.ModuleList([_lin(D, D, g, 1.0), _lin(P, D, g, 1.0)])])
                             for _ in range(E)])
    h, c = rnd(1, E, Cc, D), rnd(1, E, Cc, D)
    temb, ctemb = rnd(1, E, Cc, P), rnd(1, E, Cc, P)      # dispatched but unused on the modulated path
    pooled, cpooled = rnd(1, E, Cc, P), rnd(1, E, Cc, P)
    # slots of one (expert, sample) share their pooled vector in the real path (a 2-D kwarg broadcast per token, src/UniGenUtils.py:107-109);
    # empty slots are zero rows: keep both properties in the fixture
    for e in range(E):
        pooled[0, e, :16] = pooled[0, e, 0]; pooled[0, e, 16:22] = pooled[0, e, 16]; pooled[0, e, 22:] = 0
        cpooled[0, e, :16] = cpooled[0, e, 0]; cpooled[0, e, 16:22] = cpooled[0, e, 16]; cpooled[0, e, 22:] = 0
        h[0, e, 22:] = 0; c[0, e, 22:] = 0
    put("expert", h=h.bfloat16(), c=c.bfloat16(), pooled=pooled.bfloat16(), cpooled=cpooled.bfloat16())
    for e in range(E):
        for i in (0, 1):
            for j in (0, 1):
                put(f"expert.w.{e}.{i}.{j}", weight=experts[e][i][j].weight.bfloat16(), bias=experts[e][i][j].bias.bfloat16())
    for tag, dt in (("f32", torch.float32), ("bf16", torch.bfloat16)):
        ex = experts.to(dt)
        me = bag(num_local_experts=E, use_modulate=False, use_rope=True,
                 moe=bag(moe_layer=bag(experts=bag(deepspeed_experts=ex))))
        outs = {}
        for nm, fn in (("flux", exp_flux), ("base", exp_base)):
            with torch.no_grad():
                oh, oc = fn(me, h.to(dt), condition_hidden_states=c.to(dt), temb=temb.to(dt), condition_temb=ctemb.to(dt),
                            condition_pooled_projections=cpooled.to(dt), pooled_projections=pooled.to(dt))
            outs[nm] = (oh, oc)
        assert all(torch.equal(a, b_) for a, b_ in zip(outs["flux"], outs["base"])), "the two expert_forward definitions disagree"
        put(f"expert.{tag}", out_h=outs["flux"][0], out_c=outs["flux"][1])
        experts.float()

    # ---- JointAttnRopeProcessor.__call__ (src/UniGenUtils.py:533-622): joint attention, sample rows first; no RoPE, no q/k norm ----------------------
    # (drawn after everything above: the earlier tensors of the fixture keep their values)
    H, dh, B, N, Tc = 2, 64, 2, 20, 12
    D = H * dh
    names = ("to_q", "to_k", "to_v", "add_q_proj", "add_k_proj", "add_v_proj", "to_out0", "to_add_out")
    lins = {nm: _lin(D, D, g, 1.0) for nm in names}
    xs, es = rnd(B, N, D), rnd(B, Tc, D)
    put("attn", x=xs.bfloat16(), enc=es.bfloat16())
    for nm, lin in lins.items():
        put(f"attn.w.{nm}", weight=lin.weight.bfloat16(), bias=lin.bias.bfloat16())
    for tag, dt in (("f32", torch.float32), ("bf16", torch.bfloat16)):
        mods = {nm: lin.to(dt) for nm, lin in lins.items()}
        for cpo in (False, True):
            attn = bag(heads=H, norm_q=None, norm_k=None, norm_added_q=None, norm_added_k=None, context_pre_only=cpo,
                       to_q=mods["to_q"], to_k=mods["to_k"], to_v=mods["to_v"], add_q_proj=mods["add_q_proj"], add_k_proj=mods["add_k_proj"],
                       add_v_proj=mods["add_v_proj"], to_out=nn.ModuleList([mods["to_out0"], nn.Dropout(0.0)]), to_add_out=mods["to_add_out"])
            with torch.no_grad():
                ho, eo = attn_call(None, attn, xs.to(dt), encoder_hidden_states=es.to(dt))
            put(f"attn.joint.cpo{int(cpo)}.{tag}", out=ho, ctx=eo)
        with torch.no_grad():
            so = attn_call(None, attn, xs.to(dt))
        put(f"attn.self.{tag}", out=so)
        for lin in lins.values():
            lin.float()

    out = os.path.join(os.environ.get("UG_GOLDEN_OUT", HERE), "ref_leaf.safetensors")
    # ONE metadata key: safetensors writes the metadata map in an unspecified order, a second key makes the file's bytes vary run to run
    save_file(fx, out, metadata={"origin": "reference functions executed by tests/golden/make_ref_leaf_golden.py, seed 12443"})
    print(f"wrote {out}: {len(fx)} tensors, {os.path.getsize(out) / 1024:.0f} KiB")


if __name__ == "__main__":
    main()
