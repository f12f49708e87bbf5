"""The oracle's WIRING and SD3 BLOCK BODIES against fixtures that ORIGINATE IN THE REFERENCE (tests/golden/ref_wiring.safetensors,
tests/golden/ref_blocks.safetensors, written by tests/golden/make_ref_wiring_golden.py from the reference's own methods:
src/UniGenTransformer.py:269-296, 498-623, 969-1180, 1275-1357 and src/UniGenUtils.py:386-522).

The functions held here are the ones unigen_flux_forward / unigen_sd3_forward execute (oracle/unigen_ref.py: flux_base_forward,
flux_preprocess_moe_forward, flux_moe_forward, sd3_base_forward, sd3_preprocess_moe_forward; sd3_joint_block, sd3_single_block) - not a copy made
for the test: test_oracle_forwards_run_the_pinned_wiring checks that by patching them."""
import os

import pytest
import torch
from safetensors import safe_open

import wiring_cases as W
from oracle import unigen_ref as R

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _load(name):
    with safe_open(os.path.join(GOLD, name), "pt") as f:
        return {k: f.get_tensor(k) for k in f.keys()}


@pytest.fixture(scope="module")
def wx():
    return _load("ref_wiring.safetensors")


@pytest.fixture(scope="module")
def bx():
    return _load("ref_blocks.safetensors")


def rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))


def same(a, b, tol=1e-12):
    """float64 stand-ins: the same torch ops in the same order - equal, or a last-bit difference of another host's matmul kernels"""
    return a.shape == b.shape and (torch.equal(a, b) or rel(a, b) <= tol)


def _inp(fx, name, n_cond=1):
    g = lambda k: fx[f"{name}.in.{k}"]
    d = {k: g(k) for k in ("x", "enc", "temb", "pooled", "timestep", "guidance", "img_ids", "prompt_ids")}
    if n_cond == 1:
        d.update(cond=g("cond"), cond_pooled=g("cond_pooled"), condition_ids=g("condition_ids"))
    else:
        for k in ("cond", "cond_pooled", "condition_ids"):
            d[k] = [g(f"{k}.{i}") for i in range(n_cond)]
    return d


@pytest.mark.parametrize("case", W.FLUX_CASES, ids=lambda c: c["name"])
def test_flux_wiring_matches_the_reference_methods(wx, case):
    """UniGenFlux.base_forward + control_forward + preprocess_moe_forward + moe_forward (and MultiCondtionUniGenFlux's two overrides) run from the
    reference's source on the stand-in modules: block map int(i / (n / n_c)) at 19 / 9 and 38 / 19, CoMoE once with the text stream after base
    block 0, control blocks on the base stream, the single_hd flag, overall_add vs single_add, use_rope on / off (which ids reach which block),
    conditioning_scale != 1, guidance, zeroed pooled vector, the consistency module's two calls, three conditions summed, last condition's loss."""
    inp = _inp(wx, case["name"], case["n_cond"])
    cfg = W.flux_cfg(R, case)
    m = W.oracle_flux_modules(W.StandIns())
    x, enc, moe_out = R.flux_base_forward(m, cfg, inp["x"], inp["cond"], inp["enc"], inp["pooled"], inp["cond_pooled"], inp["timestep"],
                                          conditioning_scale=case["scale"], temb=inp["temb"], guidance=inp["guidance"] if case["guidance"] else None,
                                          img_ids=inp["img_ids"], prompt_ids=inp["prompt_ids"], condition_ids=inp["condition_ids"])
    o = lambda k: wx[f"{case['name']}.out.{k}"]
    assert same(x, o("x")), rel(x, o("x"))
    assert same(enc, o("enc")), rel(enc, o("enc"))
    assert same(moe_out["l_aux"].reshape(1), o("moe_loss")) and torch.equal(moe_out["exp_counts"], o("exp_count"))


@pytest.mark.parametrize("case", W.SD3_CASES, ids=lambda c: c["name"])
def test_sd3_wiring_matches_the_reference_methods(wx, case):
    """UniGenSD3.base_forward + control_forward + preprocess_moe_forward + UniGenBase.moe_forward: 24 blocks with 24 and with 12 control blocks
    (the int(i / interval) map), use_rope on / off, with and without shared experts."""
    inp = _inp(wx, case["name"])
    cfg = R.SD3Config(num_layers=case["n_layers"], use_rope=case["use_rope"], use_shared_expert=case["use_shared_expert"],
                      use_pooled_prompt_embeds=case["use_pooled_prompt_embeds"])
    m = W.oracle_sd3_modules(W.StandIns(), case["n_layers"])
    ids = dict(img_ids=inp["img_ids"], prompt_ids=inp["prompt_ids"], condition_ids=inp["condition_ids"])
    x, enc, moe_out = R.sd3_base_forward(m, cfg, inp["x"], inp["cond"], inp["enc"], inp["pooled"], inp["cond_pooled"], inp["timestep"],
                                         conditioning_scale=case["scale"], temb=inp["temb"], n_control=case["n_control"], ids=ids)
    o = lambda k: wx[f"{case['name']}.out.{k}"]
    assert enc is None
    assert same(x, o("x")), rel(x, o("x"))
    assert same(moe_out["l_aux"].reshape(1), o("moe_loss")) and torch.equal(moe_out["exp_counts"], o("exp_count"))


@pytest.mark.parametrize("case", W.MOE_CASES, ids=lambda c: c["name"])
def test_moe_forward_matches_the_reference_methods(wx, case):
    """UniGenFlux.moe_forward :969-1026 and UniGenBase.moe_forward :269-296 alone (one function in the oracle)."""
    n = case["name"]
    g = lambda k: wx[f"{n}.in.{k}"]
    cfg = R.FluxConfig(use_rope=case["use_rope"], use_consis_module=case["use_consis_module"], use_shared_expert=case["use_shared_expert"])
    ids = dict(img_ids=g("img_ids"), prompt_ids=g("prompt_ids"), condition_ids=g("condition_ids")) if case["use_rope"] else None
    m = W.oracle_flux_modules(W.StandIns())
    (oh, oc), l_aux, cnt = R.flux_moe_forward(m, cfg, g("x"), g("c"), ctrl_enc=g("ctrl_enc"), control_temb=g("control_temb"), condition_temb=g("condition_temb"),
                                              pooled=g("pooled"), cond_pooled=g("cond_pooled"), ids=ids)
    assert same(oh, wx[f"{n}.out.h"]) and same(oc, wx[f"{n}.out.c"])
    assert same(l_aux.reshape(1), wx[f"{n}.out.l_aux"]) and torch.equal(cnt, wx[f"{n}.out.exp_count"])


def test_fixture_tells_wiring_slips_apart(wx):
    """The stand-ins are sensitive to the slips the fixture is there to catch: each of these one-line changes to the wiring moves the output by
    far more than the tolerance."""
    case = W.FLUX_CASES[0]
    inp = _inp(wx, case["name"])
    want = wx[f"{case['name']}.out.x"]

    def run(cfg=None, mutate=None, **over):
        cfg = cfg or W.flux_cfg(R, case)
        m = W.oracle_flux_modules(W.StandIns())
        if mutate:
            mutate(m)
        kw = dict(conditioning_scale=case["scale"], temb=inp["temb"], guidance=None, img_ids=inp["img_ids"], prompt_ids=inp["prompt_ids"],
                  condition_ids=inp["condition_ids"])
        kw.update(over)
        return R.flux_base_forward(m, cfg, inp["x"], inp["cond"], inp["enc"], inp["pooled"], inp["cond_pooled"], inp["timestep"], **kw)[0]

    assert same(run(), want)
    assert rel(run(conditioning_scale=1.0), want) > 1e-3                                   # the scale
    assert rel(run(prompt_ids=inp["img_ids"][:W.T]), want) > 1e-3                          # which ids reach the control blocks

    def chained(m):                                                                        # control blocks fed by each other instead of the base stream
        prev, orig = {}, m.control_joint
        def f(k, z, enc, temb, hd, ehd):
            out = orig(k, prev.get("z", z), enc, temb, hd, ehd)
            prev["z"] = out[1]
            return out
        m.control_joint = f
    assert rel(run(mutate=chained), want) > 1e-3

    def floor_map(m):                                                                      # k = i // 2 instead of int(i / (19 / 9))
        orig, calls = m.control_joint, []
        def f(k, z, enc, temb, hd, ehd):
            i = len(calls); calls.append(k)
            return orig(min(i // 2, 8), z, enc, temb, hd, ehd)
        m.control_joint = f
    assert rel(run(mutate=floor_map), want) > 1e-3
    import dataclasses
    assert rel(run(cfg=dataclasses.replace(W.flux_cfg(R, case), single_block_control_method="single_add")), want) > 1e-3


def test_oracle_forwards_run_the_pinned_wiring(monkeypatch):
    """unigen_flux_forward / unigen_sd3_forward go THROUGH the functions held above (no second copy of the loops)."""
    calls = []
    for name in ("flux_base_forward", "flux_preprocess_moe_forward", "flux_moe_forward", "sd3_base_forward", "sd3_preprocess_moe_forward"):
        orig = getattr(R, name)
        monkeypatch.setattr(R, name, (lambda o, n: (lambda *a, **k: (calls.append(n), o(*a, **k))[1]))(orig, name))
    cfg = R.FluxConfig(num_layers=2, num_single_layers=2, attention_head_dim=16, num_attention_heads=2, joint_attention_dim=32, pooled_projection_dim=16,
                       axes_dims_rope=(4, 6, 6))
    st = R.make_state(cfg, seed=1, std=0.05)
    inp = R.make_inputs(cfg, B=1, grid=2, T=3)
    R.unigen_flux_forward(st, cfg, timestep=torch.tensor([0.5]), dtype=torch.float32, **inp)
    assert calls == ["flux_base_forward", "flux_preprocess_moe_forward", "flux_moe_forward"]
    calls.clear()
    scfg = R.SD3Config(sample_size=4, num_layers=2, attention_head_dim=16, num_attention_heads=2, joint_attention_dim=32, caption_projection_dim=32,
                       pooled_projection_dim=16, pos_embed_max_size=4, dual_attention_layers=(0,))
    sst = R.make_sd3_state(scfg, seed=1, std=0.05)
    sinp = R.make_sd3_inputs(scfg, B=1, hw=4, T=3)
    R.unigen_sd3_forward(sst, scfg, timestep=torch.tensor([500.0]), dtype=torch.float32, **sinp)
    assert calls == ["sd3_base_forward", "sd3_preprocess_moe_forward", "flux_moe_forward"]


# ---------------------------------------------------------------------------------------------------------------------
# block bodies
# ---------------------------------------------------------------------------------------------------------------------

DT = {"f32": torch.float32, "bf16": torch.bfloat16}


def _block_state(bx, kind, dt):
    pre = f"w.{kind}."
    return {"b." + k[len(pre):]: v.to(dt) for k, v in bx.items() if k.startswith(pre)}


def _close(got, want, tag):
    # same torch ops in the same order as the reference's block: bit-equal in bf16 eager and fp32 here; the tolerance only allows for another
    # host's matmul kernels (fp32 summation order), never a bf16 rounding-point difference (that would be >= 4e-3 on some element)
    return got.shape == want.shape and got.dtype == want.dtype and (torch.equal(got, want) or rel(got, want) <= (2e-6 if tag == "f32" else 1e-3))


@pytest.mark.parametrize("tag", ["f32", "bf16"])
@pytest.mark.parametrize("kind", ["joint.plain", "joint.cpo", "joint.dual", "joint.dual_cpo"])
def test_sd3_joint_block_matches_the_reference_forward(bx, kind, tag):
    """JointTransformerBlock.forward (src/UniGenUtils.py:438-522) run from the reference's source: plain, context_pre_only (AdaLayerNormContinuous on
    the context, no context output), use_dual_attention (SD35AdaLayerNormZeroX, attn2 on the block's INPUT), both; per-sample and per-token temb."""
    dt = DT[tag]
    st = _block_state(bx, kind, dt)
    cpo, dual = "cpo" in kind, "dual" in kind
    x, enc, temb = (bx[f"in.{k}"].to(dt) for k in ("x", "enc", "temb"))
    eo, xo = R.sd3_joint_block(st, "b", 2, x, enc, temb, context_pre_only=cpo, dual=dual)
    assert _close(xo, bx[f"out.{kind}.sample.{tag}.x"], tag), rel(xo, bx[f"out.{kind}.sample.{tag}.x"])
    if cpo:
        assert eo is None
    else:
        assert _close(eo, bx[f"out.{kind}.sample.{tag}.enc"], tag)
        xq, encq, tt = (bx[f"in.{k}"].to(dt) for k in ("xq", "encq", "temb_tok"))
        eo, xo = R.sd3_joint_block(st, "b", 2, xq, encq, tt, context_pre_only=False, dual=dual)
        assert _close(xo, bx[f"out.{kind}.token.{tag}.x"], tag) and _close(eo, bx[f"out.{kind}.token.{tag}.enc"], tag)


@pytest.mark.parametrize("tag", ["f32", "bf16"])
def test_sd3_single_block_matches_the_reference_forward(bx, tag):
    """SD3SingleTransformerBlock.forward (src/UniGenUtils.py:386-414): per-sample temb, per-token temb, and the SD3 experts' call - batch 1 over an
    expert's capacity slots with the dispatched per-token temb, empty slots zero (src/UniGenTransformer.py:261-262)."""
    dt = DT[tag]
    st = _block_state(bx, "single.block", dt)
    x, temb, xq, tt = (bx[f"in.{k}"].to(dt) for k in ("x", "temb", "xq", "temb_tok"))
    assert _close(R.sd3_single_block(st, "b", 2, x, temb), bx[f"out.single.block.sample.{tag}.x"], tag)
    assert _close(R.sd3_single_block(st, "b", 2, xq, tt), bx[f"out.single.block.token.{tag}.x"], tag)
    xs, rows, sos = bx["in.single_tok.x"].to(dt), bx["in.single_tok.temb_rows"].to(dt), bx["in.single_tok.sample_of_slot"].long()
    for e in range(2):
        ts = torch.where((sos[e] >= 0)[:, None], rows[sos[e].clamp_min(0)], torch.zeros((), dtype=dt))
        st = _block_state(bx, f"single.expert{e}", dt)
        assert _close(R.sd3_single_block(st, "b", 2, xs[e][None], ts[None]), bx[f"out.single.expert{e}.token.{tag}.x"], tag)


# ---- the pipelines' control-image preparation (src/UniGenPipeline.py:107, :457) ---------------------------------------------------------------------
def _pipeline_fixture():
    from safetensors import safe_open
    with safe_open(os.path.join(os.path.dirname(__file__), "golden", "ref_pipeline.safetensors"), "pt") as f:
        return {k: f.get_tensor(k) for k in f.keys()}


def test_prepare_image_of_both_pipelines_equals_the_reference_methods():
    """`prepare_image` of the two pipeline twins against outputs of the reference's own methods (tests/golden/make_ref_pipeline_golden.py): one image
    for the whole batch, one per prompt x num_images_per_prompt, packed latents, the CFG doubling, guess mode, the one-channel depth map."""
    import importlib
    mod = importlib.import_module("src.UniGenPipeline")
    fx = _pipeline_fixture()
    seen = 0
    for kind, cls in (("flux", mod.UniGenFLUXPipeline), ("sd3", mod.UniGenSD3Pipeline)):
        pipe = cls.from_pretrained(None, transformer=None)
        for name in sorted({k.split(".")[1] for k in fx if k.startswith(kind + ".")}):
            img, (bs, nipp, cfg, guess), want = fx[f"{kind}.{name}.image"], fx[f"{kind}.{name}.args"].tolist(), fx[f"{kind}.{name}.out"]
            kw = dict(do_classifier_free_guidance=bool(cfg), guess_mode=bool(guess)) if kind == "sd3" else {}
            got = pipe.prepare_image(image=img, width=img.shape[-1], height=img.shape[-2], batch_size=bs, num_images_per_prompt=nipp, device="cpu",
                                     dtype=torch.bfloat16, **kw)
            assert got.dtype == want.dtype and got.shape == want.shape and torch.equal(got, want), (kind, name)
            seen += 1
    assert seen == 9
