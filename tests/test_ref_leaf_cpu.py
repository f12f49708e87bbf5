"""The oracle against fixtures that ORIGINATE IN THE REFERENCE (tests/golden/ref_leaf.safetensors, written by
tests/golden/make_ref_leaf_golden.py from the reference's own torch-only functions: src/UniGenUtils.py:194-228, 340-373 and
src/UniGenTransformer.py:925-967 / 225-267). These rows of the oracle are reference-pinned; everything else stays restated."""
import os

import pytest
import torch
from safetensors import safe_open

from oracle import unigen_ref as R

FIX = os.path.join(os.path.dirname(__file__), "golden", "ref_leaf.safetensors")


@pytest.fixture(scope="module")
def fx():
    with safe_open(FIX, "pt") as f:
        return {k: f.get_tensor(k) for k in f.keys()}


def rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))


DT = {"f32": torch.float32, "bf16": torch.bfloat16}


@pytest.mark.parametrize("tag", ["f32", "bf16"])
def test_modulated_flatten_matches_the_reference_function(fx, tag):
    """modulated_flatten (src/UniGenUtils.py:204-228): the oracle's literal form is the reference's 3-D branch op for op (bit-equal, also in bf16
    eager); the `Linear(s * x)` form used at scale equals it exactly in fp32 and within bf16 rounding in bf16; the 2-D (conv1d) branch is the same
    function with a per-sample s."""
    dt = DT[tag]
    x, w, s3, s2 = (fx[f"mf.{k}"].to(dt) for k in ("x", "w", "s3", "s2"))
    y3, y2 = fx[f"mf.y3_{tag}"], fx[f"mf.y2_{tag}"]
    assert torch.equal(R.modulated_flatten_literal(x, w, s3), y3) or rel(R.modulated_flatten_literal(x, w, s3), y3) < 2e-7
    tol = 2e-6 if tag == "f32" else 4e-3
    assert rel(R.modulated_linear(x, w, s3), y3) <= tol
    assert rel(R.modulated_linear(x, w, s2[:, None, :]), y2) <= tol
    if tag == "bf16":      # the restated form is no further from the reference's fp32 result than the reference's own bf16 evaluation
        t3, t2 = fx["mf.y3_f32"], fx["mf.y2_f32"]
        assert rel(R.modulated_linear(x, w, s3), t3) <= 1.1 * rel(y3, t3)
        assert rel(R.modulated_linear(x, w, s2[:, None, :]), t2) <= 1.1 * rel(y2, t2)


def _state(fx, key):
    return {"p.linear.weight": fx[key + ".w"], "p.linear.bias": fx[key + ".b"]}


CASES = [("d128", "zero"), ("d128", "zerox"), ("d128", "cont"), ("d1536", "zerox"), ("d1536", "cont"), ("d3072", "zero")]


@pytest.mark.parametrize("tag", ["f32", "bf16"])
@pytest.mark.parametrize("name,kind", CASES)
def test_adaln_forwards_match_the_reference_functions(fx, name, kind, tag):
    """adanorm_forward :354-363 (AdaLayerNormZero, per-sample and the reference's per-token extension), sd35adanormX_forward :340-352,
    adanormContinuous_forward :365-373 against the oracle's adaln_zero / adaln_zero_any / adaln_zero_x / adaln_continuous: same torch ops in the
    same order, so bf16 eager is bit-equal too."""
    dt = DT[tag]
    st = _state(fx, f"ada.{name}.{kind}")
    x = fx[f"ada.{name}.x"].to(dt)
    for etag in ("sample", "token"):
        if kind == "cont" and etag == "token":
            continue
        emb = fx[f"ada.{name}.emb_{etag}"].to(dt)
        if kind == "zero":
            outs = R.adaln_zero_any(st, "p", x, emb)
            if etag == "sample":
                outs2 = R.adaln_zero(st, "p", x, emb)       # the per-sample form the FLUX blocks use
                assert all(torch.equal(a, b) for a, b in zip(outs, outs2))
        elif kind == "zerox":
            outs = R.adaln_zero_x(st, "p", x, emb)
        else:
            outs = (R.adaln_continuous(st, "p", x, emb),)
        for i, o in enumerate(outs):
            want = fx[f"ada.{name}.{kind}.{etag}.{tag}.o{i}"]
            assert o.shape == want.shape and o.dtype == want.dtype
            assert torch.equal(o, want) or rel(o, want) < (1e-6 if tag == "f32" else 1e-3), (name, kind, etag, i, rel(o, want))


def _expert_state(fx, E=3):
    st = {}
    for e in range(E):
        for i in (0, 1):
            for j in (0, 1):
                for k in ("weight", "bias"):
                    st[f"moe.moe_layer.experts.deepspeed_experts.{e}.{i}.{j}.{k}"] = fx[f"expert.w.{e}.{i}.{j}.{k}"]
    return st


@pytest.mark.parametrize("tag", ["f32", "bf16"])
def test_expert_forward_matches_the_reference_method(fx, tag):
    """UniGenFlux.expert_forward (src/UniGenTransformer.py:925-967) == UniGenBase.expert_forward (:225-267), run by the generator on dispatched
    [1, E, C, *] tensors: the oracle's literal form is bit-equal; the form used at scale within bf16 rounding and no worse than the reference's bf16."""
    dt = DT[tag]
    st = _expert_state(fx)
    h, c, pooled, cpooled = (fx[f"expert.{k}"][0].to(dt) for k in ("h", "c", "pooled", "cpooled"))
    want_h, want_c = fx[f"expert.{tag}.out_h"][0], fx[f"expert.{tag}.out_c"][0]
    yh, yc = R.expert_forward(st, h, c, pooled, cpooled, literal=True)
    assert (torch.equal(yh, want_h) and torch.equal(yc, want_c)) or max(rel(yh, want_h), rel(yc, want_c)) < 1e-6
    yh, yc = R.expert_forward(st, h, c, pooled, cpooled, literal=False)
    tol = 2e-6 if tag == "f32" else 6e-3
    assert rel(yh, want_h) <= tol and rel(yc, want_c) <= tol
    if tag == "bf16":
        th, tc = fx["expert.f32.out_h"][0], fx["expert.f32.out_c"][0]
        assert rel(yh, th) <= 1.1 * rel(want_h, th) and rel(yc, tc) <= 1.1 * rel(want_c, tc)


def test_zero_module_fixture_is_all_zero(fx):
    """zero_module (src/UniGenUtils.py:194-197) zeroes every parameter: what `controlnet_add_*` must look like after init_condition_block."""
    assert not fx["zero_module.weight"].any() and not fx["zero_module.bias"].any()


def _attn_state(fx, dt):
    m = {"to_q": "to_q", "to_k": "to_k", "to_v": "to_v", "add_q_proj": "add_q_proj", "add_k_proj": "add_k_proj", "add_v_proj": "add_v_proj",
         "to_out0": "to_out.0", "to_add_out": "to_add_out"}
    st = {}
    for src, dst in m.items():
        st[f"a.{dst}.weight"], st[f"a.{dst}.bias"] = fx[f"attn.w.{src}.weight"].to(dt), fx[f"attn.w.{src}.bias"].to(dt)
    return st


@pytest.mark.parametrize("tag", ["f32", "bf16"])
def test_joint_attention_matches_the_reference_processor(fx, tag):
    """JointAttnRopeProcessor.__call__ (src/UniGenUtils.py:533-622), the attention of the control branch's joint blocks, run from the reference's
    source with rope_embed = None and no q/k norms: per-head view, SAMPLE rows first in the concatenation, SDPA, split at the sample length,
    to_out[0] / to_add_out (skipped when context_pre_only). The oracle's sd3_attention is the same torch ops in the same order: bit-equal in
    bf16 eager too. Pins the concatenation order and the split of every sample-first attention of the oracle (joint_attention text_first=False
    and sd3_attention share them); the branches behind diffusers' RMSNorm / apply_rotary_emb stay restated."""
    dt = DT[tag]
    st = _attn_state(fx, dt)
    x, enc = fx["attn.x"].to(dt), fx["attn.enc"].to(dt)
    for cpo in (False, True):
        xo, eo = R.sd3_attention(st, "a", 2, x, enc, context_pre_only=cpo)
        assert torch.equal(xo, fx[f"attn.joint.cpo{int(cpo)}.{tag}.out"]) or rel(xo, fx[f"attn.joint.cpo{int(cpo)}.{tag}.out"]) < 2e-6
        if not cpo:
            assert torch.equal(eo, fx[f"attn.joint.cpo0.{tag}.ctx"]) or rel(eo, fx[f"attn.joint.cpo0.{tag}.ctx"]) < 2e-6
        else:
            assert eo is None                         # the reference hands back the raw context rows; every caller drops them (context_pre_only)
    so, _ = R.sd3_attention(st, "a", 2, x, None)
    assert torch.equal(so, fx[f"attn.self.{tag}.out"]) or rel(so, fx[f"attn.self.{tag}.out"]) < 2e-6
    # a context-first concatenation (the base FLUX blocks' order) is a different function: the fixture tells the two apart
    q = torch.cat([enc, x], 1)
    assert rel(R.sd3_attention(st, "a", 2, q[:, :x.shape[1]], q[:, x.shape[1]:])[0], fx[f"attn.joint.cpo0.{tag}.out"]) > 1e-2


@pytest.mark.skipif(not os.path.isdir("/root/reference/src"), reason="the reference tree exists in the build container only")
def test_committed_reference_fixtures_regenerate_bit_identically(tmp_path):
    """The pin verifies itself: every generator that executes the reference's own code (make_ref_leaf / _wiring / _pipeline)
    is run into a scratch directory and each tensor it writes must equal, bit for bit, the tensor of the same name in the committed fixture -
    and no committed tensor may be missing from the regenerated file. A fixture that drifted from the reference (or a generator edited
    without regenerating) fails here instead of silently re-defining what "the reference says"."""
    import subprocess
    import sys
    golden = os.path.join(os.path.dirname(__file__), "golden")
    env = dict(os.environ, UG_GOLDEN_OUT=str(tmp_path))
    for gen in ("make_ref_leaf_golden.py", "make_ref_wiring_golden.py", "make_ref_pipeline_golden.py"):
        r = subprocess.run([sys.executable, os.path.join(golden, gen)], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, f"{gen}: {r.stderr[-800:]}"
    produced = sorted(p for p in os.listdir(tmp_path) if p.endswith(".safetensors"))
    assert produced == ["ref_blocks.safetensors", "ref_leaf.safetensors", "ref_pipeline.safetensors", "ref_wiring.safetensors"]
    for name in produced:
        with safe_open(os.path.join(tmp_path, name), "pt") as new, safe_open(os.path.join(golden, name), "pt") as old:
            assert sorted(new.keys()) == sorted(old.keys()), name
            for k in old.keys():
                a, b = new.get_tensor(k), old.get_tensor(k)
                assert a.dtype == b.dtype and a.shape == b.shape and torch.equal(a.view(torch.uint8) if a.is_floating_point() else a,
                                                                                b.view(torch.uint8) if b.is_floating_point() else b), f"{name}:{k}"
