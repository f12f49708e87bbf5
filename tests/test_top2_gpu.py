"""GPU parity of top-2 gating (control_params.top_num = 2 -> MoE(..., k = 2), /root/reference/src/UniGenTransformer.py:162,197 / :808,857 /
:1565,1650 -> deepspeed 0.16.5 sharded_moe.top2gating): the routing kernels through the C ABI against the oracle's statement-by-statement
restatement (oracle/unigen_ref.py top2gating, dense S x E x C form, and routing_top2, index form), and the UniGenFlux /
MultiCondtionUniGenFlux / UniGenSD3 forwards with top_num = 2 against the oracle's, fp32 verification path and bf16 product path.
deepspeed is not part of /root/reference, so this row is parity-unpinned like every other deepspeed-derived one."""
import importlib

import pytest
import torch
import torch.nn.functional as F

from oracle import unigen_ref as R
from tests.util import report, rel_l2

pytestmark = pytest.mark.gpu
BF, F32 = torch.bfloat16, torch.float32


def _rand(g, *shape, scale=1.0):
    return (torch.randn(*shape, generator=g) * scale).to(BF)


def _gumbel(g, *shape):
    u = torch.rand(*shape, generator=g).clamp_(1e-7, 1 - 1e-7)
    return -torch.log(-torch.log(u))


@pytest.mark.parametrize("S,E,D,sampling", [(128, 6, 256, True), (1000, 12, 128, True), (4099, 2, 64, True), (777, 3, 128, False)])
def test_top2_routing_dispatch_combine(gpu, S, E, D, sampling):
    from unigen_amd import ops
    g = torch.Generator().manual_seed(S + E)
    x, c = _rand(g, S, D), _rand(g, S, D)
    wg = _rand(g, E, D, scale=0.2)
    wg[0] += 0.05                                    # a favoured expert: its queue overflows and second choices get dropped
    noise = _gumbel(g, S, E) if sampling else None
    C = R.moe_capacity(S, E, capacity_factor=2.0)
    gates = torch.empty(S, E, device=gpu, dtype=F32)
    idx = torch.empty(2, S, device=gpu, dtype=torch.int32)
    ops.moe_gate_top2(x.to(gpu), c.to(gpu), wg.to(gpu), None if noise is None else noise.to(gpu), gates, idx)
    logits = F.linear((x + c).float(), wg.float())
    ref_gates = F.softmax(logits, dim=1)
    m = report(f"moe_top2_gates_S{S}", gates, ref_gates)
    assert m["rel_l2"] <= 1e-5, m
    # both choices must agree with the oracle wherever neither arg-max is a floating-point near-tie
    ridx0 = R.routing_top2(ref_gates, logits, noise, C)[0]
    top = torch.topk(ref_gates, 2, dim=1)[0]
    noisy = (logits if noise is None else logits + noise).masked_fill(F.one_hot(ridx0[0], E).bool(), float("-inf"))
    top_n = torch.topk(noisy, min(2, E - 1) if E > 2 else 1, dim=1)[0]
    clear = (top[:, 0] - top[:, 1]) > 1e-5
    if E > 2:
        clear &= (top_n[:, 0] - top_n[:, 1]) > 1e-4
    assert torch.equal(idx.cpu().long()[:, clear], ridx0[:, clear])
    assert float(clear.float().mean()) > 0.99
    # from here on the device's own gates and choices are the input, so that every comparison is exact
    gates_h, idx_h = gates.cpu(), idx.cpu().long()
    ridx, rslot, rtos, rw = R.routing_top2(gates_h, logits, noise, C, idx=idx_h)
    same_choices = torch.equal(idx_h, ridx0)          # no near-tie fell the other way: the dense form (host arg-max) describes the same routing
    l_aux_ref, cw, dm, cnt_ref = R.top2gating(logits, noise, C)
    slot = torch.empty(2, S, device=gpu, dtype=torch.int32)
    tos = torch.empty(E, C, device=gpu, dtype=torch.int32)
    w = torch.empty(2, S, device=gpu, dtype=F32)
    cnt = torch.empty(E, device=gpu, dtype=torch.int64)
    l_aux = torch.empty(1, device=gpu, dtype=F32)
    ops.moe_capacity_top2(gates, idx, C, slot, tos, w, cnt, l_aux)
    assert torch.equal(slot.cpu().long(), rslot), "slot assignment differs from deepspeed top2gating"
    assert torch.equal(tos.cpu().long(), rtos)
    assert torch.equal(cnt.cpu(), torch.stack([(idx_h == e).sum() for e in range(E)]))
    assert int((rslot < 0).sum()) > 0 or E == 2, "the case should drop some choices"
    assert torch.equal(w.cpu(), rw), float((w.cpu() - rw).abs().max())          # same fp32 operations in the same order
    l_aux_dev = float(torch.mean(gates_h.mean(0) * F.one_hot(idx_h[0], E).float().mean(0)) * E * E)      # top2gating's formula on the device's gates
    assert abs(float(l_aux) - l_aux_dev) <= 1e-5 * abs(l_aux_dev)
    # the dense tensors of top2gating (from the HOST logits) say the same thing as the index form (host gates differ in the last bits)
    cw_idx = torch.zeros(S, E, C)
    for k in range(2):
        kept = rslot[k] >= 0
        s_ar = torch.arange(S)[kept]
        cw_idx[s_ar, ridx[k][kept], rslot[k][kept]] += rw[k][kept]
    if same_choices:
        assert abs(float(l_aux) - float(l_aux_ref)) <= 1e-5 * abs(float(l_aux_ref))
        assert torch.equal(cnt.cpu(), cnt_ref)
        assert torch.equal(cw_idx.bool(), dm)
        assert torch.allclose(cw_idx, cw, rtol=1e-4, atol=1e-7)
    dm = cw_idx.bool()
    # dispatch: the token_of_slot layout is the one ug_moe_dispatch_modulate reads
    B = 1 if S % 2 else 2
    N = S // B
    mod = _rand(g, E, B, D)
    out = torch.empty(E, C, D, device=gpu, dtype=BF)
    ops.moe_dispatch_modulate(x.to(gpu), None, mod.to(gpu), tos, out, E=E, capacity=C, tokens_per_sample=N, mod_estride=B * D, mod_bstride=D)
    xd = torch.einsum("sec,sm->ecm", dm.to(BF).float(), x.float())                                     # src/UniGenUtils.py:140
    samp = torch.where(rtos >= 0, rtos // N, torch.zeros_like(rtos))
    ref = (mod.float()[torch.arange(E)[:, None], samp] * xd).to(BF)
    m = report(f"moe_top2_dispatch_S{S}", out, ref)
    assert m["mismatch_frac"] == 0.0, m
    # combine: einsum("sec,ecm->sm") in bf16 (fp32 accumulation, one rounding) + the CoMoE residual sums
    yh, yc, xs, cs = _rand(g, E, C, D), _rand(g, E, C, D), _rand(g, S, D), _rand(g, S, D)
    cwb = cw_idx.to(BF).float()
    eh = torch.einsum("sec,ecm->sm", cwb, yh.float()).to(BF)
    ec = torch.einsum("sec,ecm->sm", cwb, yc.float()).to(BF)
    o = torch.empty(S, D, device=gpu, dtype=BF)
    ops.moe_combine_topk(yh.to(gpu), yc.to(gpu), w, idx, slot, o, E=E, capacity=C, xs=xs.to(gpu), cs=cs.to(gpu))
    ref = (xs + eh) + (cs + ec)
    m = report(f"moe_top2_combine_S{S}", o, ref)
    assert m["mismatch_frac"] == 0.0, m
    ops.moe_combine_topk(yh.to(gpu), yc.to(gpu), w, idx, slot, o, E=E, capacity=C, accumulate=True)
    m = report(f"moe_top2_combine_acc_S{S}", o, ref + (eh + ec))
    assert m["mismatch_frac"] == 0.0, m
    # a slice of the token axis (one sample) keeps the parent's choice stride; the row-mapped shared-expert buffer
    if B == 2:
        o2 = torch.empty(N, D, device=gpu, dtype=BF)
        ops.moe_combine_topk(yh.to(gpu), yc.to(gpu), w[:, N:], idx[:, N:], slot[:, N:], o2, E=E, capacity=C)
        m = report(f"moe_top2_combine_slice_S{S}", o2, (eh + ec)[N:])
        assert m["mismatch_frac"] == 0.0, m
        xc = torch.cat([xs.view(B, N, D), cs.view(B, N, D)], 1).contiguous().to(gpu).view(B * 2 * N, D)
        ops.moe_combine_topk(yh.to(gpu), yc.to(gpu), w, idx, slot, o, E=E, capacity=C, xs=xc, cs=xc[N:], s_map=ops.RowMap(N, 2 * N))
        m = report(f"moe_top2_combine_rowmap_S{S}", o, ref)
        assert m["mismatch_frac"] == 0.0, m
    # K = 1 with the gate probability as the weight is ug_moe_combine
    o1, o1k = torch.empty(S, D, device=gpu, dtype=BF), torch.empty(S, D, device=gpu, dtype=BF)
    p1 = gates.gather(1, idx[0].long().unsqueeze(1)).squeeze(1).contiguous()
    ops.moe_combine(yh.to(gpu), yc.to(gpu), gates, idx[0], slot[0], o1, E=E, capacity=C)
    ops.moe_combine_topk(yh.to(gpu), yc.to(gpu), p1.view(1, S), idx[:1], slot[:1], o1k, E=E, capacity=C)
    assert torch.equal(o1, o1k)


def test_top2_fp32_twins(gpu):
    """The verification twins of the two typed top-2 entry points."""
    from unigen_amd import ops
    S, E, D = 300, 4, 64
    g = torch.Generator().manual_seed(3)
    x, c, wg = torch.randn(S, D, generator=g), torch.randn(S, D, generator=g), torch.randn(E, D, generator=g) * 0.2
    noise = _gumbel(g, S, E)
    C = R.moe_capacity(S, E, capacity_factor=2.0)
    gates, idx = torch.empty(S, E, device=gpu, dtype=F32), torch.empty(2, S, device=gpu, dtype=torch.int32)
    ops.moe_gate_top2(x.to(gpu), c.to(gpu), wg.to(gpu), noise.to(gpu), gates, idx)
    logits = F.linear(x + c, wg)
    assert rel_l2(gates, F.softmax(logits, 1)) <= 1e-5
    ridx, rslot, rtos, rw = R.routing_top2(gates.cpu(), logits, noise, C)
    assert float((idx.cpu().long() == ridx).float().mean()) >= 0.995
    slot, tos = torch.empty(2, S, device=gpu, dtype=torch.int32), torch.empty(E, C, device=gpu, dtype=torch.int32)
    w, cnt, l_aux = torch.empty(2, S, device=gpu, dtype=F32), torch.empty(E, device=gpu, dtype=torch.int64), torch.empty(1, device=gpu, dtype=F32)
    ops.moe_capacity_top2(gates, idx, C, slot, tos, w, cnt, l_aux)
    yh, yc = torch.randn(E, C, D, generator=g), torch.randn(E, C, D, generator=g)
    o = torch.empty(S, D, device=gpu, dtype=F32)
    ops.moe_combine_topk(yh.to(gpu), yc.to(gpu), w, idx, slot, o, E=E, capacity=C)
    ih, sh, wh = idx.cpu().long(), slot.cpu().long(), w.cpu()
    ref = torch.zeros(S, D)
    for k in range(2):
        kept = sh[k] >= 0
        ref[kept] += wh[k][kept, None] * (yh + yc)[ih[k][kept], sh[k][kept]]
    assert rel_l2(o, ref) <= 1e-6


TINY = dict(num_layers=2, num_single_layers=4, attention_head_dim=128, num_attention_heads=2, joint_attention_dim=64, pooled_projection_dim=64)
CONTROL = dict(use_rope=True, use_shared_expert=True, use_single_trans_blocks=True, single_control_dev=2, single_block_control_method="overall_add",
               top_num=2, expert_num_each_condition=3)


def _counts_close(a, b, S):
    return int((a.cpu() - b).abs().sum()) <= max(2, S // 100)


@pytest.mark.parametrize("cls_name,n_cond,B,grid,T,consis", [("UniGenFlux", 1, 2, 8, 32, False), ("MultiCondtionUniGenFlux", 3, 2, 8, 32, False),
                                                             ("UniGenFlux", 1, 2, 6, 20, True)])
def test_flux_forward_top2_matches_oracle(gpu, cls_name, n_cond, B, grid, T, consis):
    cls = getattr(importlib.import_module("src.UniGenTransformer"), cls_name)
    ctrl = dict(CONTROL, use_consis_module=consis)
    models = {}
    for dt in (BF, F32):
        mm = cls.from_config(dict(TINY), device=gpu, dtype=dt)
        mm.init_condition_block(condition_nums=n_cond, condition_types=["canny", "depth", "openpose"][:n_cond], control_params=dict(ctrl))
        models[dt] = mm
    models[BF].init_synthetic_(seed=11, std=0.05, bias_std=0.02)
    state = {k: v.detach().cpu() for k, v in models[BF].state_dict().items()}
    models[F32].load_state_dict({k: v.to(gpu, F32 if v.dtype == BF else v.dtype) for k, v in state.items()})
    rcfg = R.FluxConfig(condition_nums=n_cond, top_num=2, use_consis_module=consis, **TINY)
    assert set(state) == set(R.state_shapes(rcfg))
    inp = R.make_inputs(rcfg, B=B, grid=grid, T=T, n_cond=n_cond)
    g = torch.Generator().manual_seed(5)
    S = B * grid * grid
    draws = [_gumbel(g, S, rcfg.expert_nums) for _ in range(n_cond)]
    inp["gate_uniform"] = draws[0] if n_cond == 1 else draws          # the gate's random draw: Gumbel(0, 1) for k = 2
    t = torch.full((B,), 0.75, dtype=BF)
    trace = {}
    truth, loss_t, cnt_t = R.unigen_flux_forward(state, rcfg, timestep=t, dtype=F32, trace=trace, **inp)
    ref16, loss16, cnt16 = R.unigen_flux_forward(state, rcfg, timestep=t, dtype=BF, **inp)
    dev = lambda v: [q.to(gpu) for q in v] if isinstance(v, (list, tuple)) else v.to(gpu)
    dinp = {k: dev(v) for k, v in inp.items()}
    assert int((trace["routing"][0]["slot"] < 0).sum()) >= 0
    out32, loss32, outs32 = models[F32](timestep=t.to(gpu), conditioning_scale=1.0, **dinp)
    m = report(f"top2_forward_{cls_name}_c{int(consis)}_f32", out32, truth)
    assert _counts_close(outs32["expert_counts"], cnt_t["expert_counts"], S), (outs32["expert_counts"], cnt_t["expert_counts"])
    assert m["rel_l2"] <= 1e-3, m
    assert abs(float(loss32["moe_loss"]) - float(loss_t["moe_loss"])) <= 1e-4 * abs(float(loss_t["moe_loss"]))
    out, losses, outs = models[BF](timestep=t.to(gpu), conditioning_scale=1.0, **dinp)
    torch.cuda.synchronize()
    err_hip, err_ref = rel_l2(out, truth), rel_l2(ref16, truth)
    m = report(f"top2_forward_{cls_name}_c{int(consis)}_bf16", out, ref16, err_hip_vs_fp32=err_hip, err_oraclebf16_vs_fp32=err_ref)
    assert torch.isfinite(out.float()).all()
    assert err_hip <= 1.25 * err_ref + 1e-3, m
    assert m["rel_l2"] <= 2e-2, m
    assert _counts_close(outs["expert_counts"], cnt16["expert_counts"], S), (outs["expert_counts"], cnt16["expert_counts"])
    assert int(outs["expert_counts"].sum()) == 2 * S                    # every token makes two choices (counted before the capacity drop)
    assert abs(float(losses["moe_loss"]) - float(loss16["moe_loss"])) <= 1e-3 * abs(float(loss16["moe_loss"]))


SD3_TINY = dict(sample_size=16, num_layers=3, attention_head_dim=64, num_attention_heads=2, joint_attention_dim=64, caption_projection_dim=128,
                pooled_projection_dim=64, pos_embed_max_size=12, dual_attention_layers=(0, 1))


@pytest.mark.parametrize("modulated", [False, True])
def test_sd3_forward_top2_matches_oracle(gpu, modulated):
    cls = importlib.import_module("src.UniGenTransformer").UniGenSD3
    B, hw, T = 2, 16, 24
    models = {}
    for dt in (BF, F32):
        mm = cls.from_config(dict(SD3_TINY), device=gpu, dtype=dt)
        mm.init_condition_block(condition_nums=1, condition_types=["depth"], control_params=dict(use_shared_expert=True, use_modulate=modulated, top_num=2))
        models[dt] = mm
    models[BF].init_synthetic_(seed=5, std=0.05, bias_std=0.02)
    state = {k: v.detach().cpu() for k, v in models[BF].state_dict().items()}
    models[F32].load_state_dict({k: v.to(gpu, F32 if v.dtype == BF else v.dtype) for k, v in state.items()})
    rcfg = R.SD3Config(use_modulate=modulated, top_num=2, **SD3_TINY)
    inp = R.make_sd3_inputs(rcfg, B=B, hw=hw, T=T)
    S = inp["gate_uniform"].shape[0]
    inp["gate_uniform"] = _gumbel(torch.Generator().manual_seed(9), S, rcfg.expert_nums)
    t = torch.full((B,), 600.0)
    truth, loss_t, cnt_t = R.unigen_sd3_forward(state, rcfg, timestep=t, dtype=F32, **inp)
    ref16, loss16, cnt16 = R.unigen_sd3_forward(state, rcfg, timestep=t, dtype=BF, **inp)
    dinp = {k: v.to(gpu) for k, v in inp.items()}
    out32, _, outs32 = models[F32](timestep=t.to(gpu), **dinp)
    m = report(f"top2_sd3_mod{int(modulated)}_f32", out32, truth)
    assert _counts_close(outs32["expert_counts"], cnt_t["expert_counts"], S)
    assert m["rel_l2"] <= 1e-3, m
    out, losses, outs = models[BF](timestep=t.to(gpu), **dinp)
    torch.cuda.synchronize()
    err_hip, err_ref = rel_l2(out, truth), rel_l2(ref16, truth)
    m = report(f"top2_sd3_mod{int(modulated)}_bf16", out, ref16, err_hip_vs_fp32=err_hip, err_oraclebf16_vs_fp32=err_ref)
    assert torch.isfinite(out.float()).all() and err_hip <= 1.25 * err_ref + 1e-3 and m["rel_l2"] <= 2.5e-2, m
    assert _counts_close(outs["expert_counts"], cnt16["expert_counts"], S)
    assert abs(float(losses["moe_loss"]) - float(loss16["moe_loss"])) <= 1e-3 * abs(float(loss16["moe_loss"]))


def test_top2_draws_its_own_gumbel_sample(gpu):
    """Without a supplied draw the engine samples Gumbel(0, 1) on the device (deepspeed gumbel_rsample): finite output, two choices per token,
    and a different second choice from run to run."""
    cls = importlib.import_module("src.UniGenTransformer").UniGenFlux
    model = cls.from_config(dict(TINY), device=gpu, dtype=BF)
    model.init_condition_block(condition_nums=1, condition_types=["canny"], control_params=dict(CONTROL))
    model.init_synthetic_(seed=2, std=0.05, bias_std=0.02)
    rcfg = R.FluxConfig(condition_nums=1, top_num=2, **TINY)
    inp = R.make_inputs(rcfg, B=2, grid=8, T=32)
    inp.pop("gate_uniform")
    t = torch.full((2,), 0.5, dtype=BF)
    torch.manual_seed(0)
    o1, _, c1 = model(timestep=t.to(gpu), **{k: v.to(gpu) for k, v in inp.items()})
    i1 = model._w("moe_idx2", (2, 128), torch.int32).clone()
    o2, _, c2 = model(timestep=t.to(gpu), **{k: v.to(gpu) for k, v in inp.items()})
    i2 = model._w("moe_idx2", (2, 128), torch.int32)
    assert torch.isfinite(o1.float()).all() and int(c1["expert_counts"].sum()) == 256
    assert torch.equal(i1[0], i2[0]) and not torch.equal(i1[1], i2[1])


def test_top2_routing_at_full_size_properties(gpu):
    """cfg2's token count (S = 4 x 4096, E = 6, D = 3072) with two choices per token: capacity ceil(2 S / E) = 5462; properties that need no
    oracle run - every kept (token, choice) owns exactly one slot and every occupied slot names the token that owns it, the two choices differ,
    counts = 2 S before the drop, drops only where an expert's queue is full, weights of a token's kept choices sum to 1, bitwise repeatable."""
    from unigen_amd import ops
    S, E, D = 16384, 6, 3072
    g = torch.Generator(device=gpu).manual_seed(4)
    x, c = (torch.randn(S, D, generator=g, device=gpu)).to(BF), (torch.randn(S, D, generator=g, device=gpu)).to(BF)
    wg = (torch.randn(E, D, generator=g, device=gpu) * 0.02).to(BF)
    wg[0] += 0.004                                                  # a favoured expert: its queue overflows
    u = torch.rand(S, E, generator=g, device=gpu).clamp_(1e-7, 1 - 1e-7)
    noise = -torch.log(-torch.log(u))
    C = max(-(-2 * S // E), 4)
    res = []
    for _ in range(2):
        gates, idx = torch.empty(S, E, device=gpu, dtype=F32), torch.empty(2, S, device=gpu, dtype=torch.int32)
        slot, tos = torch.empty(2, S, device=gpu, dtype=torch.int32), torch.empty(E, C, device=gpu, dtype=torch.int32)
        w, cnt, l_aux = torch.empty(2, S, device=gpu, dtype=F32), torch.empty(E, device=gpu, dtype=torch.int64), torch.empty(1, device=gpu, dtype=F32)
        ops.moe_gate_top2(x, c, wg, noise, gates, idx)
        ops.moe_capacity_top2(gates, idx, C, slot, tos, w, cnt, l_aux)
        res.append((gates, idx, slot, tos, w, cnt, l_aux))
    assert all(torch.equal(a, b) for a, b in zip(res[0], res[1])), "not bitwise repeatable"
    gates, idx, slot, tos, w, cnt, l_aux = (t.cpu() for t in res[0])
    idx, slot, tos = idx.long(), slot.long(), tos.long()
    assert C == 5462 and bool((idx[0] != idx[1]).all()) and int(cnt.sum()) == 2 * S
    assert torch.equal(idx[0], gates.argmax(1)) or float((idx[0] == gates.argmax(1)).float().mean()) > 0.9999
    kept = slot >= 0
    flat = (idx * C + slot)[kept]
    assert flat.unique().numel() == flat.numel(), "two (token, choice) pairs share a slot"
    tok = torch.arange(S).expand(2, S)[kept]
    assert torch.equal(tos.view(-1)[flat], tok), "an occupied slot does not name its owner"
    assert int((tos >= 0).sum()) == int(kept.sum())
    for e in range(E):
        n = int(cnt[e])
        assert int((tos[e] >= 0).sum()) == min(n, C)
        assert bool((tos[e, :min(n, C)] >= 0).all())                # slots fill from the front, no holes
    assert int((~kept).sum()) == int((cnt - C).clamp_min(0).sum()) > 0
    ws = (w * kept.float()).sum(0)
    anyk = kept.any(0)
    assert torch.allclose(ws[anyk], torch.ones(int(anyk.sum())), atol=1e-6) and bool((w[~kept] == 0).all())
    assert abs(float(l_aux) - float((gates.mean(0) * torch.bincount(idx[0], minlength=E).float() / S).sum() * E)) < 1e-5
