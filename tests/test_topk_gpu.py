"""GPU parity of top-k gating with k > 2 (control_params.top_num > 2 -> MoE(..., k) -> deepspeed 0.16.5 TopKGate -> sharded_moe.topkgating,
/root/reference/src/UniGenTransformer.py:808 -> src/UniGenUtils.py:33-36): the routing kernels through the C ABI against the oracle's
statement-by-statement restatement (oracle/unigen_ref.py topkgating, dense form, and routing_topk, index form), and the UniGenFlux / UniGenSD3
forwards with top_num = 3 against the oracle's, fp32 verification path and bf16 product path. No configuration of the reference uses k > 2 and
deepspeed is not part of /root/reference: parity-unpinned like every other deepspeed-derived row."""
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


@pytest.mark.parametrize("S,E,D,K", [(128, 6, 256, 3), (1000, 12, 128, 5), (4099, 4, 64, 3), (777, 6, 128, 6), (2048, 16, 128, 4)])
def test_topk_routing_dispatch_combine(gpu, S, E, D, K):
    from unigen_amd import ops
    g = torch.Generator().manual_seed(S + E + K)
    x, c = _rand(g, S, D), _rand(g, S, D)
    wg = _rand(g, E, D, scale=0.2)
    wg[0] += 0.05
    C = R.moe_capacity(S, E, capacity_factor=float(K))
    gates, logits_d = torch.empty(S, E, device=gpu, dtype=F32), torch.empty(S, E, device=gpu, dtype=F32)
    idx = torch.empty(K, S, device=gpu, dtype=torch.int32)
    ops.moe_gate_topk(x.to(gpu), c.to(gpu), wg.to(gpu), K, gates, logits_d, idx)
    logits = F.linear((x + c).float(), wg.float())
    assert report(f"moe_topk_gates_S{S}", gates, F.softmax(logits, dim=1))["rel_l2"] <= 1e-5
    assert report(f"moe_topk_logits_S{S}", logits_d, logits)["rel_l2"] <= 1e-5
    # the K choices agree with torch.topk wherever the K-th and (K+1)-th logits (and neighbours) are not a floating-point near-tie
    srt = torch.sort(logits, dim=1, descending=True)
    gaps = (srt[0][:, :-1] - srt[0][:, 1:])[:, :K].min(1)[0] if K < E else (srt[0][:, :-1] - srt[0][:, 1:]).min(1)[0]
    clear = gaps > 1e-4
    assert torch.equal(idx.cpu().long().t()[clear], srt[1][:, :K][clear]) and float(clear.float().mean()) > 0.97
    # from here on the device's own gates, logits and choices are the input: every comparison is exact
    gates_h, logits_h, idx_h = gates.cpu(), logits_d.cpu(), idx.cpu().long()
    ridx, rslot, rtos, rw = R.routing_topk(gates_h, logits_h, K, C, idx=idx_h)
    slot, tos = torch.empty(K, S, device=gpu, dtype=torch.int32), torch.empty(E, C, device=gpu, dtype=torch.int32)
    w, cnt, l_aux = torch.empty(K, S, device=gpu, dtype=F32), torch.empty(E, device=gpu, dtype=torch.int64), torch.empty(1, device=gpu, dtype=F32)
    ops.moe_capacity_topk(gates, logits_d, idx, C, slot, tos, w, cnt, l_aux)
    assert torch.equal(slot.cpu().long(), rslot), "slot assignment differs from the restated topkgating"
    assert torch.equal(tos.cpu().long(), rtos)
    assert torch.equal(cnt.cpu(), torch.stack([(idx_h == e).sum() for e in range(E)])) and int(cnt.sum()) == K * S
    if K < E:
        assert int((rslot < 0).sum()) > 0, "the case should drop some choices"
    assert torch.allclose(w.cpu(), rw, rtol=1e-6, atol=1e-8), float((w.cpu() - rw).abs().max())
    chosen = torch.zeros(S, E).scatter_(1, idx_h.t(), 1.0)
    l_ref = float(torch.mean(gates_h.mean(0) * chosen.mean(0)) * E * E / K)
    assert abs(float(l_aux) - l_ref) <= 1e-5 * abs(l_ref)
    # the dense tensors of topkgating computed from the device's logits describe the same routing (unless a near-tie choice fell the other way)
    l_dense, cw, dm, cnt_ref = R.topkgating(logits_h, K, C)
    cw_idx = torch.zeros(S, E, C)
    for k in range(K):
        kept = rslot[k] >= 0
        cw_idx[torch.arange(S)[kept], ridx[k][kept], rslot[k][kept]] = rw[k][kept]
    if torch.equal(idx_h.t().sort(1)[0], torch.topk(logits_h, K, dim=1)[1].sort(1)[0]):
        assert torch.equal(cw_idx.bool(), dm) and torch.allclose(cw_idx, cw, rtol=1e-5, atol=1e-7)
        assert torch.equal(cnt.cpu(), cnt_ref) and abs(float(l_aux) - float(l_dense)) <= 1e-5 * abs(float(l_dense))
    dm = cw_idx.bool()
    # dispatch and combine on the shared kernels
    B = 1 if S % 2 else 2
    N = S // B
    mod = _rand(g, E, B, D)
    out = torch.empty(E, C, D, device=gpu, dtype=BF)
    ops.moe_dispatch_modulate(x.to(gpu), None, mod.to(gpu), tos, out, E=E, capacity=C, tokens_per_sample=N, mod_estride=B * D, mod_bstride=D)
    xd = torch.einsum("sec,sm->ecm", dm.to(BF).float(), x.float())
    samp = torch.where(rtos >= 0, rtos // N, torch.zeros_like(rtos))
    ref = (mod.float()[torch.arange(E)[:, None], samp] * xd).to(BF)
    assert report(f"moe_topk_dispatch_S{S}", out, ref)["mismatch_frac"] == 0.0
    yh, yc, xs, cs = _rand(g, E, C, D), _rand(g, E, C, D), _rand(g, S, D), _rand(g, S, D)
    o = torch.empty(S, D, device=gpu, dtype=BF)
    ops.moe_combine_topk(yh.to(gpu), yc.to(gpu), w, idx, slot, o, E=E, capacity=C, xs=xs.to(gpu), cs=cs.to(gpu))
    # einsum("sec,ecm->sm") in bf16: fp32 sum over the kept choices IN CHOICE ORDER of bf16(weight) * y, one rounding; then the CoMoE residual sums
    wb = w.cpu().to(BF).float()
    eh, ec = torch.zeros(S, D), torch.zeros(S, D)
    for k in range(K):
        kept = rslot[k] >= 0
        rows = (ridx[k] * C + rslot[k].clamp_min(0))
        eh += torch.where(kept[:, None], wb[k][:, None] * yh.float().view(E * C, D)[rows], torch.zeros(()))
        ec += torch.where(kept[:, None], wb[k][:, None] * yc.float().view(E * C, D)[rows], torch.zeros(()))
    ref = (xs + eh.to(BF)) + (cs + ec.to(BF))
    m = report(f"moe_topk_combine_S{S}", o, ref)
    assert m["rel_l2"] <= 2e-3 and m["mismatch_frac"] <= 0.02, m          # fma vs mul + add inside the fp32 sum moves the odd last bit


def test_topk_fp32_twin_and_repeatability(gpu):
    from unigen_amd import ops
    S, E, D, K = 600, 6, 64, 3
    g = torch.Generator().manual_seed(3)
    x, c, wg = torch.randn(S, D, generator=g), torch.randn(S, D, generator=g), torch.randn(E, D, generator=g) * 0.2
    C = R.moe_capacity(S, E, capacity_factor=float(K))
    res = []
    for _ in range(2):
        gates, logits, idx = torch.empty(S, E, device=gpu, dtype=F32), torch.empty(S, E, device=gpu, dtype=F32), torch.empty(K, S, device=gpu, dtype=torch.int32)
        ops.moe_gate_topk(x.to(gpu), c.to(gpu), wg.to(gpu), K, gates, logits, idx)
        slot, tos = torch.empty(K, S, device=gpu, dtype=torch.int32), torch.empty(E, C, device=gpu, dtype=torch.int32)
        w, cnt, l_aux = torch.empty(K, S, device=gpu, dtype=F32), torch.empty(E, device=gpu, dtype=torch.int64), torch.empty(1, device=gpu, dtype=F32)
        ops.moe_capacity_topk(gates, logits, idx, C, slot, tos, w, cnt, l_aux)
        res.append((gates, logits, idx, slot, tos, w, cnt, l_aux))
    assert all(torch.equal(a, b) for a, b in zip(res[0], res[1])), "not bitwise repeatable"
    gates, logits, idx, slot, tos, w, cnt, l_aux = res[0]
    ref_logits = F.linear(x + c, wg)
    assert rel_l2(logits, ref_logits) <= 1e-5 and rel_l2(gates, F.softmax(ref_logits, 1)) <= 1e-5
    ridx, rslot, rtos, rw = R.routing_topk(gates.cpu(), logits.cpu(), K, C, idx=idx.cpu().long())
    assert torch.equal(slot.cpu().long(), rslot) and torch.equal(tos.cpu().long(), rtos)
    yh, yc = torch.randn(E, C, D, generator=g), torch.randn(E, C, D, generator=g)
    o = torch.empty(S, D, device=gpu, dtype=F32)
    ops.moe_combine_topk(yh.to(gpu), yc.to(gpu), w, idx, slot, o, E=E, capacity=C)
    ih, sh, wh = idx.cpu().long(), slot.cpu().long(), w.cpu()
    ref = torch.zeros(S, D)
    for k in range(K):
        kept = sh[k] >= 0
        ref[kept] += wh[k][kept, None] * (yh + yc)[ih[k][kept], sh[k][kept]]
    assert rel_l2(o, ref) <= 1e-6


TINY = dict(num_layers=2, num_single_layers=4, attention_head_dim=128, num_attention_heads=2, joint_attention_dim=64, pooled_projection_dim=64)
CONTROL = dict(use_rope=True, use_shared_expert=True, use_single_trans_blocks=True, single_control_dev=2, single_block_control_method="overall_add",
               top_num=3, expert_num_each_condition=3)


def _counts_close(a, b, S):
    return int((a.cpu() - b).abs().sum()) <= max(2, S // 100)


@pytest.mark.parametrize("cls_name,n_cond", [("UniGenFlux", 1), ("MultiCondtionUniGenFlux", 2)])
def test_flux_forward_top3_matches_oracle(gpu, cls_name, n_cond):
    cls = getattr(importlib.import_module("src.UniGenTransformer"), cls_name)
    B, grid, T = 2, 8, 32
    models = {}
    for dt in (BF, F32):
        mm = cls.from_config(dict(TINY), device=gpu, dtype=dt)
        mm.init_condition_block(condition_nums=n_cond, condition_types=["canny", "depth"][:n_cond], control_params=dict(CONTROL))
        models[dt] = mm
    models[BF].init_synthetic_(seed=13, std=0.05, bias_std=0.02)
    state = {k: v.detach().cpu() for k, v in models[BF].state_dict().items()}
    models[F32].load_state_dict({k: v.to(gpu, F32 if v.dtype == BF else v.dtype) for k, v in state.items()})
    rcfg = R.FluxConfig(condition_nums=n_cond, top_num=3, **TINY)
    inp = R.make_inputs(rcfg, B=B, grid=grid, T=T, n_cond=n_cond)
    inp["gate_uniform"] = None if n_cond == 1 else [None] * n_cond            # topkgating takes no random draw
    t = torch.full((B,), 0.75, dtype=BF)
    S = B * grid * grid
    truth, loss_t, cnt_t = R.unigen_flux_forward(state, rcfg, timestep=t, dtype=F32, **inp)
    ref16, loss16, cnt16 = R.unigen_flux_forward(state, rcfg, timestep=t, dtype=BF, **inp)
    dev = lambda v: v if v is None else ([None if q is None else q.to(gpu) for q in v] if isinstance(v, (list, tuple)) else v.to(gpu))
    dinp = {k: dev(v) for k, v in inp.items()}
    out32, loss32, outs32 = models[F32](timestep=t.to(gpu), conditioning_scale=1.0, **dinp)
    m = report(f"top3_forward_{cls_name}_f32", out32, truth)
    assert _counts_close(outs32["expert_counts"], cnt_t["expert_counts"], S) and m["rel_l2"] <= 1e-3, m
    assert abs(float(loss32["moe_loss"]) - float(loss_t["moe_loss"])) <= 1e-4 * abs(float(loss_t["moe_loss"]))
    out, losses, outs = models[BF](timestep=t.to(gpu), conditioning_scale=1.0, **dinp)
    torch.cuda.synchronize()
    err_hip, err_ref = rel_l2(out, truth), rel_l2(ref16, truth)
    m = report(f"top3_forward_{cls_name}_bf16", out, ref16, err_hip_vs_fp32=err_hip, err_oraclebf16_vs_fp32=err_ref)
    assert torch.isfinite(out.float()).all() and err_hip <= 1.25 * err_ref + 1e-3 and m["rel_l2"] <= 2e-2, m
    assert int(outs["expert_counts"].sum()) == 3 * S and _counts_close(outs["expert_counts"], cnt16["expert_counts"], S)
    assert abs(float(losses["moe_loss"]) - float(loss16["moe_loss"])) <= 1e-3 * abs(float(loss16["moe_loss"]))


SD3_TINY = dict(sample_size=16, num_layers=3, attention_head_dim=64, num_attention_heads=2, joint_attention_dim=64, caption_projection_dim=128,
                pooled_projection_dim=64, pos_embed_max_size=12, dual_attention_layers=(0, 1))


def test_sd3_forward_top3_matches_oracle(gpu):
    cls = importlib.import_module("src.UniGenTransformer").UniGenSD3
    B, hw, T = 2, 16, 24
    models = {}
    for dt in (BF, F32):
        mm = cls.from_config(dict(SD3_TINY), device=gpu, dtype=dt)
        mm.init_condition_block(condition_nums=1, condition_types=["depth"], control_params=dict(use_shared_expert=True, top_num=3))
        models[dt] = mm
    models[BF].init_synthetic_(seed=6, std=0.05, bias_std=0.02)
    state = {k: v.detach().cpu() for k, v in models[BF].state_dict().items()}
    models[F32].load_state_dict({k: v.to(gpu, F32 if v.dtype == BF else v.dtype) for k, v in state.items()})
    rcfg = R.SD3Config(top_num=3, **SD3_TINY)
    inp = R.make_sd3_inputs(rcfg, B=B, hw=hw, T=T)
    S = inp["gate_uniform"].shape[0]
    inp["gate_uniform"] = None
    t = torch.full((B,), 600.0)
    truth, loss_t, cnt_t = R.unigen_sd3_forward(state, rcfg, timestep=t, dtype=F32, **inp)
    ref16, loss16, cnt16 = R.unigen_sd3_forward(state, rcfg, timestep=t, dtype=BF, **inp)
    dinp = {k: (None if v is None else v.to(gpu)) for k, v in inp.items()}
    out32, _, outs32 = models[F32](timestep=t.to(gpu), **dinp)
    m = report("top3_sd3_f32", out32, truth)
    assert _counts_close(outs32["expert_counts"], cnt_t["expert_counts"], S) and m["rel_l2"] <= 1e-3, m
    out, losses, outs = models[BF](timestep=t.to(gpu), **dinp)
    torch.cuda.synchronize()
    err_hip, err_ref = rel_l2(out, truth), rel_l2(ref16, truth)
    m = report("top3_sd3_bf16", out, ref16, err_hip_vs_fp32=err_hip, err_oraclebf16_vs_fp32=err_ref)
    assert torch.isfinite(out.float()).all() and err_hip <= 1.25 * err_ref + 1e-3 and m["rel_l2"] <= 2.5e-2, m
    assert _counts_close(outs["expert_counts"], cnt16["expert_counts"], S)
