"""Seeded random-shape sweeps of the two dominant entry points through the C ABI - ug_gemm_bf16 (both tile kernels, every epilogue, row maps,
groups, column split, split-K tail shapes, R aliased to C) and ug_flash_attn_fwd (both head widths, ragged lengths, query offsets, strided
buffers) - against the fp32 formula with the reference's rounding points (torch F.linear / SDPA semantics: src/UniGenTransformer.py:1129,1151,
src/UniGenUtils.py:601). The fixed-shape tests pin the cases the forward uses; this sweep looks for what nobody thought of: tile-edge
arithmetic, tails, strides. Small sizes: the whole file runs in a few seconds."""
import random

import pytest
import torch
import torch.nn.functional as F

from tests.util import rel_l2

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


def _rand(g, *shape, scale=1.0):
    return (torch.randn(*shape, generator=g) * scale).to(BF)


def _gemm_case(rng: random.Random, long_k: bool = False):
    if long_k:
        # round 6: few tiles and a long K loop - what the dispatcher now sends to the 256^2 kernel's split-K tail (every tile cut into K-slices, partial sums
        # through the system-scope slabs): ragged M / N rims, row maps, aliased residuals, the column split, no bias
        epi = rng.choice(["bias", "gelu", "res_gate", "res_scale", "split"])
        return dict(epi=epi, M=rng.choice([192, 200, 256, 300, 512, 640, 777, 1024, 1500]), N=4 * rng.choice([48, 64, 65, 128, 130, 192, 256, 300, 512, 768]),
                    K=64 * rng.choice([96, 100, 128, 160, 200]), groups=1, rowmap=rng.choice([False, False, True]),
                    alias=epi in ("res_gate", "res_scale") and rng.random() < 0.5, nobias=rng.random() < 0.2)
    epi = rng.choice(["bias", "gelu", "res_gate", "res_scale", "f32", "split"])
    K = 64 * rng.choice([1, 2, 3, 5, 8, 16, 24])
    M = rng.choice([1, 7, 64, 100, 127, 128, 129, 255, 256, 257, 300, 511, 513, 777, 1024, 1500])
    N = 4 * rng.choice([1, 2, 16, 31, 32, 33, 48, 64, 65, 96, 128, 130, 192, 256])
    groups = rng.choice([1, 1, 1, 2, 3]) if epi not in ("split",) else 1
    rowmap = rng.choice([False, False, True]) and M >= 64 and groups == 1
    alias = epi in ("res_gate", "res_scale") and rng.random() < 0.5
    nobias = rng.random() < 0.2
    return dict(epi=epi, M=M, N=N, K=K, groups=groups, rowmap=rowmap, alias=alias, nobias=nobias)


@pytest.mark.parametrize("seed", range(64))
def test_gemm_random_shapes(gpu, seed):
    from unigen_amd import lib as L, ops
    rng = random.Random(1000 + seed)
    c = _gemm_case(rng, long_k=seed >= 48)         # seeds 48-63: the sliced small-M regime (K = 6144 ... 12800)
    M, N, K, G, epi = c["M"], c["N"], c["K"], c["groups"], c["epi"]
    g = torch.Generator().manual_seed(seed)
    # A: optionally two batches of rows inside a taller buffer (row map), with a row stride wider than K
    lda = K + 8 * rng.choice([0, 0, 1, 4])
    rpb = M // 2 if c["rowmap"] else 0
    if c["rowmap"]:
        M = 2 * rpb
    bstride = rpb + rng.choice([0, 3, 64]) if c["rowmap"] else 0
    a_rows = (2 * bstride if c["rowmap"] else M)
    a_buf = _rand(g, G, a_rows, lda)
    a = a_buf[:, :, :K]
    w = _rand(g, G, N, K, scale=K ** -0.5)
    b = None if c["nobias"] else _rand(g, G, N, scale=0.1)
    rps = rng.choice([1, 50, 64, 256, max(M, 1)])
    n_s = (M + rps - 1) // rps
    gate = _rand(g, G, n_s, N)
    res = _rand(g, G, M, N)
    shift = 8 * rng.choice([0, 1, 8]) if epi == "split" else 0
    N1 = 0
    if epi == "split":
        N = 256 * rng.choice([2, 3])
        N1 = 256 * rng.choice([1, N // 256 - 1])
        w = _rand(g, G, N, K, scale=K ** -0.5)
        b = _rand(g, G, N, scale=0.1)
        res = _rand(g, G, M, N)
    ldc = N + shift + 4 * rng.choice([0, 0, 2])
    # logical rows of A as the kernel sees them
    if c["rowmap"]:
        a_log = torch.cat([a[:, :rpb], a[:, bstride:bstride + rpb]], 1)
    else:
        a_log = a
    acc = torch.einsum("gmk,gnk->gmn", a_log.float(), w.float())
    v = (acc + (b.float()[:, None] if b is not None else 0)).to(BF)
    if epi == "bias":
        ref = v
    elif epi == "gelu":
        ref = F.gelu(v, approximate="tanh")
    elif epi == "res_gate":
        gg = gate.repeat_interleave(rps, dim=1)[:, :M]
        ref = res + gg * v
    elif epi == "res_scale":
        ref = res + v * 0.7
    elif epi == "f32":
        ref = acc + (b.float()[:, None] if b is not None else 0)
    else:
        ref = torch.cat([v[..., :N1], F.gelu(v[..., N1:], approximate="tanh")], -1)
    dev = lambda t: None if t is None else t.to(gpu)
    out_dt = torch.float32 if epi == "f32" else BF
    out = torch.full((G, M, ldc), 7.0, device=gpu, dtype=out_dt)
    kw = dict(M=M, lda=lda, ldc=ldc, groups=G, a_gstride=a_rows * lda, w_gstride=N * K, bias_gstride=N, c_gstride=M * ldc)
    if c["rowmap"]:
        kw.update(a_map=ops.RowMap(rpb, bstride))
    code = dict(bias=L.EPI_BIAS, gelu=L.EPI_BIAS_GELU, res_gate=L.EPI_RES_GATE, res_scale=L.EPI_RES_SCALE, f32=L.EPI_F32, split=L.EPI_BIAS_GELU)[epi]
    kw.update(epilogue=code)
    if epi == "split":
        kw.update(gelu_from_n=N1, c_shift_from_n=N1, c_shift=shift)
    resd = dev(res)
    if epi in ("res_gate", "res_scale"):
        if c["alias"] and ldc == N:
            out = resd.clone()
            kw.update(residual=out, r_gstride=M * N)
        else:
            kw.update(residual=resd, r_gstride=M * N)
        kw.update(alpha=0.7)
    if epi == "res_gate":
        kw.update(gate=dev(gate), gate_ld=N, rows_per_sample=rps, gate_gstride=n_s * N)
    ops.gemm(dev(a_buf), dev(w), dev(b), out, **kw)        # K, N come from w; lda says how wide A's rows are
    torch.cuda.synchronize()
    got = out[..., :N + shift].cpu()
    if epi == "split":
        assert torch.all(got[..., N1:N1 + shift] == 7.0), "the gap of the column split was written"
        got = torch.cat([got[..., :N1], got[..., N1 + shift:]], -1)
    else:
        got = got[..., :N]
    assert torch.all(out[..., N + shift:].cpu() == 7.0) or (c["alias"] and ldc == N), "columns beyond N were written"
    e = rel_l2(got, ref)
    assert e <= (1e-5 if epi == "f32" else 1e-3), (c, e)


@pytest.mark.parametrize("seed", range(32))
def test_attention_random_shapes(gpu, seed):
    from unigen_amd import ops
    rng = random.Random(2000 + seed)
    dh = rng.choice([128, 128, 64])
    B, H = rng.choice([1, 2, 3]), rng.choice([1, 2, 3])
    Lkv = rng.choice([1, 17, 63, 64, 65, 128, 200, 333, 512, 700, 1023, 1025])
    Lq = rng.choice([1, 5, 31, 32, 33, 100, 256, 257, 300, 520])
    joint = rng.random() < 0.5 and Lq <= Lkv           # queries are rows [qoff, qoff + Lq) of the key buffer itself
    D = H * dh
    W = 3 * D + 8 * rng.choice([0, 0, 2, 64])           # row stride wider than the three projections
    g = torch.Generator().manual_seed(seed)
    kvbuf = _rand(g, B, Lkv, W)
    if joint:
        qoff = rng.randint(0, Lkv - Lq)
        qbuf, q_rows = kvbuf, kvbuf[:, qoff:qoff + Lq, :D]
    else:
        qoff = 0
        qbuf = _rand(g, B, Lq, W)
        q_rows = qbuf[:, :, :D]
    scale = rng.choice([None, 0.05])
    dk, dq = kvbuf.to(gpu), (None if joint else qbuf.to(gpu))
    qd = dk if joint else dq
    out = torch.full((B, Lq, D + 8), 3.0, device=gpu, dtype=BF)
    ops.flash_attn(qd[0, qoff:], dk[0, 0, D:], dk[0, 0, 2 * D:], out, batches=B, heads=H, dh=dh, Lq=Lq, Lkv=Lkv,
                   q_strides=(W, (Lkv if joint else Lq) * W), k_strides=(W, Lkv * W), v_strides=(W, Lkv * W), o_strides=(D + 8, Lq * (D + 8)), scale=scale)
    torch.cuda.synchronize()
    hd = lambda t, L_: t.reshape(B, L_, H, dh).transpose(1, 2).float()
    ref = F.scaled_dot_product_attention(hd(q_rows, Lq), hd(kvbuf[:, :, D:2 * D], Lkv), hd(kvbuf[:, :, 2 * D:3 * D], Lkv), scale=scale)
    ref = ref.transpose(1, 2).reshape(B, Lq, D)
    assert torch.all(out[..., D:].cpu() == 3.0), "columns beyond the heads were written"
    e = rel_l2(out[..., :D], ref)
    assert torch.isfinite(out.float()).all() and e <= 4e-3, (dict(dh=dh, B=B, H=H, Lq=Lq, Lkv=Lkv, joint=joint, qoff=qoff, W=W, scale=scale), e)


@pytest.mark.parametrize("seed", range(24))
def test_moe_routing_random_shapes(gpu, seed):
    """Both gates (deepspeed top1gating with random token selection, top2gating) on random token counts / expert counts / skews, from the device's
    own gate probabilities and choices so that every comparison with the oracle's index forms is exact: slots, token-of-slot, counts, combine
    weights, l_aux; then dispatch + combine against the dense einsums. Includes S < E, S = 1, one expert taking everything, tied RTS draws."""
    from oracle import unigen_ref as R
    from unigen_amd import ops
    rng = random.Random(3000 + seed)
    S = rng.choice([1, 2, 5, 17, 64, 100, 255, 256, 257, 1000, 1023, 3001])
    E = rng.choice([2, 3, 6, 12, 16])
    D = 8 * rng.choice([1, 8, 16, 33])
    top = rng.choice([1, 2])
    g = torch.Generator().manual_seed(seed)
    x, c = _rand(g, S, D), _rand(g, S, D)
    wg = _rand(g, E, D, scale=D ** -0.5)
    mode = rng.choice(["even", "skew", "one"])
    if mode == "skew":
        wg[rng.randrange(E)] *= 3.0
    if mode == "one":
        x = x.abs() + 1; c = c.abs(); wg = wg.abs() * 0.01; wg[E - 1] += 0.5       # every token's first choice is the last expert
    draw = torch.rand(S, E, generator=g)
    if rng.random() < 0.3:
        draw = (draw * 4).floor() / 4                                              # many exactly equal draws: the radix select's tie rule
    gates = torch.empty(S, E, device=gpu, dtype=torch.float32)
    cnt, l_aux = torch.empty(E, device=gpu, dtype=torch.int64), torch.empty(1, device=gpu, dtype=torch.float32)
    info = dict(S=S, E=E, D=D, top=top, mode=mode)
    if top == 1:
        C = R.moe_capacity(S, E)
        idx = torch.empty(S, device=gpu, dtype=torch.int32)
        ops.moe_gate_top1(x.to(gpu), c.to(gpu), wg.to(gpu), gates, idx)
        slot, tos = torch.empty(S, device=gpu, dtype=torch.int32), torch.empty(E, C, device=gpu, dtype=torch.int32)
        ops.moe_capacity_rts(gates, idx, draw.to(gpu), C, slot, tos, cnt, l_aux)
        gh, ih = gates.cpu(), idx.cpu().long()
        # the oracle's routing from the DEVICE's choices: per expert keep the C largest draws (ties: the earlier token), slots in token order
        rslot, rtos = torch.full((S,), -1, dtype=torch.long), torch.full((E, C), -1, dtype=torch.long)
        for e in range(E):
            toks = torch.nonzero(ih == e).flatten()
            if toks.numel() > C:
                keys = draw[toks, e]
                order = sorted(range(toks.numel()), key=lambda i: (-float(keys[i]), i))[:C]
                toks = toks[sorted(order)]
            rslot[toks] = torch.arange(toks.numel()); rtos[e, :toks.numel()] = toks
        assert torch.equal(slot.cpu().long(), rslot), info
        assert torch.equal(tos.cpu().long(), rtos), info
        assert torch.equal(cnt.cpu(), torch.bincount(ih, minlength=E)), info
        w1 = gh[torch.arange(S), ih] * (rslot >= 0).float()
        idx2, slot2, w2 = ih[None], rslot[None], w1[None]
    else:
        C = R.moe_capacity(S, E, capacity_factor=2.0)
        noise = -torch.log(-torch.log(draw.clamp(1e-7, 1 - 1e-7)))
        idx = torch.empty(2, S, device=gpu, dtype=torch.int32)
        ops.moe_gate_top2(x.to(gpu), c.to(gpu), wg.to(gpu), noise.to(gpu), gates, idx)
        slot, tos = torch.empty(2, S, device=gpu, dtype=torch.int32), torch.empty(E, C, device=gpu, dtype=torch.int32)
        w = torch.empty(2, S, device=gpu, dtype=torch.float32)
        ops.moe_capacity_top2(gates, idx, C, slot, tos, w, cnt, l_aux)
        gh, ih = gates.cpu(), idx.cpu().long()
        assert bool((ih[0] != ih[1]).all()), info
        _, rslot, rtos, rw = R.routing_top2(gh, None, None, C, idx=ih)
        assert torch.equal(slot.cpu().long(), rslot) and torch.equal(tos.cpu().long(), rtos) and torch.equal(w.cpu(), rw), info
        assert torch.equal(cnt.cpu(), torch.bincount(ih.flatten(), minlength=E)), info
        idx2, slot2, w2 = ih, rslot, rw
    la = float((gh.mean(0) * torch.bincount(idx2[0], minlength=E).float() / S).sum() * E)
    assert abs(float(l_aux) - la) <= 1e-5 * abs(la) + 1e-7, info
    logits = F.linear((x + c).float(), wg.float())
    assert rel_l2(gh, F.softmax(logits, 1)) <= 1e-5, info
    # dense tensors from the index form; dispatch and combine against the einsums of MOELayer.forward
    K = idx2.shape[0]
    cw = torch.zeros(S, E, C)
    for k in range(K):
        kept = slot2[k] >= 0
        cw[torch.arange(S)[kept], idx2[k][kept], slot2[k][kept]] += w2[k][kept]
    xd = torch.empty(E, C, D, device=gpu, dtype=BF)
    ops.moe_dispatch_modulate(x.to(gpu), None, None, tos, xd, E=E, capacity=C, tokens_per_sample=S)
    ref = torch.zeros(E, C, D, dtype=BF)
    v = rtos >= 0
    ref[v] = x[rtos[v]]
    assert torch.equal(xd.cpu(), ref), info
    yh, yc = _rand(g, E, C, D), _rand(g, E, C, D)
    cwb = cw.to(BF).float()
    want = torch.einsum("sec,ecm->sm", cwb, yh.float()).to(BF) + torch.einsum("sec,ecm->sm", cwb, yc.float()).to(BF)
    out = torch.empty(S, D, device=gpu, dtype=BF)
    if top == 1:
        ops.moe_combine(yh.to(gpu), yc.to(gpu), gates, idx, slot, out, E=E, capacity=C)
    else:
        ops.moe_combine_topk(yh.to(gpu), yc.to(gpu), w, idx, slot, out, E=E, capacity=C)
    assert torch.equal(out.cpu(), want), (info, float((out.cpu().float() - want.float()).abs().max()))


@pytest.mark.parametrize("seed", range(24))
def test_adaln_and_small_linear_random_shapes(gpu, seed):
    """ug_small_linear_bf16 (M <= 64 rows in chunks, SiLU on the input, residual) and ug_adaln_modulate (per-sample / per-token modulation rows,
    strided x, row-mapped x, widths on both sides of the 16-byte fast kernel's conditions) against the formulas with the reference's rounding
    points (AdaLayerNormZero: Linear(SiLU(emb)), LayerNorm without affine, x * (1 + scale) + shift - src/UniGenUtils.py:340-373)."""
    from unigen_amd import ops
    rng = random.Random(4000 + seed)
    g = torch.Generator().manual_seed(seed)
    # small linear
    M, K, N = rng.choice([1, 2, 3, 16, 17, 33, 64]), 8 * rng.choice([1, 4, 32, 96, 100]), 8 * rng.choice([1, 3, 48, 129, 768])
    silu, use_res = rng.random() < 0.5, rng.random() < 0.4
    x, w, b = _rand(g, M, K), _rand(g, N, K, scale=K ** -0.5), _rand(g, N, scale=0.1)
    res = _rand(g, M, N)
    out = torch.empty(M, N, device=gpu, dtype=BF)
    ops.small_linear(x.to(gpu), w.to(gpu), b.to(gpu), out, silu_in=silu, residual=res.to(gpu) if use_res else None)
    xin = F.silu(x) if silu else x
    ref = F.linear(xin.float(), w.float(), b.float()).to(BF)
    if use_res:
        ref = res + ref
    e = rel_l2(out, ref)
    assert e <= 2e-3, (dict(M=M, K=K, N=N, silu=silu, res=use_res), e)
    # AdaLN modulate
    D = rng.choice([64, 128, 200, 256, 1536, 3072, 1544])
    B = rng.choice([1, 2, 3])
    Ls = rng.choice([1, 7, 64, 130])
    per_token = rng.random() < 0.3
    rows = B * Ls
    pad = 8 * rng.choice([0, 0, 2])
    bstride = Ls + rng.choice([0, 0, 5])
    xbuf = _rand(g, B * bstride, D + pad) + _rand(g, B * bstride, 1)
    nmod = rows if per_token else B
    mod = _rand(g, nmod, 3 * D, scale=0.5)
    o = torch.full((rows, D + 8), 5.0, device=gpu, dtype=BF)
    md = mod.to(gpu)
    ops.adaln_modulate(xbuf.to(gpu), md[:, D:], md[:, 2 * D:], o, rows=rows, D=D, rows_per_sample=1 if per_token else Ls, mod_ld=3 * D,
                       x_map=ops.RowMap(Ls, bstride) if bstride != Ls else ops.IDENT)
    xl = torch.cat([xbuf[bi * bstride: bi * bstride + Ls, :D] for bi in range(B)])
    sh, sc = mod[:, D:2 * D], mod[:, 2 * D:]
    if not per_token:
        sh, sc = sh.repeat_interleave(Ls, 0), sc.repeat_interleave(Ls, 0)
    n = F.layer_norm(xl.float(), (D,), eps=1e-6).to(BF)
    want = n * (1 + sc) + sh
    assert torch.all(o[:, D:].cpu() == 5.0)
    e = rel_l2(o[:, :D], want)
    assert e <= 3e-3, (dict(D=D, B=B, Ls=Ls, per_token=per_token, pad=pad, bstride=bstride), e)
