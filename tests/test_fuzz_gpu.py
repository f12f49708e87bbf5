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


def _gemm_case(rng: random.Random):
    epi = rng.choice(["bias", "gelu", "res_gate", "res_scale", "f32", "split"])
    K = 64 * rng.choice([1, 2, 3, 5, 8, 16, 24])
    M = rng.choice([1, 7, 64, 100, 127, 128, 129, 255, 256, 257, 300, 511, 513, 777, 1024, 1500])
    N = 4 * rng.choice([1, 2, 16, 31, 32, 33, 48, 64, 65, 96, 128, 130, 192, 256])
    groups = rng.choice([1, 1, 1, 2, 3]) if epi not in ("split",) else 1
    rowmap = rng.choice([False, False, True]) and M >= 64 and groups == 1
    alias = epi in ("res_gate", "res_scale") and rng.random() < 0.5
    nobias = rng.random() < 0.2
    return dict(epi=epi, M=M, N=N, K=K, groups=groups, rowmap=rowmap, alias=alias, nobias=nobias)


@pytest.mark.parametrize("seed", range(48))
def test_gemm_random_shapes(gpu, seed):
    from unigen_amd import lib as L, ops
    rng = random.Random(1000 + seed)
    c = _gemm_case(rng)
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
