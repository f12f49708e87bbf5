"""GPU parity tests, kernel level: every C-ABI entry point of libunigen_hip.so against the CPU oracle (oracle/unigen_ref.py
primitives, i.e. the reference's torch semantics) on identical seeded bf16 inputs.

Stated tolerance (north star: <= 1e-3 vs reference): relative L2 error <= 1e-3 against the oracle evaluated with the
reference's own rounding points; attention is allowed 4e-3 because P is quantised to bf16 for the P.V MFMA exactly as
the reference's SDPA flash kernels do, with a different (but equally valid) tile order. Integer routing outputs are exact.
"""
import math

import pytest
import torch
import torch.nn.functional as F

from oracle import unigen_ref as R
from tests.util import bf, rel_l2, report

pytestmark = pytest.mark.gpu
TOL = 1e-3
BF = torch.bfloat16


def _rand(g, *shape, scale=1.0):
    return (torch.randn(*shape, generator=g) * scale).to(BF)


def _epi_ref(acc, bias, epi, res=None, gate=None, rows_per_sample=1, alpha=1.0):
    """acc fp32 [M,N] (fp32 accumulate of bf16 products) -> reference rounding points of Linear + following torch ops."""
    v = (acc + (bias.float() if bias is not None else 0)).to(BF)
    if epi == "bias":
        return v
    if epi == "gelu":
        return F.gelu(v, approximate="tanh")
    if epi == "res_gate":
        g = gate.repeat_interleave(rows_per_sample, dim=0)[: res.shape[0]]
        return res + g * v
    if epi == "res_scale":
        return res + v * alpha
    raise ValueError(epi)


@pytest.mark.parametrize("M,N,K", [(256, 256, 128), (300, 260, 64), (128, 64, 192), (1, 8, 64), (513, 384, 1024)])
@pytest.mark.parametrize("epi", ["bias", "gelu", "res_gate", "res_scale", "f32"])
def test_gemm_epilogues(gpu, M, N, K, epi):
    from unigen_amd import lib as L, ops
    g = torch.Generator().manual_seed(M * 7 + N * 3 + K)
    a, w, b = _rand(g, M, K), _rand(g, N, K, scale=K ** -0.5), _rand(g, N, scale=0.1)
    res = _rand(g, M, N)
    rps = 50 if M >= 100 else 1
    nsamp = (M + rps - 1) // rps
    gate = _rand(g, nsamp, N)
    acc = a.float() @ w.float().t()
    ad, wd, bd, resd, gated = (t.to(gpu) for t in (a, w, b, res, gate))
    if epi == "f32":
        out = torch.empty(M, N, device=gpu, dtype=torch.float32)
        ops.gemm(ad, wd, bd, out, M=M, epilogue=L.EPI_F32)
        ref = acc + b.float()
        m = report(f"gemm_f32_{M}x{N}x{K}", out, ref)
        assert m["rel_l2"] <= 1e-5
        return
    out = torch.empty(M, N, device=gpu, dtype=BF)
    code = dict(bias=L.EPI_BIAS, gelu=L.EPI_BIAS_GELU, res_gate=L.EPI_RES_GATE, res_scale=L.EPI_RES_SCALE)[epi]
    ops.gemm(ad, wd, bd, out, M=M, epilogue=code, residual=resd if epi.startswith("res") else None,
             gate=gated if epi == "res_gate" else None, gate_ld=N, rows_per_sample=rps, alpha=0.75)
    ref = _epi_ref(acc, b, epi, res, gate, rps, 0.75)[:M]
    m = report(f"gemm_{epi}_{M}x{N}x{K}", out, ref)
    assert m["rel_l2"] <= TOL, m


@pytest.mark.parametrize("M,N1,N2,K,force", [(300, 512, 256, 128, 0), (520, 256, 512, 192, 128), (256, 768, 1024, 64, 256)])
def test_gemm_column_split(gpu, M, N1, N2, K, force):
    """One launch over two concatenated Linear layers (single block: [to_q; to_k; to_v; proj_mlp]): bias-only columns [0, N1), GELU columns
    [N1, N1 + N2) stored behind a gap of `shift` columns; both kernels (UG_GEMM_FORCE_TILE is read once, so the shapes pick the kernel)."""
    from unigen_amd import lib as L, ops
    g = torch.Generator().manual_seed(M + N1 + K)
    N, shift = N1 + N2, 64
    a, w, b = _rand(g, M, K), _rand(g, N, K, scale=K ** -0.5), _rand(g, N, scale=0.1)
    out = torch.full((M, N + shift), 7.0, device=gpu, dtype=BF)
    ops.gemm(a.to(gpu), w.to(gpu), b.to(gpu), out, M=M, epilogue=L.EPI_BIAS_GELU, ldc=N + shift, gelu_from_n=N1, c_shift_from_n=N1, c_shift=shift)
    v = (a.float() @ w.float().t() + b.float()).to(BF).float()
    ref = torch.full((M, N + shift), 7.0)
    ref[:, :N1] = v[:, :N1]
    ref[:, N1 + shift:] = F.gelu(v[:, N1:], approximate="tanh")
    m = report(f"gemm_split_{M}x{N1}+{N2}x{K}", out, ref.to(BF))
    assert m["rel_l2"] <= TOL, m
    assert torch.equal(out[:, N1:N1 + shift].cpu(), torch.full((M, shift), 7.0, dtype=BF)), "the gap was written"
    with pytest.raises(Exception):
        ops.gemm(a.to(gpu), w.to(gpu), b.to(gpu), out, M=M, epilogue=L.EPI_BIAS_GELU, ldc=N + shift, gelu_from_n=N1 + 8)


def test_gemm_rowmaps_inplace_grouped_lora(gpu):
    from unigen_amd import lib as L, ops
    g = torch.Generator().manual_seed(5)
    # A is the image slice of a [B, T+N, K] buffer, C the image slice of a [B, T+N, N] buffer, residual in place
    B, T, Nn, K, No = 3, 24, 40, 128, 192
    Lt = T + Nn
    abuf, cbuf = _rand(g, B, Lt, K), _rand(g, B, Lt, No)
    w, b = _rand(g, No, K, scale=K ** -0.5), _rand(g, No, scale=0.1)
    ad, cd = abuf.to(gpu), cbuf.to(gpu)
    ops.gemm(ad[0, T:], w.to(gpu), b.to(gpu), cd[0, T:], M=B * Nn, epilogue=L.EPI_RES_SCALE, lda=K, ldc=No,
             a_map=ops.RowMap(Nn, Lt), c_map=ops.RowMap(Nn, Lt), residual=cd[0, T:], ldr=No, r_map=ops.RowMap(Nn, Lt), alpha=0.5)
    ref = cbuf.clone()
    acc = abuf[:, T:].float() @ w.float().t()
    ref[:, T:] = cbuf[:, T:] + (acc + b.float()).to(BF) * 0.5
    m = report("gemm_rowmap_inplace", cd, ref)
    assert m["rel_l2"] <= TOL and torch.equal(cd[:, :T].cpu(), cbuf[:, :T]), m
    # grouped (experts): E groups of C rows each
    E, Cc, D = 5, 70, 128
    xa, we, be = _rand(g, E, Cc, D), _rand(g, E, D, D, scale=D ** -0.5), _rand(g, E, D, scale=0.1)
    out = torch.empty(E, Cc, D, device=gpu, dtype=BF)
    ops.gemm(xa.to(gpu), we.to(gpu), be.to(gpu), out, M=Cc, groups=E, a_gstride=Cc * D, w_gstride=D * D, bias_gstride=D, c_gstride=Cc * D)
    ref = (torch.einsum("ecd,eod->eco", xa.float(), we.float()) + be.float()[:, None]).to(BF)
    m = report("gemm_grouped", out, ref)
    assert m["rel_l2"] <= TOL, m
    # LoRA epilogue: y = x W^T + b + scale * (x A^T) B^T, rank padded to 64
    M, K, N, r = 200, 128, 256, 16
    x, w, b = _rand(g, M, K), _rand(g, N, K, scale=K ** -0.5), _rand(g, N, scale=0.1)
    A, Bm, sc = _rand(g, r, K, scale=K ** -0.5), _rand(g, N, r, scale=0.3), 2.0
    Ap = torch.zeros(64, K, dtype=BF); Ap[:r] = A
    Bp = torch.zeros(N, 64, dtype=BF); Bp[:, :r] = (Bm.float() * sc).to(BF)
    t = torch.empty(M, 64, device=gpu, dtype=BF)
    ops.gemm(x.to(gpu), Ap.to(gpu), None, t, M=M)
    out = torch.empty(M, N, device=gpu, dtype=BF)
    ops.gemm(x.to(gpu), w.to(gpu), b.to(gpu), out, M=M, lora_t=t, lora_b=Bp.to(gpu))
    ref = R.lora_linear(x.float(), w.float(), b.float(), [(A.float(), Bm.float(), sc)])
    m = report("gemm_lora", out, ref)
    assert m["rel_l2"] <= 3e-3, m   # T = x A^T is rounded to bf16 between the two products (peft does the same in bf16)


def test_gemm_a_row_maps_short_batches_and_broadcast(gpu):
    """the 256^2 kernel's buffer-form DMA offsets are unsigned distances from a tile's first row. (a) rows-per-batch below the tile
    height (a 256-row tile spans several batch jumps) on a shape the dispatcher gives to the 256^2 kernel (160 tiles) must read the right rows;
    (b) a NON-monotonic map (batch stride 0 = one batch broadcast to all) on the same shape class must leave that kernel instead of wrapping its
    offsets; (c) the fused q/k epilogue, which only exists in that kernel, refuses such a map."""
    from unigen_amd import lib as L, ops
    g = torch.Generator().manual_seed(9)
    K, N = 256, 2048
    w, b = _rand(g, N, K, scale=K ** -0.5), _rand(g, N, scale=0.1)
    wd, bd = w.to(gpu), b.to(gpu)
    # (a) 128 batches of 40 rows inside a [128, 72, K] buffer: M = 5120 = 20 row tiles, 6-7 batch jumps of 32 rows inside every tile
    Bn, rpb, Lt = 128, 40, 72
    abuf = _rand(g, Bn, Lt, K)
    ref = ((abuf[:, 32:].float().reshape(Bn * rpb, K) @ w.float().t()) + b.float()).to(BF)
    out = torch.empty(Bn * rpb, N, device=gpu, dtype=BF)
    ops.gemm(abuf.to(gpu)[0, 32:], wd, bd, out, M=Bn * rpb, lda=K, a_map=ops.RowMap(rpb, Lt))
    m = report("gemm_rowmap_short_batches", out, ref)
    assert m["rel_l2"] <= TOL, m
    # (b) one 320-row batch read by 16 logical batches (batch stride 0): M = 5120 again
    rows = 320
    a1 = _rand(g, rows, K)
    ref = ((a1.float() @ w.float().t()) + b.float()).to(BF).repeat(16, 1)
    out = torch.empty(16 * rows, N, device=gpu, dtype=BF)
    ops.gemm(a1.to(gpu), wd, bd, out, M=16 * rows, lda=K, a_map=ops.RowMap(rows, 0))
    m = report("gemm_rowmap_broadcast", out, ref)
    assert m["rel_l2"] <= TOL, m
    # (c)
    wq = torch.ones(128, device=gpu, dtype=BF)
    cs = torch.zeros(512, 64, 2, device=gpu); cs[..., 0] = 1.0
    with pytest.raises(L.UniGenHipError, match="monotonic"):
        ops.gemm(a1.to(gpu), wd, bd, torch.empty(512, N, device=gpu, dtype=BF), M=512, lda=K, a_map=ops.RowMap(256, 0),
                 qk_rope=ops.QkRope(wq, wq, cs, 0, 0, 256, dh=128))


def test_gemm_rejects_bad_args(gpu):
    from unigen_amd import lib as L, ops
    a = torch.zeros(8, 72, device=gpu, dtype=BF)
    w = torch.zeros(8, 72, device=gpu, dtype=BF)
    with pytest.raises(L.UniGenHipError, match="multiple of 64"):
        ops.gemm(a, w, None, torch.empty(8, 8, device=gpu, dtype=BF), M=8)
    with pytest.raises(L.UniGenHipError):
        ops.gemm(a.cpu(), w, None, torch.empty(8, 8, device=gpu, dtype=BF), M=8)


@pytest.mark.parametrize("M,N,K,silu,resid", [(4, 768, 256, False, False), (2, 1536, 256, True, True), (8, 100, 64, True, False), (16, 264, 3072, True, True)])
def test_small_linear(gpu, M, N, K, silu, resid):
    from unigen_amd import ops
    g = torch.Generator().manual_seed(M + N + K)
    x, w, b, r = _rand(g, M, K), _rand(g, N, K, scale=K ** -0.5), _rand(g, N, scale=0.1), _rand(g, M, N)
    out = torch.empty(M, N, device=gpu, dtype=BF)
    ops.small_linear(x.to(gpu), w.to(gpu), b.to(gpu), out, silu_in=silu, residual=r.to(gpu) if resid else None)
    xin = F.silu(x) if silu else x
    ref = F.linear(xin.float(), w.float(), b.float()).to(BF)
    if resid:
        ref = r + ref
    m = report(f"small_linear_{M}x{N}x{K}", out, ref)
    assert m["rel_l2"] <= TOL, m


@pytest.mark.parametrize("B,rows,D", [(2, 37, 256), (3, 64, 3072), (1, 5, 1536)])
def test_adaln_modulate(gpu, B, rows, D):
    from unigen_amd import ops
    g = torch.Generator().manual_seed(D + rows)
    x = _rand(g, B, rows, D, scale=2.0) + 0.5
    emb = _rand(g, B, 6 * D, scale=0.5)
    out = torch.empty(B * rows, D, device=gpu, dtype=BF)
    embd = emb.to(gpu)
    ops.adaln_modulate(x.to(gpu).view(B * rows, D), embd[:, 3 * D:], embd[:, 4 * D:], out, rows=B * rows, D=D, rows_per_sample=rows, mod_ld=6 * D)
    shift, scale = emb[:, 3 * D:4 * D], emb[:, 4 * D:5 * D]
    ref = R.layer_norm(x) * (1 + scale[:, None]) + shift[:, None]
    m = report(f"adaln_{B}x{rows}x{D}", out.view(B, rows, D), ref)
    assert m["rel_l2"] <= TOL, m


@pytest.mark.parametrize("D", [1536, 3072, 4096])
def test_adaln_modulate_fast_kernel_matches_generic(gpu, D, monkeypatch):
    """Round 3: the model widths take a branch-free kernel (all loads in flight at once, packed row, DPP reductions). Against the generic kernel
    (UG_ADALN_FAST=0) on a row-mapped joint buffer (x rows at a batch stride, as the double blocks read them) and a ragged last block of
    rows: same formula and rounding points, only the association of the fp32 row sums differs -> at most a rare one-bf16-step difference."""
    from unigen_amd import ops
    from unigen_amd.ops import RowMap
    monkeypatch.setenv("UG_ENV_DYNAMIC", "1")
    g = torch.Generator().manual_seed(D)
    B, rows, stride = 3, 37, 50                                   # 111 rows: the last block of 4 is ragged; physical row = b * 50 + r
    xbuf = (_rand(g, B * stride, D, scale=2.0) + 0.5).to(gpu)
    emb = _rand(g, B, 6 * D, scale=0.5).to(gpu)
    outs = []
    for fast in ("1", "0"):
        monkeypatch.setenv("UG_ADALN_FAST", fast)
        out = torch.full((B * rows, D), 7.0, device=gpu, dtype=BF)
        ops.adaln_modulate(xbuf, emb[:, 3 * D:], emb[:, 4 * D:], out, rows=B * rows, D=D, rows_per_sample=rows, mod_ld=6 * D, x_map=RowMap(rows, stride))
        outs.append(out.float().cpu())
    x = xbuf.view(B, stride, D)[:, :rows].cpu()
    e = emb.cpu()
    ref = R.layer_norm(x) * (1 + e[:, 4 * D:5 * D][:, None]) + e[:, 3 * D:4 * D][:, None]
    m = report(f"adaln_fast_{D}", outs[0].view(B, rows, D), ref)
    mism = float((outs[0] != outs[1]).float().mean())
    assert m["rel_l2"] <= TOL and mism <= 1e-3 and float((outs[0] - outs[1]).abs().max()) <= 0.07, (m, mism)


@pytest.mark.parametrize("dh,H", [(128, 2), (64, 3)])
def test_qk_rmsnorm_rope(gpu, dh, H):
    from unigen_amd import ops
    g = torch.Generator().manual_seed(dh)
    B, T, N = 2, 5, 12
    Lt, D = T + N, H * dh
    buf = _rand(g, B, Lt, 3 * D)
    wq_a, wk_a, wq_b, wk_b = (1 + _rand(g, dh, scale=0.1) for _ in range(4))
    ids = torch.cat([torch.zeros(T, 3), R.make_ids(3, 4, torch.float32)], 0).to(BF)
    axes = (16, 56, 56) if dh == 128 else (8, 28, 28)
    cos, sin = R.flux_pos_embed(ids, axes)
    d = buf.to(gpu)
    ops.qk_rmsnorm_rope(d, batches=B, rows_per_batch=Lt, ld=3 * D, q_off=0, k_off=D, heads=H, dh=dh, wq_a=wq_a.to(gpu), wk_a=wk_a.to(gpu),
                        wq_b=wq_b.to(gpu), wk_b=wk_b.to(gpu), split=T, cos=cos.to(gpu), sin=sin.to(gpu))
    ref = buf.clone()
    for off, wa, wb in ((0, wq_a, wq_b), (D, wk_a, wk_b)):
        t = buf[:, :, off:off + D].view(B, Lt, H, dh).transpose(1, 2)
        t = torch.cat([R.rms_norm(t[:, :, :T], wa), R.rms_norm(t[:, :, T:], wb)], 2)
        t = R.apply_rotary_emb(t, cos, sin)
        ref[:, :, off:off + D] = t.transpose(1, 2).reshape(B, Lt, D)
    m = report(f"qk_rmsnorm_rope_dh{dh}", d, ref)
    assert m["rel_l2"] <= TOL and torch.equal(d[:, :, 2 * D:].cpu(), buf[:, :, 2 * D:]), m
    # k-only on a row sub-range (control blocks: text K/V cached, image rows re-done), no q
    d2 = buf.to(gpu)
    ops.qk_rmsnorm_rope(d2[0, T:], batches=B, rows_per_batch=N, batch_stride_rows=Lt, pos_offset=T, ld=3 * D, q_off=-1, k_off=D, heads=H, dh=dh,
                        wk_a=wk_a.to(gpu), wk_b=wk_b.to(gpu), split=T, cos=cos.to(gpu), sin=sin.to(gpu))
    ref2 = buf.clone()
    ref2[:, T:, D:2 * D] = ref[:, T:, D:2 * D]
    m = report(f"qk_rmsnorm_rope_sub_dh{dh}", d2, ref2)
    assert m["rel_l2"] <= TOL, m


@pytest.mark.parametrize("B,H,Lq,Lkv,qoff", [(1, 2, 256, 256, 0), (2, 3, 300, 333, 0), (1, 2, 64, 200, 136), (1, 1, 512, 1024, 0),
                                             (1, 2, 300, 2100, 1700), (2, 1, 2049, 2049, 0), (1, 3, 70, 2048, 0)])
def test_flash_attn(gpu, B, H, Lq, Lkv, qoff):
    """qoff > 0: queries are rows [qoff, qoff+Lq) of the joint sequence (image-only queries of a control joint block). The last three cases: long
    key sequences with ragged query / key counts (a one-key last tile, fewer queries than one workgroup owns, queries starting deep in the buffer)."""
    from unigen_amd import ops
    dh = 128
    D = H * dh
    g = torch.Generator().manual_seed(Lq + Lkv)
    qkv = _rand(g, B, Lkv, 3 * D)
    d = qkv.to(gpu)
    out = torch.zeros(B, Lq, D, device=gpu, dtype=BF)
    ops.flash_attn(d[0, qoff:], d[0, 0, D:], d[0, 0, 2 * D:], out, batches=B, heads=H, dh=dh, Lq=Lq, Lkv=Lkv,
                   q_strides=(3 * D, Lkv * 3 * D), k_strides=(3 * D, Lkv * 3 * D), v_strides=(3 * D, Lkv * 3 * D), o_strides=(D, Lq * D))
    q = qkv[:, qoff:qoff + Lq, :D].view(B, Lq, H, dh).transpose(1, 2).float()
    k = qkv[:, :, D:2 * D].view(B, Lkv, H, dh).transpose(1, 2).float()
    v = qkv[:, :, 2 * D:].view(B, Lkv, H, dh).transpose(1, 2).float()
    ref = F.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(B, Lq, D)
    m = report(f"flash_attn_B{B}H{H}_{Lq}x{Lkv}", out, ref)
    assert m["rel_l2"] <= 4e-3, m


@pytest.mark.parametrize("Lkv", [320, 2304])
def test_flash_attn_rescale_branch(gpu, Lkv):
    """Force the online-softmax rescale: one key row far larger than the rest in a LATE tile (cdna guide rule 26); Lkv = 2304 adds spikes in tiles
    14 and 23 for queries of one wave (40), and for two queries of ANOTHER wave in the same tile (100, 120: rows of both 16-row halves)."""
    from unigen_amd import ops
    B, H, Lq, dh = 1, 1, 256, 128
    g = torch.Generator().manual_seed(3)
    q, k, v = _rand(g, B, Lq, dh), _rand(g, B, Lkv, dh, scale=0.3), _rand(g, B, Lkv, dh)
    k[0, 200] = q[0, 17] * 4.0     # spikes the score of query 17 (and correlates with others) in tile 3
    k[0, Lkv - 20] = q[0, 99] * 6.0     # and again in the last tile
    if Lkv > 1000:
        k[0, 900] = q[0, 40] * 5.0; k[0, 1500] = (q[0, 100] + q[0, 120]) * 4.0
    out = torch.zeros(B, Lq, dh, device=gpu, dtype=BF)
    ops.flash_attn(q.to(gpu), k.to(gpu), v.to(gpu), out, batches=B, heads=H, dh=dh, Lq=Lq, Lkv=Lkv, q_strides=(dh, Lq * dh),
                   k_strides=(dh, Lkv * dh), v_strides=(dh, Lkv * dh), o_strides=(dh, Lq * dh))
    ref = F.scaled_dot_product_attention(q.float()[:, None], k.float()[:, None], v.float()[:, None])[:, 0]
    m = report(f"flash_attn_rescale_{Lkv}", out, ref)
    assert m["rel_l2"] <= 4e-3, m


def test_timestep_euler_add(gpu):
    from unigen_amd import ops
    t = torch.tensor([1000.0, 752.0, 500.0, 250.0, 0.0])
    out = torch.empty(5, 256, device=gpu, dtype=BF)
    ops.timestep_embed(t.to(gpu), out)
    ref = R.timestep_sinusoid(t).to(BF)
    m = report("timestep_embed", out, ref)
    assert m["max_rel"] <= 8e-3, m          # bf16 ulp of values in [0.5, 1]; sin/cos of ~1e3 rad differ in the last fp32 bits
    g = torch.Generator().manual_seed(0)
    x, v = _rand(g, 4, 64, 64), _rand(g, 4, 64, 64)
    xd = x.to(gpu).clone()
    ops.euler_step(xd, v.to(gpu), -0.25)
    ref = R.euler_step(x, v, 1.0, 0.75)
    m = report("euler_step", xd, ref)
    assert m["mismatch_frac"] == 0.0, m
    # a step that is no power of two (SD3's shifted 28-step schedule): the scheduler's `step * model_output` is a bf16 op (0-dim fp32 times bf16: both operands
    # cast to bf16, result rounded) before the fp32 add - a good part of the elements differ by one bf16 ulp from the unrounded form; bit-equal to the oracle
    from unigen_amd.pipeline import _step32
    s0, s1 = 0.9873806, 0.9741077
    xd = x.to(gpu).clone()
    ops.euler_step(xd, v.to(gpu), _step32(s0, s1))
    ref = R.euler_step(x, v, s0, s1)
    assert report("euler_step_rounded_product", xd, ref)["mismatch_frac"] == 0.0
    unrounded = (x.float() + (s1 - s0) * v.float()).to(BF)
    assert 0.0 < float((ref != unrounded).float().mean()) < 0.5
    x32, v32 = x.float().to(gpu), v.float().to(gpu)                       # the fp32 twin: nothing to round
    ops.euler_step(x32, v32, _step32(s0, s1))
    assert torch.equal(x32.cpu(), R.euler_step(x.float(), v.float(), s0, s1))
    o = torch.empty(4, 64, 64, device=gpu, dtype=BF)
    ops.add(x.to(gpu), v.to(gpu), o)
    assert torch.equal(o.cpu(), x + v)


@pytest.mark.parametrize("S,E,D", [(128, 6, 256), (1000, 12, 128), (4099, 6, 64)])
def test_moe_routing_dispatch_combine(gpu, S, E, D):
    from unigen_amd import ops
    g = torch.Generator().manual_seed(S + E)
    B = 1 if S % 2 else 2
    N = S // B
    x, c = _rand(g, S, D), _rand(g, S, D)
    wg = _rand(g, E, D, scale=0.2)
    uni = torch.rand(S, E, generator=g)
    C = R.moe_capacity(S, E)
    gates = torch.empty(S, E, device=gpu, dtype=torch.float32)
    idx = torch.empty(S, device=gpu, dtype=torch.int32)
    ops.moe_gate_top1(x.to(gpu), c.to(gpu), wg.to(gpu), gates, idx)
    logits = F.linear((x + c).float(), wg.float())
    ref_gates = F.softmax(logits, dim=1)
    m = report(f"moe_gates_S{S}", gates, ref_gates)
    assert m["rel_l2"] <= 1e-5, m
    # argmax must agree wherever the top-2 gap is not a floating-point tie
    top2 = torch.topk(ref_gates, 2, dim=1)[0]
    clear = (top2[:, 0] - top2[:, 1]) > 1e-5
    assert torch.equal(idx.cpu().long()[clear], ref_gates.argmax(1)[clear])
    # routing from the DEVICE gates (so that later comparisons are exact index comparisons)
    gates_h = gates.cpu()
    ridx, rslot, rtos = R.routing_from_gates(gates_h, uni, C)
    l_aux_ref, cw, dm, cnt_ref = R.top1gating(torch.log(gates_h), uni, C)
    slot = torch.empty(S, device=gpu, dtype=torch.int32)
    tos = torch.empty(E, C, device=gpu, dtype=torch.int32)
    cnt = torch.empty(E, device=gpu, dtype=torch.int64)
    l_aux = torch.empty(1, device=gpu, dtype=torch.float32)
    ops.moe_capacity_rts(gates, idx, uni.to(gpu), C, slot, tos, cnt, l_aux)
    assert torch.equal(idx.cpu().long(), ridx)
    assert torch.equal(slot.cpu().long(), rslot), "slot assignment differs from deepspeed top1gating"
    assert torch.equal(tos.cpu().long(), rtos)
    assert torch.equal(cnt.cpu(), cnt_ref)
    assert abs(float(l_aux) - float(l_aux_ref)) <= 1e-5 * abs(float(l_aux_ref))
    # the dense combine_weights of top1gating say the same thing as (idx, slot)
    s_ar = torch.arange(S)
    kept = rslot >= 0
    assert torch.equal(dm.sum((1, 2)).bool(), kept)
    assert torch.allclose(cw[s_ar[kept], ridx[kept], rslot[kept]], gates_h[s_ar[kept], ridx[kept]])
    # dispatch + modulate, with and without the added expert output
    mod = _rand(g, E, B, D)
    add_ = _rand(g, E, C, D)
    for use_add in (False, True):
        out = torch.empty(E, C, D, device=gpu, dtype=BF)
        ops.moe_dispatch_modulate(x.to(gpu), add_.to(gpu) if use_add else None, mod.to(gpu), tos, out, E=E, capacity=C, tokens_per_sample=N,
                                  mod_estride=B * D, mod_bstride=D)
        ref = torch.zeros(E, C, D, dtype=BF)
        for e in range(E):
            for s_ in range(C):
                t = int(rtos[e, s_])
                if t >= 0:
                    xin = x[t] + add_[e, s_] if use_add else x[t]
                    ref[e, s_] = (mod[e, t // N].float() * xin.float()).to(BF)
        m = report(f"moe_dispatch_S{S}_add{int(use_add)}", out, ref)
        assert m["mismatch_frac"] == 0.0, m
    # combine
    yh, yc, xs, cs = _rand(g, E, C, D), _rand(g, E, C, D), _rand(g, S, D), _rand(g, S, D)
    cw2 = torch.zeros(S, E, C)
    cw2[s_ar[kept], ridx[kept], rslot[kept]] = gates_h[s_ar[kept], ridx[kept]]   # combine_weights built from the device gates
    cwb = cw2.to(BF)
    eh = torch.einsum("sec,ecm->sm", cwb.float(), yh.float()).to(BF)
    ec = torch.einsum("sec,ecm->sm", cwb.float(), yc.float()).to(BF)
    out = torch.empty(S, D, device=gpu, dtype=BF)
    ops.moe_combine(yh.to(gpu), yc.to(gpu), gates, idx, slot, out, E=E, capacity=C, xs=xs.to(gpu), cs=cs.to(gpu))
    ref = (xs + eh) + (cs + ec)
    m = report(f"moe_combine_S{S}", out, ref)
    assert m["mismatch_frac"] == 0.0, m
    ops.moe_combine(yh.to(gpu), yc.to(gpu), gates, idx, slot, out, E=E, capacity=C, accumulate=True)
    ref2 = ref + (eh + ec)
    m = report(f"moe_combine_acc_S{S}", out, ref2)
    assert m["mismatch_frac"] == 0.0, m
    # shared-expert streams as the two halves of one [B][2N][D] buffer (row map), all samples in one launch
    xc = torch.cat([xs.view(B, N, D), cs.view(B, N, D)], 1).contiguous().to(gpu).view(B * 2 * N, D)
    ops.moe_combine(yh.to(gpu), yc.to(gpu), gates, idx, slot, out, E=E, capacity=C, xs=xc, cs=xc[N:], s_map=ops.RowMap(N, 2 * N))
    m = report(f"moe_combine_rowmap_S{S}", out, ref)
    assert m["mismatch_frac"] == 0.0, m


@pytest.mark.parametrize("B,H,Lq,Lkv", [(1, 2, 256, 256), (2, 3, 300, 333), (6, 2, 86, 86), (1, 24, 1024, 1357)])
def test_flash_attn_head_dim_64(gpu, B, H, Lq, Lkv):
    """SD3.5 heads (dh = 64); (6, 2, 86, 86) is the shape of the expert self-attention over capacity slots."""
    from unigen_amd import ops
    dh = 64
    D = H * dh
    g = torch.Generator().manual_seed(Lq * 3 + Lkv)
    qkv = _rand(g, B, Lkv, 3 * D)
    d = qkv.to(gpu)
    out = torch.zeros(B, Lq, D, device=gpu, dtype=BF)
    st = (3 * D, Lkv * 3 * D)
    ops.flash_attn(d[0, Lkv - Lq:], d[0, 0, D:], d[0, 0, 2 * D:], out, batches=B, heads=H, dh=dh, Lq=Lq, Lkv=Lkv, q_strides=st, k_strides=st,
                   v_strides=st, o_strides=(D, Lq * D))
    q = qkv[:, Lkv - Lq:, :D].view(B, Lq, H, dh).transpose(1, 2).float()
    k = qkv[:, :, D:2 * D].view(B, Lkv, H, dh).transpose(1, 2).float()
    v = qkv[:, :, 2 * D:].view(B, Lkv, H, dh).transpose(1, 2).float()
    ref = F.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(B, Lq, D)
    m = report(f"flash_attn64_B{B}H{H}_{Lq}x{Lkv}", out, ref)
    assert m["rel_l2"] <= 4e-3, m


def test_grouped_residual_gate_gather_rowbcast_plain_dispatch(gpu):
    from unigen_amd import lib as L, ops
    g = torch.Generator().manual_seed(9)
    # grouped GEMM with a per-ROW gate and residual per group (per-token AdaLN gates of the SD3 expert blocks)
    E, C, D = 4, 70, 128
    x, w, b = _rand(g, E, C, D), _rand(g, E, D, D, scale=D ** -0.5), _rand(g, E, D, scale=0.1)
    res, emb = _rand(g, E, C, D), _rand(g, E * C, 6 * D, scale=0.5)
    out = res.to(gpu).clone()
    embd = emb.to(gpu)
    ops.gemm(x.to(gpu), w.to(gpu), b.to(gpu), out, M=C, epilogue=L.EPI_RES_GATE, groups=E, a_gstride=C * D, w_gstride=D * D, bias_gstride=D,
             c_gstride=C * D, residual=out, r_gstride=C * D, gate=embd[:, 2 * D:], gate_ld=6 * D, rows_per_sample=1, gate_gstride=C * 6 * D)
    v = (torch.einsum("ecd,eod->eco", x.float(), w.float()) + b.float()[:, None]).to(BF)
    ref = res + emb[:, 2 * D:3 * D].view(E, C, D) * v
    m = report("gemm_grouped_res_gate_per_row", out, ref)
    assert m["rel_l2"] <= TOL, m
    # row gather with empty rows
    src = _rand(g, 11, 64)
    idx = torch.tensor([3, -1, 10, 0, 3, -1, 7], dtype=torch.int32)
    o = torch.empty(7, 64, device=gpu, dtype=BF)
    ops.gather_rows(src.to(gpu), idx.to(gpu), o)
    ref = torch.where((idx >= 0)[:, None], src[idx.clamp_min(0).long()], torch.zeros(1, 64, dtype=BF))
    assert torch.equal(o.cpu(), ref)
    # fp32 row-broadcast add (PatchEmbed pos table)
    xx, tab = _rand(g, 3 * 10, 64), torch.randn(10, 64, generator=g)
    xd = xx.to(gpu).clone()
    ops.add_rowbcast_f32(xd, tab.to(gpu), 10)
    ref = (xx.float().view(3, 10, 64) + tab[None]).to(BF).view(30, 64)
    assert torch.equal(xd.cpu(), ref)
    # plain dispatch (mod = None)
    tos = torch.tensor([[2, 5, -1], [0, -1, -1]], dtype=torch.int32)
    xs = _rand(g, 6, 64)
    o = torch.empty(2, 3, 64, device=gpu, dtype=BF)
    ops.moe_dispatch_modulate(xs.to(gpu), None, None, tos.to(gpu), o, E=2, capacity=3, tokens_per_sample=6)
    ref = torch.zeros(2, 3, 64, dtype=BF); ref[0, 0], ref[0, 1], ref[1, 0] = xs[2], xs[5], xs[0]
    assert torch.equal(o.cpu(), ref)


def test_lora_linear_switch(gpu):
    """LoRALinear forward = one fused GEMM (base + adapter K-segment); enable_lora switches adapters per condition (src/lora_switching_module.py:4-38,
    PEFT 0.15 `y = base(x) + sum_a B_a(A_a(x)) * alpha_a / r_a`). Tolerance as everywhere: the fp32 verification twin <= 1e-3 against the fp32 formula
    (measured ~1e-6); the bf16 product kernel <= 4e-3 against the formula evaluated with PEFT's own bf16 rounding points and NO further from the
    fp32 result than that evaluation (it rounds once, after accumulating base and adapters in fp32, where PEFT rounds every op)."""
    import importlib
    mod = importlib.import_module("src.lora_switching_module")
    g = torch.Generator().manual_seed(4)
    K, N, M = 128, 192, 100
    w, b = _rand(g, N, K, scale=K ** -0.5), _rand(g, N, scale=0.1)
    ads = {"canny": (_rand(g, 8, K, scale=K ** -0.5), _rand(g, N, 8, scale=0.3), 8, 16.0), "depth": (_rand(g, 4, K, scale=K ** -0.5), _rand(g, N, 4, scale=0.3), 4, 4.0)}
    x = _rand(g, 2, M // 2, K)
    for dt in (torch.float32, BF):
        lin = mod.LoRALinear(K, N, device=gpu, dtype=dt)
        lin.weight.data.copy_(w); lin.bias.data.copy_(b)
        for name, (A, Bm, r, alpha) in ads.items():
            lin.add_adapter(name, r=r, lora_alpha=alpha, A=A, B=Bm)
        for active in (["canny", "depth"], ["depth"]):
            spec = [(ads[n][0], ads[n][1], ads[n][3] / ads[n][2]) for n in active]
            truth = R.lora_linear(x.float(), w.float(), b.float(), [(A.float(), Bm.float(), sc) for A, Bm, sc in spec])
            if len(active) == 2:
                got = lin(x.to(gpu, dt))
            else:
                with mod.enable_lora([lin], active):
                    got = lin(x.to(gpu, dt))
                # the reference restores through set_scale (src/lora_switching_module.py:36-38), i.e. multiplies the saved scaling by alpha / r AGAIN:
                # idempotent only when alpha == r ("depth" here); mirrored, not fixed
                assert lin.scaling["canny"] == (16.0 / 8) ** 2 and lin.scaling["depth"] == 1.0
            tag = "f32" if dt == torch.float32 else "bf16"
            if dt == torch.float32:
                m = report(f"lora_linear_{'+'.join(active)}_{tag}", got, truth)
                assert m["rel_l2"] <= 1e-5, m
            else:
                ref16 = R.lora_linear(x, w, b, spec)                # PEFT's bf16 eager rounding points
                m = report(f"lora_linear_{'+'.join(active)}_{tag}", got, ref16, err_hip_vs_fp32=rel_l2(got, truth), err_ref16_vs_fp32=rel_l2(ref16, truth))
                assert m["rel_l2"] <= 4e-3 and m["err_hip_vs_fp32"] <= 1.05 * m["err_ref16_vs_fp32"], m


@pytest.mark.parametrize("M,N,K,epi", [(2304, 3072, 6144, "res_gate"), (4608, 4096, 8192, "gelu"), (2000, 2900 // 4 * 4, 6208, "bias"),
                                       (512, 3072, 12288, "res_gate"), (1024, 3072, 12288, "res_gate"), (1536, 3072, 15360, "res_gate"), (1000, 3072, 12288, "bias")])
def test_gemm_splitk_tail(gpu, M, N, K, epi):
    """Shapes whose 256x256 tiles leave a partially filled last round: the remainder tiles are split along K, partial sums go through
    the fp32 workspace and the last arriver reduces them in fixed slice order -> correct AND bitwise repeatable. The last four are the batch-1 / 512^2
    forms of ff.net.2 and of the single blocks' proj_out (24-72 tiles, K = 12288 / 15360; round 6): fewer tiles than CUs, every tile cut into K-slices."""
    from unigen_amd import lib as L, ops
    g = torch.Generator().manual_seed(M + N + K)
    a, w, b = _rand(g, M, K), _rand(g, N, K, scale=K ** -0.5), _rand(g, N, scale=0.1)
    res, gate = _rand(g, M, N), _rand(g, (M + 255) // 256, N)
    code = dict(bias=L.EPI_BIAS, gelu=L.EPI_BIAS_GELU, res_gate=L.EPI_RES_GATE)[epi]
    outs = []
    for _ in range(3):
        out = torch.empty(M, N, device=gpu, dtype=BF)
        ops.gemm(a.to(gpu), w.to(gpu), b.to(gpu), out, M=M, epilogue=code, residual=res.to(gpu) if epi == "res_gate" else None,
                 gate=gate.to(gpu) if epi == "res_gate" else None, gate_ld=N, rows_per_sample=256)
        outs.append(out.cpu())
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2]), "split-K tail is not bitwise repeatable"
    acc = a.float() @ w.float().t()
    ref = _epi_ref(acc, b, epi, res, gate, 256, 1.0)[:M]
    m = report(f"gemm_splitk_{epi}_{M}x{N}x{K}", outs[0], ref)
    assert m["rel_l2"] <= TOL, m


@pytest.mark.parametrize("M,N,K,path", [(100, 192, 128, "128^2 kernel, ragged M"), (130, 64, 64, "128^2 kernel, N below one tile"),
                                        (8192, 4096, 256, "256^2 kernel, whole tiles: fast epilogue, 8 row-groups of residual prefetched"),
                                        (8000, 4100, 192, "256^2 kernel, ragged M / N rims: slow edge epilogue"),
                                        (2304, 3072, 6144, "256^2 kernel, split-K tail: the last arriver's epilogue")])
@pytest.mark.parametrize("epi", ["res_gate", "res_scale"])
def test_gemm_residual_aliased_to_output_every_dispatch_path(gpu, M, N, K, path, epi):
    """include/unigen_hip.h: `R` may alias `C` (the engine's in-place `x = x + gate * f(x)` / zero-res launches, and LinearN.backward's dX
    accumulation). Every dispatch path must read a residual element before the store that overwrites it - pinned here per path: aliased == the
    out-of-place result, bitwise."""
    from unigen_amd import lib as L, ops
    g = torch.Generator().manual_seed(M + N + K)
    a, w, b = _rand(g, M, K).to(gpu), _rand(g, N, K, scale=K ** -0.5).to(gpu), _rand(g, N, scale=0.1).to(gpu)
    res, gate = _rand(g, M, N).to(gpu), _rand(g, (M + 255) // 256, N).to(gpu)
    code = L.EPI_RES_GATE if epi == "res_gate" else L.EPI_RES_SCALE
    kw = dict(M=M, epilogue=code, alpha=0.7)
    if epi == "res_gate":
        kw.update(gate=gate, gate_ld=N, rows_per_sample=256)
    out = torch.empty(M, N, device=gpu, dtype=BF)
    ops.gemm(a, w, b, out, residual=res, **kw)
    inplace = res.clone()
    ops.gemm(a, w, b, inplace, residual=inplace, **kw)
    assert torch.equal(inplace, out), path
    ref = _epi_ref(a.cpu().float() @ w.cpu().float().t(), b.cpu(), epi, res.cpu(), gate.cpu(), 256, 0.7)
    assert report(f"gemm_alias_{epi}_{M}x{N}x{K}", inplace, ref)["rel_l2"] <= TOL


def test_gemm_lora_segment_in_256_kernel_matches_128(gpu):
    """The LoRA K-segment (T . B^T appended to the K loop, src/lora_switching_module.py:11-38) runs in BOTH tile kernels; same MFMA shape and
    K order -> bit-identical. Shapes where the dispatcher picks the 256^2 kernel, ragged M, 1-2 LoRA K-tiles, K = 128 (shortest) .. 320."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r"""
import os, sys
os.environ["UG_ENV_DYNAMIC"] = "1"
sys.path.insert(0, %r)
import torch
from unigen_amd import ops, lib as L
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
rn = lambda *s: torch.randn(*s, generator=g, device=dev).to(torch.bfloat16)
bad = 0
for (M, N, K, r, epi) in [(4000, 4096, 128, 64, L.EPI_BIAS), (4096, 4096, 192, 128, L.EPI_BIAS_GELU), (4090, 4096, 320, 64, L.EPI_RES_GATE),
                          (4096, 3840, 256, 192, L.EPI_RES_SCALE), (8192, 2048, 1024, 64, L.EPI_BIAS)]:
    a, w, b = rn(M, K), rn(N, K) * 0.1, rn(N)
    t, lb = rn(M, r), rn(N, r) * 0.1
    res, gate = rn(M, N), rn((M + 999) // 1000, N)
    outs = []
    for tile in ("128", "256", "0"):
        os.environ["UG_GEMM_FORCE_TILE"] = tile
        out = torch.zeros(M, N, device=dev, dtype=torch.bfloat16)
        kw = dict(M=M, epilogue=epi, lora_t=t, lora_b=lb)
        if epi in (L.EPI_RES_GATE, L.EPI_RES_SCALE):
            kw.update(residual=res, alpha=0.7)
        if epi == L.EPI_RES_GATE:
            kw.update(gate=gate, gate_ld=N, rows_per_sample=1000)
        ops.gemm(a, w, b, out, **kw)
        outs.append(out)
    torch.cuda.synchronize()
    ref = (a.float() @ w.float().t() + t.float() @ lb.float().t() + b.float())
    if epi == L.EPI_BIAS:
        e = float((outs[1].float() - ref).norm() / ref.norm())
        if e > 4e-3:
            bad += 1; print("REF MISMATCH", M, N, K, r, e)
    for o in outs[1:]:
        if not torch.equal(o, outs[0]):
            bad += 1
            print("MISMATCH", M, N, K, r, epi, float((o.float() - outs[0].float()).abs().max()))
sys.exit(1 if bad else 0)
""" % root
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout + p.stderr


def test_gemm_fused_qk_rmsnorm_rope_epilogue(gpu):
    """UG_EPI_QKV_ROPE = ug_gemm_bf16 + ug_qk_rmsnorm_rope in one launch (Attention.norm_q/k + apply_rotary_emb, src/UniGenUtils.py:561-599):
    the single block's [q | k | v | gelu(mlp)] launch with the column shift, and a double block's sample rows written through a row map into
    the joint buffer at position offset Lc. v / mlp columns must be bit-identical to the two-launch path; q / k columns may differ by the
    fp32 summation order of the row's sum of squares (a last-bit effect on rstd): one bf16 step on a handful of elements, and as close to
    the fp32 restatement as the two-launch path."""
    from unigen_amd import lib as L, ops
    g = torch.Generator().manual_seed(11)
    D, dh = 512, 128
    H = D // dh

    def tables(S):
        ang = torch.rand(S, dh // 2, generator=g) * 6.28
        cos, sin = ang.cos().repeat_interleave(2, 1).contiguous(), ang.sin().repeat_interleave(2, 1).contiguous()
        cs = torch.stack([cos[:, 0::2], sin[:, 0::2]], -1).contiguous()
        return cos.to(gpu), sin.to(gpu), cs.to(gpu)

    def ref_qk(y, wn, cos, sin):          # y [rows, H*dh] fp32 of bf16 values, positions = rows
        x = y.view(y.shape[0], H, dh)
        rs = torch.rsqrt((x * x).mean(-1, keepdim=True) + 1e-6)
        x = ((x * rs).to(BF).float() * wn.float()).to(BF).float()
        xr = torch.stack([-x[..., 1::2], x[..., 0::2]], -1).flatten(-2)
        return (x * cos[:, None] + xr * sin[:, None]).reshape(y.shape[0], H * dh)

    wq, wk = (1 + 0.2 * torch.randn(dh, generator=g)).to(BF), (1 + 0.2 * torch.randn(dh, generator=g)).to(BF)
    # --- single block: M = B * Lj rows, N = 7D, GELU from 3D on, stored behind the attention slot --------------------------------------
    B, Lj, K = 2, 1280, 256
    M = B * Lj
    x, w, b = _rand(g, M, K), _rand(g, 7 * D, K, scale=K ** -0.5), _rand(g, 7 * D, scale=0.1)
    cos, sin, cs = tables(Lj)
    xd, wd, bd, wqd, wkd = x.to(gpu), w.to(gpu), b.to(gpu), wq.to(gpu), wk.to(gpu)
    two = torch.zeros(M, 8 * D, device=gpu, dtype=BF)
    ops.gemm(xd, wd, bd, two, M=M, ldc=8 * D, epilogue=L.EPI_BIAS_GELU, gelu_from_n=3 * D, c_shift_from_n=3 * D, c_shift=D)
    lin = two.clone()
    ops.qk_rmsnorm_rope(two, batches=B, rows_per_batch=Lj, ld=8 * D, q_off=0, k_off=D, heads=H, dh=dh, wq_b=wqd, wk_b=wkd, split=0, cos=cos, sin=sin)
    one = torch.zeros(M, 8 * D, device=gpu, dtype=BF)
    ops.gemm(xd, wd, bd, one, M=M, ldc=8 * D, epilogue=L.EPI_BIAS_GELU, gelu_from_n=3 * D, c_shift_from_n=3 * D, c_shift=D,
             qk_rope=ops.QkRope(wqd, wkd, cs, Lj, 0, 2 * D))
    torch.cuda.synchronize()
    assert torch.equal(one[:, 2 * D:], two[:, 2 * D:]), "v / mlp columns differ"
    pos = torch.arange(M, device=gpu) % Lj
    for name, c0, wn in (("q", 0, wqd), ("k", D, wkd)):
        ref = ref_qk(lin[:, c0:c0 + D].float(), wn, cos[pos], sin[pos])
        e1 = float((one[:, c0:c0 + D].float() - ref).norm() / ref.norm())
        e2 = float((two[:, c0:c0 + D].float() - ref).norm() / ref.norm())
        diff = (one[:, c0:c0 + D].float() - two[:, c0:c0 + D].float()).abs()
        frac = float((diff > 0).float().mean())
        ulp = float(diff.max())            # values are O(1): one bf16 step of the normalised value (a rotated sum may land near zero)
        print(f"fused qk-rope single {name}: rel_l2 fused {e1:.3e} two-launch {e2:.3e}; differing elements {frac:.2e}, max abs diff {ulp:.3e}")
        assert e1 <= 3e-3 and e1 <= 1.05 * e2 + 1e-5 and frac < 2e-2 and ulp <= 0.04, (name, e1, e2, frac, ulp)
    # --- double block sample rows: row map into the joint buffer [B, Lc + Ls, 3D], positions Lc + m % Ls ------------------------------
    B, Lc, Ls = 3, 128, 768
    Ljn = Lc + Ls
    M = B * Ls
    x, w, b = _rand(g, M, K), _rand(g, 3 * D, K, scale=K ** -0.5), _rand(g, 3 * D, scale=0.1)
    cos, sin, cs = tables(Ljn)
    xd, wd, bd = x.to(gpu), w.to(gpu), b.to(gpu)
    junk = _rand(g, B * Ljn, 3 * D).to(gpu)
    two, one = junk.clone(), junk.clone()
    ops.gemm(xd, wd, bd, two[Lc:], M=M, ldc=3 * D, c_map=ops.RowMap(Ls, Ljn))
    ops.qk_rmsnorm_rope(two[Lc:], batches=B, rows_per_batch=Ls, batch_stride_rows=Ljn, pos_offset=Lc, ld=3 * D, q_off=0, k_off=D, heads=H, dh=dh,
                        wq_b=wqd, wk_b=wkd, split=0, cos=cos, sin=sin)
    ops.gemm(xd, wd, bd, one[Lc:], M=M, ldc=3 * D, c_map=ops.RowMap(Ls, Ljn), qk_rope=ops.QkRope(wqd, wkd, cs, Ls, Lc, 2 * D))
    torch.cuda.synchronize()
    o3, t3 = one.view(B, Ljn, 3 * D), two.view(B, Ljn, 3 * D)
    assert torch.equal(o3[:, :Lc], junk.view(B, Ljn, 3 * D)[:, :Lc]) and torch.equal(o3[:, Lc:, 2 * D:], t3[:, Lc:, 2 * D:])
    diff = (o3[:, Lc:, :2 * D].float() - t3[:, Lc:, :2 * D].float()).abs()
    rel = float(diff.norm() / t3[:, Lc:, :2 * D].float().norm())
    frac = float((diff > 0).float().mean())
    print(f"fused qk-rope double: rel_l2 vs two-launch {rel:.3e}, differing elements {frac:.2e}")
    assert rel <= 5e-4 and frac < 2e-2, (rel, frac)
    # --- rows per sample not a multiple of the tile: 256-row tiles straddle two samples (one position wrap inside a tile) ---------------------
    B, Ls2 = 4, 320
    M = B * Ls2
    x, w, b = _rand(g, M, K), _rand(g, 3 * D, K, scale=K ** -0.5), _rand(g, 3 * D, scale=0.1)
    cos, sin, cs = tables(Ls2 + 7)
    xd, wd, bd = x.to(gpu), w.to(gpu), b.to(gpu)
    two, one = torch.zeros(M, 3 * D, device=gpu, dtype=BF), torch.zeros(M, 3 * D, device=gpu, dtype=BF)
    ops.gemm(xd, wd, bd, two, M=M)
    ops.qk_rmsnorm_rope(two, batches=B, rows_per_batch=Ls2, pos_offset=7, ld=3 * D, q_off=0, k_off=D, heads=H, dh=dh, wq_b=wqd, wk_b=wkd, split=0, cos=cos, sin=sin)
    ops.gemm(xd, wd, bd, one, M=M, qk_rope=ops.QkRope(wqd, wkd, cs, Ls2, 7, 2 * D))
    torch.cuda.synchronize()
    diff = (one.float() - two.float()).abs()
    print(f"fused qk-rope straddling tiles: rel_l2 vs two-launch {float(diff.norm() / two.float().norm()):.3e}, differing {float((diff > 0).float().mean()):.2e}")
    assert float(diff.norm() / two.float().norm()) <= 5e-4 and float((diff > 0).float().mean()) < 2e-2
    # --- head width 64 (SD3.5): four heads per tile, RMSNorm only (no RoPE), and with a table ---------------------------------------------
    dh4, H4 = 64, D // 64
    wq4, wk4 = (1 + 0.2 * torch.randn(dh4, generator=g)).to(BF).to(gpu), (1 + 0.2 * torch.randn(dh4, generator=g)).to(BF).to(gpu)
    B, Ls4 = 2, 512
    M = B * Ls4
    x, w, b = _rand(g, M, K), _rand(g, 3 * D, K, scale=K ** -0.5), _rand(g, 3 * D, scale=0.1)
    xd, wd, bd = x.to(gpu), w.to(gpu), b.to(gpu)
    ang = torch.rand(Ls4, dh4 // 2, generator=g) * 6.28
    cos4, sin4 = ang.cos().repeat_interleave(2, 1).contiguous().to(gpu), ang.sin().repeat_interleave(2, 1).contiguous().to(gpu)
    cs4 = torch.stack([cos4[:, 0::2], sin4[:, 0::2]], -1).contiguous()
    for tab in (None, (cos4, sin4, cs4)):
        two, one = torch.zeros(M, 3 * D, device=gpu, dtype=BF), torch.zeros(M, 3 * D, device=gpu, dtype=BF)
        ops.gemm(xd, wd, bd, two, M=M)
        ops.qk_rmsnorm_rope(two, batches=B, rows_per_batch=Ls4, ld=3 * D, q_off=0, k_off=D, heads=H4, dh=dh4, wq_b=wq4, wk_b=wk4, split=0,
                            cos=tab[0] if tab else None, sin=tab[1] if tab else None)
        ops.gemm(xd, wd, bd, one, M=M, qk_rope=ops.QkRope(wq4, wk4, tab[2] if tab else None, Ls4, 0, 2 * D, dh=dh4))
        torch.cuda.synchronize()
        diff = (one.float() - two.float()).abs()
        rel, frac = float(diff.norm() / two.float().norm()), float((diff > 0).float().mean())
        print(f"fused qk-norm dh64 rope={tab is not None}: rel_l2 vs two-launch {rel:.3e}, differing {frac:.2e}")
        assert torch.equal(one[:, 2 * D:], two[:, 2 * D:]) and rel <= 5e-4 and frac < 2e-2, (rel, frac)
    # --- argument checks ----------------------------------------------------------------------------------------------------------------
    with pytest.raises(L.UniGenHipError, match="multiples of 256"):
        ops.gemm(xd[:300], wd, bd, torch.empty(300, 3 * D, device=gpu, dtype=BF), M=300, qk_rope=ops.QkRope(wqd, wkd, cs, 0, 0, 2 * D))
    with pytest.raises(ValueError, match="positions"):
        ops.gemm(xd, wd, bd, one[Lc:], M=M, ldc=3 * D, c_map=ops.RowMap(Ls, Ljn), qk_rope=ops.QkRope(wqd, wkd, cs, Ls, Ljn, 2 * D))


def test_gemm256_every_k_tile_count_class_matches_the_128_kernel(gpu):
    """Round 5: the 256^2 kernel's K loop has a generic copy (first two and last two or three K-tiles) and a steady copy (pairs, static ring
    parity, B1's DMA pieces issued a phase early) with two hand-over flags between them. Every class of K-tile count goes through it here -
    1 .. 5 (generic only), 6 (one steady pair), odd and even counts with one, two and many pairs - on shapes of several tile rounds with ragged
    M / N edges, with and without the residual epilogue, against the 128^2 kernel (plain double buffer, same MFMA shape and K order): bit-identical.
    A child process, because UG_GEMM_FORCE_TILE is read per call only with UG_ENV_DYNAMIC=1."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r"""
import os, sys
os.environ["UG_ENV_DYNAMIC"] = "1"
sys.path.insert(0, %r)
import torch
from unigen_amd import ops, lib as L
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
rn = lambda *s: torch.randn(*s, generator=g, device=dev).to(torch.bfloat16)
bad = []
for nk in (1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 17, 20, 33):
    for (M, N, epi) in ((4096 + 70, 4096, L.EPI_BIAS), (8192, 2048 + 24, L.EPI_RES_GATE)):
        K = 64 * nk
        a, w, b = rn(M, K), rn(N, K) * 0.1, rn(N)
        res, gate = rn(M, N), rn((M + 999) // 1000, N)
        outs = []
        for tile in ("128", "256"):
            os.environ["UG_GEMM_FORCE_TILE"] = tile
            out = torch.zeros(M, N, device=dev, dtype=torch.bfloat16)
            kw = dict(M=M, epilogue=epi)
            if epi == L.EPI_RES_GATE:
                kw.update(residual=res, alpha=0.7, gate=gate, gate_ld=N, rows_per_sample=1000)
            ops.gemm(a, w, b, out, **kw)
            outs.append(out)
        torch.cuda.synchronize()
        if not torch.equal(outs[0], outs[1]):
            bad.append((nk, M, N, epi, float((outs[0] != outs[1]).float().mean())))
        ref = (a.float() @ w.float().t() + b.float()).to(torch.bfloat16)
        if epi == L.EPI_RES_GATE:
            ref = res + gate.repeat_interleave(1000, 0)[:M] * ref
        e = float((outs[1].float() - ref.float()).norm() / ref.float().norm())
        if not e < 2e-3:
            bad.append((nk, M, N, epi, "vs formula", e))
print("KTILE_CLASSES", bad)
""" % root
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("KTILE_CLASSES")][-1]
    assert line == "KTILE_CLASSES []", line
