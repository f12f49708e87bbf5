"""Backward pass of the hot ops: every autograd.Function of unigen_amd/autograd.py against torch autograd of the
oracle's formulas on the CPU. fp32 tensors run the `_f32` verification twins (tolerance 1e-4: same math, different summation order); bf16 tensors
run the product kernels and must be as close to the fp32 gradients as torch's own bf16 autograd is (ratio <= 1.5, floor 2e-3)."""
import pytest
import torch
import torch.nn.functional as F

from oracle import unigen_ref as R

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


def rel(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def run_case(gpu, hip_fn, ref_fn, inputs, grad_mask=None, name="", keep_f32=()):
    """inputs: list of fp32 CPU tensors (None allowed). Returns nothing; asserts forward + gradients in both precisions."""
    g = torch.Generator().manual_seed(123)
    grad_mask = grad_mask or [t is not None and t.is_floating_point() for t in inputs]

    def leafs(dtype, device):
        out = []
        for i, (t, m) in enumerate(zip(inputs, grad_mask)):
            if t is None or not t.is_floating_point() or i in keep_f32:
                out.append(t if t is None else t.detach().clone().to(device))
            elif not m:
                out.append(t.detach().clone().to(dtype).to(device))
            else:
                out.append(t.detach().clone().to(dtype).to(device).requires_grad_(True))
        return out

    truth_in = leafs(torch.float32, "cpu")
    y32 = ref_fn(*truth_in)
    dy = torch.randn(y32.shape, generator=g)
    y32.backward(dy)
    truth = [t.grad if (t is not None and t.requires_grad) else None for t in truth_in]
    for dtype, tol in ((torch.float32, 1e-4), (BF, None)):
        hin = leafs(dtype, gpu)
        y = hip_fn(*hin)
        y.backward(dy.to(dtype).to(gpu))
        if dtype == BF:
            rin = leafs(BF, "cpu")
            yr = ref_fn(*rin)
            yr.backward(dy.to(BF))
        e_fwd = rel(y, y32)
        assert e_fwd <= (tol if tol else max(1.5 * rel(yr, y32), 2e-3)), (name, dtype, "forward", e_fwd)
        for i, (t, tr) in enumerate(zip(hin, truth)):
            if tr is None:
                continue
            e = rel(t.grad, tr)
            bound = tol if tol else max(1.5 * rel(rin[i].grad, tr), 2e-3)
            print(f"backward {name} {str(dtype)[6:]} grad[{i}] rel_l2 {e:.3e} (bound {bound:.3e})")
            assert e <= bound, (name, dtype, i, e, bound)


def test_linear_backward(gpu):
    from unigen_amd import autograd as A
    g = torch.Generator().manual_seed(1)
    for (M, K, N) in ((200, 128, 192), (64, 64, 64), (130, 256, 320)):
        x, w, b = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) * K ** -0.5, torch.randn(N, generator=g) * 0.1
        run_case(gpu, lambda x, w, b: A.linear(x, w, b), lambda x, w, b: F.linear(x, w, b), [x, w, b], name=f"linear{M}x{K}x{N}")
    # frozen weight: only dX
    x, w = torch.randn(100, 128, generator=g), torch.randn(64, 128, generator=g) * 0.1
    run_case(gpu, lambda x, w, b: A.linear(x, w, None), lambda x, w, b: F.linear(x, w), [x, w, None], grad_mask=[True, False, False], name="linear_frozen")


def test_gelu_and_adaln_backward(gpu):
    from unigen_amd import autograd as A
    g = torch.Generator().manual_seed(2)
    x = torch.randn(3, 70, 192, generator=g) * 2
    run_case(gpu, lambda x: A.GeluTanh.apply(x), lambda x: F.gelu(x, approximate="tanh"), [x], name="gelu")
    sh, sc = torch.randn(3, 192, generator=g) * 0.3, torch.randn(3, 192, generator=g) * 0.3
    ref = lambda x, sh, sc: R.layer_norm(x) * (1 + sc[:, None]) + sh[:, None]
    run_case(gpu, lambda x, sh, sc: A.adaln_modulate(x, sh, sc), ref, [x, sh, sc], name="adaln")


def test_qk_norm_rope_backward(gpu):
    from unigen_amd import autograd as A
    g = torch.Generator().manual_seed(3)
    for dh, H in ((128, 2), (64, 3)):
        B, Ls, off = 2, 40, 24
        x = torch.randn(B, Ls, H * dh, generator=g)
        w = 1 + 0.2 * torch.randn(dh, generator=g)
        ang = torch.rand(off + Ls, dh // 2, generator=g) * 6.28
        cos, sin = ang.cos().repeat_interleave(2, 1).contiguous(), ang.sin().repeat_interleave(2, 1).contiguous()

        def ref(x, w, cos, sin):
            xh = R._heads(x, H)                                   # [B, H, L, dh]
            xh = R.rms_norm(xh, w)
            xh = R.apply_rotary_emb(xh, cos[off:off + Ls], sin[off:off + Ls])
            return xh.transpose(1, 2).reshape(B, Ls, H * dh)

        hip = lambda x, w, cos, sin: A.qk_norm_rope(x, w, (cos, sin), H, pos_offset=off)
        run_case(gpu, hip, ref, [x, w, cos, sin], grad_mask=[True, True, False, False], name=f"qk_norm_rope_dh{dh}", keep_f32=(2, 3))
        run_case(gpu, lambda x, w: A.qk_norm_rope(x, w, None, H), lambda x, w: R.rms_norm(R._heads(x, H), w).transpose(1, 2).reshape(B, Ls, H * dh),
                 [x, w], name=f"qk_norm_only_dh{dh}")


def test_attention_backward(gpu):
    from unigen_amd import autograd as A
    g = torch.Generator().manual_seed(4)
    for dh, H, Lq, Lkv in ((128, 2, 128, 192), (64, 3, 64, 64)):
        B = 2
        qkv = torch.randn(B, Lkv, 3 * H * dh, generator=g)
        D = H * dh

        def ref(qkv):
            q, k, v = qkv[:, Lkv - Lq:, :D], qkv[:, :, D:2 * D], qkv[:, :, 2 * D:]
            o = R._sdpa(R._heads(q, H), R._heads(k, H), R._heads(v, H))
            return o.transpose(1, 2).reshape(B, Lq, D)

        hip = lambda qkv: A.attention(qkv[:, Lkv - Lq:, :D], qkv[:, :, D:2 * D], qkv[:, :, 2 * D:], H)
        run_case(gpu, hip, ref, [qkv], name=f"attention_dh{dh}_{Lq}x{Lkv}")


@pytest.mark.parametrize("dh,H,B,Lq,Lkv", [(128, 3, 2, 300, 333), (64, 2, 1, 64, 64), (128, 2, 2, 600, 520), (64, 4, 2, 257, 129),
                                           (128, 2, 1, 2100, 2200)])      # the last one: >= 2048 queries at dh 128 -> the pair-scheme dQ kernel by default
def test_flash_attention_backward_kernels(gpu, dh, H, B, Lq, Lkv, monkeypatch):
    """ug_flash_attn_bwd (lse / dQ / dK / dV kernels on the forward's tiling) against torch autograd of SDPA in fp32 on the CPU, ragged lengths,
    strided q / k / v inside one [B, L, 3 H dh] buffer; and against the GEMM formulation of the same backward (UG_ATTN_BWD=gemm)."""
    from unigen_amd import autograd as A
    g = torch.Generator().manual_seed(Lq + Lkv)
    D = H * dh
    qkv = torch.randn(B, Lkv, 3 * D, generator=g)
    dy = torch.randn(B, Lq, D, generator=g)

    def ref(qkv):
        q, k, v = qkv[:, :Lq, :D] if Lq <= Lkv else None, qkv[:, :, D:2 * D], qkv[:, :, 2 * D:]
        o = R._sdpa(R._heads(q, H), R._heads(k, H), R._heads(v, H))
        return o.transpose(1, 2).reshape(B, Lq, D)

    if Lq > Lkv:          # queries from their own buffer
        qsrc = torch.randn(B, Lq, D, generator=g)
        t_in = [qsrc.clone().requires_grad_(True), qkv.clone().requires_grad_(True)]
        o = R._sdpa(R._heads(t_in[0], H), R._heads(t_in[1][:, :, D:2 * D], H), R._heads(t_in[1][:, :, 2 * D:], H)).transpose(1, 2).reshape(B, Lq, D)
    else:
        t_in = [qkv.clone().requires_grad_(True)]
        o = ref(t_in[0])
    o.backward(dy)
    truth = [t.grad for t in t_in]

    def run():
        h_in = [t.detach().clone().to(BF).to(gpu).requires_grad_(True) for t in t_in]
        if Lq > Lkv:
            out = A.attention(h_in[0], h_in[1][:, :, D:2 * D], h_in[1][:, :, 2 * D:], H)
        else:
            out = A.attention(h_in[0][:, :Lq, :D], h_in[0][:, :, D:2 * D], h_in[0][:, :, 2 * D:], H)
        out.backward(dy.to(BF).to(gpu))
        return [t.grad for t in h_in]

    flash = run()
    monkeypatch.setenv("UG_ATTN_BWD", "gemm")
    gemm = run()
    for i, (f, gm, tr) in enumerate(zip(flash, gemm, truth)):
        e_f, e_g = rel(f, tr), rel(gm, tr)
        print(f"backward flash attention dh{dh} {Lq}x{Lkv} grad[{i}]: flash {e_f:.3e}, gemm formulation {e_g:.3e}")
        assert torch.isfinite(f.float()).all() and e_f <= max(1.5 * e_g, 6e-3), (i, e_f, e_g)


@pytest.mark.parametrize("dtype", [BF, torch.float32])
def test_transpose_and_colsum_helpers(gpu, dtype):
    """ug_transpose (bf16: the 16-byte fast path and the element-wise fallback; fp32 twin) with padded rows, strided sources and a batch dimension -
    bit-exact - and the two-stage ug_colsum against fp32 torch."""
    from unigen_amd import ops
    g = torch.Generator().manual_seed(9)
    for rows, cols, pad, lead, ld_extra in [(300, 192, 304, (), 0), (64, 64, 64, (), 0), (1000, 72, 1024, (3,), 0), (257, 130, 260, (), 0), (130, 256, 136, (2,), 64),
                                            (4608, 3072, 4608, (), 0), (77, 8, 80, (), 8)]:
        full = torch.randn(*lead, rows, cols + ld_extra, generator=g).to(dtype)
        src = full[..., :cols].to(gpu) if ld_extra == 0 else full.to(gpu)[..., :cols]        # ld_extra: a strided view (row stride > cols)
        got = ops.transpose(src, pad)
        want = torch.zeros(*lead, cols, pad, dtype=dtype)
        want[..., :rows] = full[..., :cols].transpose(-1, -2)
        assert got.shape == want.shape and torch.equal(got.cpu(), want), (rows, cols, pad, lead, ld_extra)
    for rows, cols in [(1000, 192), (4608, 3072), (64, 8)]:
        x = torch.randn(rows, cols, generator=g).to(dtype)
        got = ops.colsum(x.to(gpu)).float().cpu()
        want = x.float().sum(0)
        assert rel(got, want) < (3e-3 if dtype == BF else 1e-5), (rows, cols, rel(got, want))


@pytest.mark.parametrize("dh,H", [(128, 3), (64, 4), (32, 2)])
def test_qk_rmsnorm_rope_bwd_paths_agree(gpu, dh, H):
    """ug_qk_rmsnorm_rope_bwd: the 16-byte kernel (head widths 64 / 128 / 256, aligned operands) and the pair-wise fallback (other widths, or row strides
    that are not multiples of 8) against fp32 torch autograd of the oracle formula; ragged vector counts (the last wave iteration is partly empty)."""
    from unigen_amd import ops
    g = torch.Generator().manual_seed(dh)
    rows, HD = 101, H * dh
    x, dy = torch.randn(rows, HD, generator=g), torch.randn(rows, HD, generator=g)
    w = torch.randn(dh, generator=g) * 0.5 + 1.0
    ang = torch.randn(rows, dh // 2, generator=g)
    cos, sin = torch.cos(ang).repeat_interleave(2, 1).contiguous(), torch.sin(ang).repeat_interleave(2, 1).contiguous()
    xt, wt = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    xh = xt.view(rows, H, dh)
    un = xh * torch.rsqrt(xh.pow(2).mean(-1, keepdim=True) + 1e-6) * wt
    rot = torch.stack([-un[..., 1::2], un[..., 0::2]], -1).flatten(-2)
    y = un * cos[:, None] + rot * sin[:, None]
    y.reshape(rows, HD).backward(dy)
    for dt, tol in ((torch.float32, 2e-5), (BF, 1.5e-2)):
        for pad in (0, 2):                                # pad 2: row stride HD + 2 -> the fallback kernel
            xs = torch.zeros(rows, HD + pad, dtype=dt); xs[:, :HD] = x.to(dt)
            ds = torch.zeros(rows, HD + pad, dtype=dt); ds[:, :HD] = dy.to(dt)
            dx, dw = ops.qk_rmsnorm_rope_bwd(xs.to(gpu)[:, :HD], ds.to(gpu)[:, :HD], w.to(dt).to(gpu), cos.to(gpu), sin.to(gpu), rows_per_batch=rows, pos_offset=0,
                                             heads=H, dh=dh)
            # bf16 inputs are the rounded ones: compare against the fp32 gradients of the unrounded problem at bf16 tolerance
            assert rel(dx, xt.grad) < tol and rel(dw, wt.grad) < tol, (dh, dt, pad, rel(dx, xt.grad), rel(dw, wt.grad))


@pytest.mark.parametrize("rows,rps,D", [(70000, 1, 64), (3 * 333, 333, 1536), (2 * 100, 100, 520)])
def test_adaln_modulate_bwd_direct(gpu, rows, rps, D):
    """ug_adaln_modulate_bwd against fp32 torch autograd: per-token modulation with more samples than a grid dimension holds, ragged row chunks,
    a width that is not a multiple of 512."""
    from unigen_amd import ops
    g = torch.Generator().manual_seed(rows + D)
    x, dy = torch.randn(rows, D, generator=g), torch.randn(rows, D, generator=g)
    sc = torch.randn(rows // rps, D, generator=g) * 0.3
    xt, st = x.clone().requires_grad_(True), sc.clone().requires_grad_(True)
    sh = torch.zeros(rows // rps, D, requires_grad=True)
    y = F.layer_norm(xt, (D,), eps=1e-6).view(-1, rps, D) * (1 + st[:, None]) + sh[:, None]
    y.reshape(rows, D).backward(dy)
    dx, dshift, dscale = ops.adaln_modulate_bwd(x.to(gpu), dy.to(gpu), sc.to(gpu), rows_per_sample=rps)
    assert rel(dx, xt.grad) < 1e-5 and rel(dshift, sh.grad) < 1e-5 and rel(dscale, st.grad) < 1e-5, (rel(dx, xt.grad), rel(dshift, sh.grad), rel(dscale, st.grad))


def test_backward_kernels_run_to_run_bitwise(gpu):
    """No atomics anywhere in the backward: the attention backward, the q/k-norm and AdaLN backward (fixed assignment of rows to partial sums), colsum
    and the weight-gradient route return identical bits on repeated calls with the same inputs."""
    from unigen_amd import ops
    from unigen_amd import autograd as A
    g = torch.Generator().manual_seed(3)
    B, L, H, dh = 2, 1000, 4, 128
    D = H * dh
    q, k, v, do = (torch.randn(B, L, D, generator=g).to(BF).to(gpu) for _ in range(4))
    o = torch.empty_like(q)
    lse = torch.zeros(B, H, (L + 63) // 64 * 64, device=gpu, dtype=torch.float32)
    ops.flash_attn(q, k, v, o, batches=B, heads=H, dh=dh, Lq=L, Lkv=L, lse=lse, q_strides=(D, L * D), k_strides=(D, L * D), v_strides=(D, L * D), o_strides=(D, L * D))
    x2, dy2 = q.reshape(B * L, D), do.reshape(B * L, D)
    w = torch.randn(dh, generator=g).to(BF).to(gpu)
    cos, sin = torch.randn(L, dh, generator=g).to(gpu), torch.randn(L, dh, generator=g).to(gpu)
    sc = torch.randn(B, D, generator=g).to(BF).to(gpu)
    wlin = (torch.randn(256, D, generator=g) * 0.05).to(BF).to(gpu)

    def once():
        out = list(ops.flash_attn_bwd(q, k, v, o, do, heads=H, lse=lse))
        out += list(ops.qk_rmsnorm_rope_bwd(x2, dy2, w, cos, sin, rows_per_batch=L, pos_offset=0, heads=H, dh=dh))
        out += list(ops.adaln_modulate_bwd(x2, dy2, sc, rows_per_sample=L))
        out.append(ops.colsum(dy2))
        xin = x2.clone().requires_grad_(True); wl = wlin.clone().requires_grad_(True)
        A.linear(xin, wl, None).backward(dy2[:, :256].contiguous())
        out += [xin.grad, wl.grad]
        return out

    first, second = once(), once()
    for i, (a, b) in enumerate(zip(first, second)):
        assert torch.equal(a, b), f"output {i} differs between two identical calls"
    # the round-3 kernels at a length where both are the default (dh 128, >= 2048 queries: pair-scheme dQ + fused dK / dV; partial sums meet in LDS, no atomics)
    L2 = 2304
    q2, k2, v2, do2 = (torch.randn(1, L2, D, generator=g).to(BF).to(gpu) for _ in range(4))
    o2 = torch.empty_like(q2)
    lse2 = torch.zeros(1, H, L2, device=gpu, dtype=torch.float32)
    ops.flash_attn(q2, k2, v2, o2, batches=1, heads=H, dh=dh, Lq=L2, Lkv=L2, lse=lse2, q_strides=(D, L2 * D), k_strides=(D, L2 * D), v_strides=(D, L2 * D), o_strides=(D, L2 * D))
    a3, b3 = ops.flash_attn_bwd(q2, k2, v2, o2, do2, heads=H, lse=lse2), ops.flash_attn_bwd(q2, k2, v2, o2, do2, heads=H, lse=lse2)
    for x, y, nm in zip(a3, b3, ("dq", "dk", "dv")):
        assert torch.equal(x, y) and bool(torch.isfinite(x.float()).all()), nm


def test_gate_residual_backward(gpu):
    from unigen_amd import autograd as A
    g = torch.Generator().manual_seed(6)
    x, a, gate = torch.randn(3, 70, 192, generator=g), torch.randn(3, 70, 192, generator=g), torch.randn(3, 192, generator=g)
    run_case(gpu, lambda x, a, gt: A.gate_residual(x, a, gt), lambda x, a, gt: x + gt.unsqueeze(1) * a, [x, a, gate], name="gate_residual")


