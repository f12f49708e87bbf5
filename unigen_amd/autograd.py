"""torch.autograd.Function wrappers of the HIP hot ops: the second caller of the forward (reference train.py:622-662:
`accelerator.backward(loss)` through the same transformer). torch.autograd is the tape (as it is in the reference); every matrix product of the
backward is ug_gemm_bf16 again - dX = dY W through a transposed copy of W, dW = dY^T X through transposed copies of dY and X, and the attention
backward as five grouped GEMMs per sample around the elementwise kernels of csrc/backward.hip. fp32 tensors select the `_f32` verification twins,
as in the forward. There is no CPU path."""
from __future__ import annotations

import os
from typing import Optional

import torch

from . import lib as L
from . import ops

_wt_cache: dict = {}


def _pad64(n: int) -> int:
    return (n + 63) // 64 * 64


def _w_transposed(w: torch.Tensor) -> torch.Tensor:
    """[N, K] -> [K, N] (N must be a multiple of 64: it becomes the contraction length). Frozen weights are transposed once."""
    if w.requires_grad:
        return ops.transpose(w.detach())
    # the entry pins the STORAGE (a detached alias, not the Parameter object: `p.data = new_buffer` - engine._pack, nn.Module._apply - would
    # free the old storage under a pinned Parameter and let another weight of the same shape land on its address), so the address cannot be
    # recycled while the entry exists; `_version` catches in-place updates (load_state_dict, optimizer steps on a formerly frozen weight).
    # Re-pointing also calls clear_caches() (engine._pack / _pack_stack), which drops the dead transposed copies.
    key = (w.data_ptr(), tuple(w.shape), tuple(w.stride()), w.dtype, w._version)
    hit = _wt_cache.get(key)
    if hit is None:
        if len(_wt_cache) > 4096:
            _wt_cache.clear()
        hit = _wt_cache[key] = (ops.transpose(w.detach()), w.detach())
    return hit[0]


def clear_caches() -> None:
    """Drop EVERY cached transpose (frozen weights and recent activations). Only for when parameters are re-pointed at new storage
    (engine._pack / _pack_stack / _apply call it); a training loop must NOT call this per step - every frozen weight would be re-transposed
    each step - it calls clear_activation_cache()."""
    _wt_cache.clear()
    _xt_cache.clear()


def clear_activation_cache() -> None:
    """Drop the (up to 4) activation transposes of the last backward and the source tensors they pin - hundreds of MB at full size that would
    otherwise outlive the step. Call after `loss.backward()` (tools/train_bench.py, tests/fullsize_train_parity.py do); the frozen-weight
    transposes stay."""
    _xt_cache.clear()


_xt_cache: list = []        # the last few activation transposes of this backward pass: (source tensor, padded length, transposed copy)


def _x_transposed(x: torch.Tensor, Mp: int) -> torch.Tensor:
    """X^T for wgrad. to_q / to_k / to_v (+ proj_mlp), the added projections and the expert linears read the same input: their backward nodes
    run back to back, so a 4-entry cache removes most of the repeated transposes. Entries pin their source (no address reuse while cached)."""
    for src, mp, xt in _xt_cache:
        if src.data_ptr() == x.data_ptr() and src.shape == x.shape and src._version == x._version and mp == Mp and src.stride() == x.stride():
            return xt
    xt = ops.transpose(x, Mp)
    _xt_cache.append((x, Mp, xt))
    if len(_xt_cache) > 4:
        _xt_cache.pop(0)
    return xt


class Linear(torch.autograd.Function):
    """F.linear on [M, K] rows: ug_gemm_bf16 forward; backward = two more GEMMs and a column sum."""

    @staticmethod
    def forward(ctx, x, w, b):
        M, K = x.shape
        N = w.shape[0]
        out = torch.empty(M, N, device=x.device, dtype=x.dtype)
        ops.gemm(x, w, b, out, M=M)
        ctx.save_for_backward(x, w)
        ctx.has_bias = b is not None
        return out

    @staticmethod
    def _grads(need_dx, need_dw, need_db, x, w, dy):
        """(dX = dY W, dW = dY^T X, db = colsum(dY)) of y = x w^T + b for the requested ones; dy contiguous [M, N]."""
        M, K = x.shape
        N = w.shape[0]
        dx = dw = db = None
        if need_dx:
            if N % 64 != 0:
                raise L.UniGenHipError(f"Linear backward: out_features={N} must be a multiple of 64 (it is the contraction length of dX = dY W)")
            dx = torch.empty(M, K, device=x.device, dtype=x.dtype)
            ops.gemm(dy, _w_transposed(w), None, dx, M=M)
        if need_dw:
            # transposed copies + ug_gemm (also what the fp32 verification path runs). A transposed-operand kernel (dY^T X straight from the row-major
            # operands, tools/probe: csrc/gemm_tn.hip) measured 6 % SLOWER per step than this (0.884 vs 0.832 s backward at B = 2): probe library only.
            Mp = _pad64(M)
            dyt, xt = ops.transpose(dy, Mp), _x_transposed(x, Mp)          # [N, Mp], [K, Mp]
            dw = torch.empty(N, K, device=x.device, dtype=x.dtype)
            ops.gemm(dyt, xt, None, dw, M=N)
        if need_db:
            db = ops.colsum(dy).view(N)
        return dx, dw, db

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        return Linear._grads(ctx.needs_input_grad[0], ctx.needs_input_grad[1], ctx.has_bias and ctx.needs_input_grad[2], x, w, dy.contiguous())


class LinearResScale(torch.autograd.Function):
    """r + alpha * F.linear(x, w, b) on [M, K] rows - the zero-res projections `x + controlnet_add_*(z) * conditioning_scale`
    (src/UniGenTransformer.py:1104,1141,1166) - as ONE ug_gemm_bf16 with the UG_EPI_RES_SCALE epilogue (the rounding points of the three torch
    ops: Linear output, * alpha, + r each a bf16 tensor). Backward: d r = d y; d(lin) = alpha * d y, then Linear's two GEMMs and a column sum."""

    @staticmethod
    def forward(ctx, r, x, w, b, alpha):
        M, K = x.shape
        N = w.shape[0]
        out = torch.empty(M, N, device=x.device, dtype=x.dtype)
        ops.gemm(x, w, b, out, M=M, epilogue=L.EPI_RES_SCALE, residual=r, alpha=float(alpha))
        ctx.save_for_backward(x, w)
        ctx.has_bias, ctx.alpha = b is not None, float(alpha)
        return out

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dy = dy.contiguous()
        du = dy if ctx.alpha == 1.0 else (dy * ctx.alpha)
        dx, dw, db = Linear._grads(ctx.needs_input_grad[1], ctx.needs_input_grad[2], ctx.has_bias and ctx.needs_input_grad[3], x, w, du)
        return (dy if ctx.needs_input_grad[0] else None), dx, dw, db, None


def linear_res_scale(r: torch.Tensor, x: torch.Tensor, w: torch.Tensor, b: Optional[torch.Tensor], alpha: float) -> torch.Tensor:
    """r + alpha * linear(x): r [..., N], x [..., K] with equal leading dims."""
    lead = x.shape[:-1]
    return LinearResScale.apply(r.reshape(-1, r.shape[-1]).contiguous(), x.reshape(-1, x.shape[-1]).contiguous(), w, b, alpha).view(*lead, w.shape[0])


class LinearCat2(torch.autograd.Function):
    """F.linear(torch.cat([a, m], -1), w, b) WITHOUT the concatenation - the single block's proj_out over [attention | gelu(mlp)]
    (diffusers FluxSingleTransformerBlock, called at src/UniGenTransformer.py:1151): the m columns run as the K-segment extension of ug_gemm_bf16
    (the LoRA segment: T = m, B = w[:, Ka:]), one kernel, one fp32 accumulation in the concatenated K order -> the same bits as the GEMM over the
    materialised [M, Ka + Km] tensor. Backward: d a and d m as two GEMMs into their own buffers, d w column block by column block."""

    @staticmethod
    def forward(ctx, a, m, w, b):
        M, Ka = a.shape
        N = w.shape[0]
        out = torch.empty(M, N, device=a.device, dtype=a.dtype)
        ops.gemm(a, w[:, :Ka], b, out, M=M, lora_t=m, lora_b=w[:, Ka:])
        ctx.save_for_backward(a, m, w)
        ctx.has_bias = b is not None
        return out

    @staticmethod
    def backward(ctx, dy):
        a, m, w = ctx.saved_tensors
        dy = dy.contiguous()
        M, Ka = a.shape
        Km, N = m.shape[1], w.shape[0]
        da = dm = dw = db = None
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
            if N % 64 != 0:
                raise L.UniGenHipError(f"LinearCat2 backward: out_features={N} must be a multiple of 64")
            wt = _w_transposed(w)                                             # [Ka + Km, N]: row blocks are the transposed column blocks of w
            if ctx.needs_input_grad[0]:
                da = torch.empty(M, Ka, device=a.device, dtype=a.dtype)
                ops.gemm(dy, wt[:Ka], None, da, M=M)
            if ctx.needs_input_grad[1]:
                dm = torch.empty(M, Km, device=a.device, dtype=a.dtype)
                ops.gemm(dy, wt[Ka:], None, dm, M=M)
        if ctx.needs_input_grad[2]:
            Mp = _pad64(M)
            dyt = ops.transpose(dy, Mp)
            dw = torch.empty(N, Ka + Km, device=a.device, dtype=a.dtype)
            ops.gemm(dyt, _x_transposed(a, Mp), None, dw[:, :Ka], M=N)
            ops.gemm(dyt, _x_transposed(m, Mp), None, dw[:, Ka:], M=N)
        if ctx.has_bias and ctx.needs_input_grad[3]:
            db = ops.colsum(dy).view(N)
        return da, dm, dw, db


def linear_cat2(a: torch.Tensor, m: torch.Tensor, w: torch.Tensor, b: Optional[torch.Tensor]) -> torch.Tensor:
    lead = a.shape[:-1]
    return LinearCat2.apply(a.reshape(-1, a.shape[-1]).contiguous(), m.reshape(-1, m.shape[-1]).contiguous(), w, b).view(*lead, w.shape[0])


def _adjacent(ts) -> bool:
    """The tensors are consecutive row blocks of one contiguous buffer (the packed QKV / QKV+MLP weights of engine._pack)."""
    for a, b in zip(ts[:-1], ts[1:]):
        if a is None or b is None or not a.is_contiguous() or not b.is_contiguous() or a.dtype != b.dtype:
            return False
        if b.data_ptr() != a.data_ptr() + a.numel() * a.element_size() or a.shape[1:] != b.shape[1:]:
            return False
    return ts[0] is not None and ts[0].is_contiguous()


class LinearN(torch.autograd.Function):
    """n Linear layers that read the SAME rows x (to_q | to_k | to_v [| proj_mlp] of an attention block): ONE ug_gemm_bf16 over their packed
    weights when the parameters sit back to back in memory (engine._pack re-points them so), n GEMMs into the column blocks of one buffer
    otherwise. Returns the n column blocks as views. Backward: d x accumulates through the GEMM's residual epilogue (dy_1 W_1, then
    R + dy_2 W_2, ...: the same bf16 sums autograd would form with separate add kernels), d w_i = dy_i^T x with ONE transposed copy of x."""

    @staticmethod
    def forward(ctx, x, n, *wb):
        ws, bs = wb[:n], wb[n:]
        M, K = x.shape
        Ns = [w.shape[0] for w in ws]
        out = torch.empty(M, sum(Ns), device=x.device, dtype=x.dtype)
        has_b = all(b is not None for b in bs)
        if _adjacent(list(ws)) and (has_b and _adjacent([b.view(-1, 1) for b in bs]) or all(b is None for b in bs)):
            wp = torch.as_strided(ws[0], (sum(Ns), K), (K, 1))
            bp = torch.as_strided(bs[0], (sum(Ns),), (1,)) if has_b else None
            ops.gemm(x, wp, bp, out, M=M)
        else:
            c0 = 0
            for w, b, N in zip(ws, bs, Ns):
                ops.gemm(x, w, b, out[:, c0:], M=M)          # ldc = the buffer's row stride, N columns from column c0
                c0 += N
        ctx.save_for_backward(x, *ws)
        ctx.n, ctx.Ns, ctx.has_b = n, Ns, [b is not None for b in bs]
        outs, c0 = [], 0
        for N in Ns:
            outs.append(out[:, c0:c0 + N]); c0 += N
        return tuple(outs)

    @staticmethod
    def backward(ctx, *dys):
        saved = ctx.saved_tensors            # ONE access: under torch.utils.checkpoint a second unpack of the same tensors is an error
        x, ws = saved[0], saved[1:]
        n, M, K = ctx.n, x.shape[0], x.shape[1]
        dx = None
        dws, dbs = [None] * n, [None] * n
        Mp = _pad64(M)
        for i, (dy, w) in enumerate(zip(dys, ws)):
            if dy is None:
                continue
            dy = dy.contiguous()
            N = w.shape[0]
            if ctx.needs_input_grad[0]:
                if N % 64 != 0:
                    raise L.UniGenHipError(f"LinearN backward: out_features={N} must be a multiple of 64")
                if dx is None:
                    dx = torch.empty(M, K, device=x.device, dtype=x.dtype)
                    ops.gemm(dy, _w_transposed(w), None, dx, M=M)
                else:                                             # dx += dy W (in place: every element is read and written by one lane)
                    ops.gemm(dy, _w_transposed(w), None, dx, M=M, epilogue=L.EPI_RES_SCALE, residual=dx, alpha=1.0)
            if ctx.needs_input_grad[2 + i]:
                dw = torch.empty(N, K, device=x.device, dtype=x.dtype)
                ops.gemm(ops.transpose(dy, Mp), _x_transposed(x, Mp), None, dw, M=N)
                dws[i] = dw
            if ctx.has_b[i] and ctx.needs_input_grad[2 + n + i]:
                dbs[i] = ops.colsum(dy).view(N)
        return (dx, None, *dws, *dbs)


def linear_n(x: torch.Tensor, ws, bs):
    """[F.linear(x, w_i, b_i) for i]: x [..., K] -> views [..., N_i] of one buffer."""
    lead = x.shape[:-1]
    outs = LinearN.apply(x.reshape(-1, x.shape[-1]).contiguous(), len(ws), *ws, *bs)
    return [o.unflatten(0, lead) if len(lead) != 1 else o for o in outs]


class MoeGate(torch.autograd.Function):
    """gates = softmax(F.linear((x + c).float(), wg.float())) [S, E] fp32 and the arg-max expert per token: ug_moe_gate_top1 forward,
    ug_moe_gate_bwd backward (deepspeed TopKGate, src/UniGenUtils.py:99). top_k = 2 (top2gating): ug_moe_gate_top2, idx [2, S] with the second
    choice drawn through `noise` (the Gumbel sample added to the logits). idx is not differentiable, and neither is the noised arg-max.
    Third output: the fp32 logits [S, E] for top_k > 2 (topkgating's capacity rule ranks them; not differentiable), an empty tensor otherwise."""

    @staticmethod
    def forward(ctx, x, c, wg, top_k=1, noise=None):
        S, D = x.shape
        E = wg.shape[0]
        gates = torch.empty(S, E, device=x.device, dtype=torch.float32)
        wgc = wg.contiguous()
        if top_k == 1:
            idx = torch.empty(S, device=x.device, dtype=torch.int32)
            ops.moe_gate_top1(x, c, wgc, gates, idx)
        elif top_k == 2:
            idx = torch.empty(2, S, device=x.device, dtype=torch.int32)
            ops.moe_gate_top2(x, c, wgc, noise, gates, idx)
        else:                                    # topkgating: idx [K, S]
            idx = torch.empty(top_k, S, device=x.device, dtype=torch.int32)
            logits = torch.empty(S, E, device=x.device, dtype=torch.float32)
            ops.moe_gate_topk(x, c, wgc, top_k, gates, logits, idx)
        if top_k <= 2:
            logits = torch.empty(0, device=x.device, dtype=torch.float32)
        ctx.save_for_backward(gates, x, c, wgc)
        ctx.mark_non_differentiable(idx, logits)
        return gates, idx, logits

    @staticmethod
    def backward(ctx, dgates, _didx, _dlogits):
        gates, x, c, wg = ctx.saved_tensors
        dxc, dwg = ops.moe_gate_bwd(gates, dgates.float().contiguous(), x, c, wg)
        return (dxc if ctx.needs_input_grad[0] else None), (dxc if ctx.needs_input_grad[1] else None), (dwg if ctx.needs_input_grad[2] else None), None, None


def linear(x: torch.Tensor, w: torch.Tensor, b: Optional[torch.Tensor]) -> torch.Tensor:
    lead = x.shape[:-1]
    return Linear.apply(x.reshape(-1, x.shape[-1]).contiguous(), w, b).view(*lead, w.shape[0])


class GeluTanh(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        ctx.save_for_backward(x)
        return ops.gelu_tanh(x)

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        return ops.gelu_tanh_bwd(x, dy)


class AdaLNModulate(torch.autograd.Function):
    """LayerNorm(x; eps, no affine) * (1 + scale[:, None]) + shift[:, None]: x [B, L, D], shift / scale [B, D]."""

    @staticmethod
    def forward(ctx, x, shift, scale, eps):
        B, Ls, D = x.shape
        x2 = x.reshape(B * Ls, D).contiguous()
        shift, scale = shift.contiguous(), scale.contiguous()
        out = torch.empty(B * Ls, D, device=x.device, dtype=x.dtype)
        ops.adaln_modulate(x2, shift, scale, out, rows=B * Ls, D=D, rows_per_sample=Ls, mod_ld=D, eps=eps)
        ctx.save_for_backward(x2, scale)
        ctx.geom, ctx.eps = (B, Ls, D), eps
        return out.view(B, Ls, D)

    @staticmethod
    def backward(ctx, dy):
        x2, scale = ctx.saved_tensors
        B, Ls, D = ctx.geom
        dy2 = dy.reshape(B * Ls, D).contiguous()
        dx, dshift, dscale = ops.adaln_modulate_bwd(x2, dy2, scale, rows_per_sample=Ls, eps=ctx.eps)
        return dx.view(B, Ls, D), dshift, dscale, None


def adaln_modulate(x, shift, scale, eps: float = 1e-6):
    return AdaLNModulate.apply(x, shift, scale, eps)


class GateResidual(torch.autograd.Function):
    """x + gate.unsqueeze(1) * a (x, a [B, L, D]; gate [B, D]) as one kernel; backward: d x = d y, d a = gate * d y, d gate = sum_rows(d y * a)."""

    @staticmethod
    def forward(ctx, x, a, gate):
        B, Ls, D = a.shape
        a2, g2 = a.reshape(B * Ls, D).contiguous(), gate.contiguous()
        y = ops.gate_residual(x.reshape(B * Ls, D).contiguous(), a2, g2, Ls)
        ctx.save_for_backward(a2, g2)
        ctx.geom = (B, Ls, D)
        return y.view(B, Ls, D)

    @staticmethod
    def backward(ctx, dy):
        a2, g2 = ctx.saved_tensors
        B, Ls, D = ctx.geom
        dy2 = dy.reshape(B * Ls, D).contiguous()
        da = ops.gate_residual(None, dy2, g2, Ls).view(B, Ls, D) if ctx.needs_input_grad[1] else None
        dg = ops.colsum(dy2, a2, rows_per_group=Ls) if ctx.needs_input_grad[2] else None
        return dy, da, dg


def gate_residual(x, a, gate):
    return GateResidual.apply(x, a, gate)


class QKNormRope(torch.autograd.Function):
    """RMSNorm over each head (weight w [dh] or None) followed by apply_rotary_emb (cos / sin [positions, dh] fp32 or None) of one of q / k:
    x [B, L, H * dh]; row l of a sample sits at position pos_offset + l."""

    @staticmethod
    def forward(ctx, x, w, cos, sin, heads, pos_offset, eps):
        B, Ls, HD = x.shape
        dh = HD // heads
        out = x.reshape(B * Ls, HD).clone()
        ops.qk_rmsnorm_rope(out, batches=B, rows_per_batch=Ls, ld=HD, q_off=-1, k_off=0, heads=heads, dh=dh, pos_offset=pos_offset, wk_b=w, split=0,
                            cos=cos, sin=sin, eps=eps)
        ctx.save_for_backward(x.reshape(B * Ls, HD), w, cos, sin)
        ctx.geom = (B, Ls, heads, dh, pos_offset, eps)
        return out.view(B, Ls, HD)

    @staticmethod
    def backward(ctx, dy):
        x2, w, cos, sin = ctx.saved_tensors
        B, Ls, heads, dh, pos_offset, eps = ctx.geom
        dy2 = dy.reshape(B * Ls, heads * dh).contiguous()
        if x2.stride(-1) != 1:
            x2 = x2.contiguous()                 # (a column block of a fused QKV output keeps its row stride: the kernel takes any)
        dx, dw = ops.qk_rmsnorm_rope_bwd(x2, dy2, w, cos, sin, rows_per_batch=Ls, pos_offset=pos_offset, heads=heads, dh=dh, eps=eps)
        if w is None or not ctx.needs_input_grad[1]:
            dw = None
        return dx.view(B, Ls, heads * dh), dw, None, None, None, None, None


def qk_norm_rope(x, w, rope, heads: int, pos_offset: int = 0, eps: float = 1e-6):
    cos, sin = rope if rope is not None else (None, None)
    if w is None and cos is None:
        return x
    return QKNormRope.apply(x, w, cos, sin, heads, pos_offset, eps)


class FlashAttention(torch.autograd.Function):
    """F.scaled_dot_product_attention over heads packed as [B, L, H * dh] (any row stride). Forward: ug_flash_attn_fwd. Backward, per sample and
    for all heads at once (grouped GEMMs): S = Q K^T (fp32) -> lse -> P; dP = dO V^T; dS = scale P (dP - rowsum(dO O));
    dV = P^T dO, dK = dS^T Q, dQ = dS K. Lq and Lkv become contraction lengths: other lengths are zero-padded to multiples of 64 (padded keys get
    P = 0, padded queries carry dO = 0)."""

    @staticmethod
    def forward(ctx, q, k, v, heads):
        B, Lq, HD = q.shape
        Lkv = k.shape[1]
        dh = HD // heads
        o = torch.empty(B, Lq, HD, device=q.device, dtype=q.dtype)
        # bf16: the forward kernel also leaves the rows' log-sum-exp for the backward kernels (zero padding up to a multiple of 64 rows)
        lse = None
        if q.dtype == torch.bfloat16 and os.environ.get("UG_ATTN_BWD", "flash") != "gemm":
            lse = torch.zeros(B, heads, _pad64(Lq), device=q.device, dtype=torch.float32)
        ops.flash_attn(q, k, v, o, batches=B, heads=heads, dh=dh, Lq=Lq, Lkv=Lkv, q_strides=(q.stride(1), q.stride(0)), k_strides=(k.stride(1), k.stride(0)),
                       v_strides=(v.stride(1), v.stride(0)), o_strides=(HD, Lq * HD), lse=lse)
        ctx.save_for_backward(q, k, v, o)
        ctx.heads, ctx.lse = heads, lse
        return o

    @staticmethod
    def backward(ctx, do):
        q, k, v, o = ctx.saved_tensors
        H = ctx.heads
        B, Lq, HD = q.shape
        Lkv = k.shape[1]
        dh = HD // H
        dt, dev = q.dtype, q.device
        scale = dh ** -0.5
        do = do.contiguous()
        if dt == torch.bfloat16 and os.environ.get("UG_ATTN_BWD", "flash") != "gemm":
            # product path: the tiled backward kernels (csrc/attention.hip); UG_ATTN_BWD=gemm keeps the GEMM formulation below (A/B, and what fp32 runs)
            dq, dk, dv = ops.flash_attn_bwd(q, k, v, o, do, heads=H, lse=ctx.lse)
            return dq, dk, dv, None
        Lq0, Lkv0 = Lq, Lkv
        if Lq % 64 or Lkv % 64:
            Lq, Lkv = _pad64(Lq), _pad64(Lkv)
            padr = lambda t, n: torch.nn.functional.pad(t, (0, 0, 0, n - t.shape[1]))
            q, o, do = padr(q, Lq), padr(o, Lq), padr(do, Lq)
            k, v = padr(k, Lkv), padr(v, Lkv)
        dq, dk, dv = torch.empty(B, Lq, HD, device=dev, dtype=dt), torch.empty(B, Lkv, HD, device=dev, dtype=dt), torch.empty(B, Lkv, HD, device=dev, dtype=dt)
        S, dP = torch.empty(H, Lq, Lkv, device=dev, dtype=torch.float32), torch.empty(H, Lq, Lkv, device=dev, dtype=torch.float32)
        for b in range(B):
            qb, kb, vb, ob, dob = q[b], k[b], v[b], o[b], do[b]
            ops.gemm(qb[:, :dh], kb[:, :dh], None, S, M=Lq, epilogue=L.EPI_F32, groups=H, a_gstride=dh, w_gstride=dh, c_gstride=Lq * Lkv, ldc=Lkv)
            lse = ops.row_lse(S.view(H * Lq, Lkv), scale, Lkv0)
            P = ops.attn_prob(S.view(H * Lq, Lkv), lse, scale, dt, Lkv0)
            ops.gemm(dob[:, :dh], vb[:, :dh], None, dP, M=Lq, epilogue=L.EPI_F32, groups=H, a_gstride=dh, w_gstride=dh, c_gstride=Lq * Lkv, ldc=Lkv)
            delta = ops.rowdot(dob, ob, H)                                   # [H, Lq]
            dS = ops.attn_dscore(P, dP.view(H * Lq, Lkv), delta.view(-1), scale)
            P3, dS3 = P.view(H, Lq, Lkv), dS.view(H, Lq, Lkv)
            per_head = lambda t, Lr: torch.as_strided(t, (H, Lr, dh), (dh, t.stride(0), 1))
            PT, dST = ops.transpose(P3), ops.transpose(dS3)                   # [H, Lkv, Lq]
            dOT, QT, KT = ops.transpose(per_head(dob, Lq)), ops.transpose(per_head(qb, Lq)), ops.transpose(per_head(kb, Lkv))   # [H, dh, L]
            g = dict(groups=H, c_gstride=dh, ldc=HD)
            ops.gemm(PT, dOT, None, dv[b], M=Lkv, a_gstride=Lkv * Lq, w_gstride=dh * Lq, **g)
            ops.gemm(dST, QT, None, dk[b], M=Lkv, a_gstride=Lkv * Lq, w_gstride=dh * Lq, **g)
            ops.gemm(dS3, KT, None, dq[b], M=Lq, a_gstride=Lq * Lkv, w_gstride=dh * Lkv, **g)
        return dq[:, :Lq0], dk[:, :Lkv0], dv[:, :Lkv0], None


def attention(q, k, v, heads: int):
    return FlashAttention.apply(q, k, v, heads)
