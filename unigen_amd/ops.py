"""Thin torch-tensor front end over the C ABI (unigen_amd/lib.py).

torch is used only for device memory and the current HIP stream; every computation below runs in libunigen_hip.so.
All tensors must be bf16 CUDA(HIP) tensors with a contiguous last dimension unless stated otherwise.

fp32 verification mode: when the activation tensors are torch.float32 the SAME call goes to the entry point's `_f32` twin
(include/unigen_hip.h, last section) - fp32 storage, no intermediate rounding - so a model whose parameters are fp32 runs the
identical host orchestration in exact arithmetic. Mixed dtypes in one call are an error.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

from . import lib as L

bf16 = torch.bfloat16


class KernelTimer:
    """Optional per-launch timing for bench.py: HIP events recorded on the launch stream around each call of the named
    kernel classes, with the algorithmic FLOPs of the launch. Off unless installed with `set_timer`."""

    def __init__(self, kinds=("gemm", "attn")):
        self.kinds = set(kinds)
        self.records = []   # (kind, flops, start_event, end_event)
        self.tags = []      # per record: the launch's shape, e.g. (M, N, K, groups) of a GEMM (tools/shape_rates.py)

    def begin(self, kind):
        if kind not in self.kinds:
            return None
        ev = torch.cuda.Event(enable_timing=True)
        ev.record(torch.cuda.current_stream())
        return ev

    def end(self, kind, flops, ev0, tag=None):
        if ev0 is None:
            return
        ev1 = torch.cuda.Event(enable_timing=True)
        ev1.record(torch.cuda.current_stream())
        self.records.append((kind, flops, ev0, ev1))
        self.tags.append(tag)

    def summary(self):
        """-> {kind: dict(launches, flops, ms)} (call after a device synchronize)."""
        out = {}
        for kind, flops, e0, e1 in self.records:
            d = out.setdefault(kind, dict(launches=0, flops=0.0, ms=0.0))
            d["launches"] += 1
            d["flops"] += flops
            d["ms"] += e0.elapsed_time(e1)
        return out


_timer: Optional[KernelTimer] = None


def set_timer(t: Optional[KernelTimer]) -> None:
    global _timer
    _timer = t


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _p(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


f32 = torch.float32


def _chk(t: torch.Tensor, name: str, dtype=bf16) -> None:
    if not t.is_cuda:
        raise L.UniGenHipError(f"{name}: expected a GPU tensor (unigen_amd has no CPU path)")
    if t.dtype != dtype:
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")
    if t.dim() > 0 and t.stride(-1) != 1:
        raise ValueError(f"{name}: last dimension must be contiguous")


def _act(t: torch.Tensor, name: str):
    """Activation dtype of a call, taken from its first tensor: bf16 (product path) or fp32 (verification twins)."""
    if t.dtype not in (bf16, f32):
        raise TypeError(f"{name}: expected torch.bfloat16 (or torch.float32 for the verification path), got {t.dtype}")
    return t.dtype


def _fn(base: str, dt):
    """The C-ABI function for activation dtype `dt`: `base` itself for bf16, its `_f32` twin otherwise."""
    lib = L.load()
    return getattr(lib, base) if dt == bf16 else getattr(lib, _TWIN[base])


_TWIN = {v: k for k, v in L._F32_TWINS.items()}


class RowMap:
    """Logical row m -> physical row (m // rows_per_batch) * batch_stride + m % rows_per_batch (0 = identity)."""

    __slots__ = ("rpb", "bstride")

    def __init__(self, rows_per_batch: int = 0, batch_stride: int = 0):
        self.rpb, self.bstride = rows_per_batch, batch_stride


IDENT = RowMap()


class QkRope:
    """Operands of the UG_EPI_QKV_ROPE epilogue: RMSNorm weights of the q / k heads [128], the (cos, sin) pair table [positions, 64, 2]
    fp32, and the position of output row m = pos0 + m % rows_per_batch. Columns [0, until_n) of the projection are q | k heads."""

    __slots__ = ("wq", "wk", "cs", "rpb", "pos0", "until_n", "eps", "dh")

    def __init__(self, wq, wk, cs, rows_per_batch: int, pos0: int, until_n: int, eps: float = 1e-6, dh: int = 128):
        self.wq, self.wk, self.cs, self.rpb, self.pos0, self.until_n, self.eps, self.dh = wq, wk, cs, rows_per_batch, pos0, until_n, eps, dh


def qk_rope_fusable(M: int, N: int, until_n: int, dh: int, dtype: torch.dtype) -> bool:
    """Whether ug_gemm_bf16 can take q/k RMSNorm + RoPE in its epilogue for this projection: bf16, heads of 128, whole 256^2 tiles, and
    at least one full round of them (below that the dispatcher prefers the 128^2 kernel, which has no such epilogue).
    UG_GEMM_FUSE_QKROPE=0 keeps the stand-alone ug_qk_rmsnorm_rope pass (A/B measurements)."""
    if os.environ.get("UG_GEMM_FUSE_QKROPE", "1") == "0":
        return False
    return (dtype == torch.bfloat16 and dh in (64, 128) and M % 256 == 0 and N % 256 == 0 and until_n % 256 == 0 and (M // 256) * (N // 256) >= 256)


_gemm_ws: dict = {}


def _gemm_workspace(device, stream: int) -> torch.Tensor:
    """Caller-owned scratch for the GEMM's split-K tail: one per (device, stream), allocated once - launches on one stream are ordered, two
    streams must not share arrival tickets or slabs (include/unigen_hip.h: "One workspace per stream")."""
    key = (device, stream)
    ws = _gemm_ws.get(key)
    if ws is None:
        ws = torch.zeros(int(L.load().ug_gemm_workspace_bytes()), dtype=torch.uint8, device=device)   # tickets start at zero
        _gemm_ws[key] = ws
    return ws


def gemm(a: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], out: torch.Tensor, *, M: int,
         epilogue: int = L.EPI_BIAS, lda: Optional[int] = None, ldc: Optional[int] = None, a_map: RowMap = IDENT,
         c_map: RowMap = IDENT, residual: Optional[torch.Tensor] = None, ldr: Optional[int] = None,
         r_map: RowMap = IDENT, gate: Optional[torch.Tensor] = None, gate_ld: int = 0, rows_per_sample: int = 0,
         alpha: float = 1.0, groups: int = 1, a_gstride: int = 0, w_gstride: int = 0, bias_gstride: int = 0,
         c_gstride: int = 0, r_gstride: int = 0, gate_gstride: int = 0, lora_t: Optional[torch.Tensor] = None,
         lora_b: Optional[torch.Tensor] = None, gelu_from_n: int = 0, c_shift_from_n: int = 0, c_shift: int = 0,
         qk_rope: Optional[QkRope] = None) -> torch.Tensor:
    """out[m, :N] = epilogue(a[m, :K] @ w[:N, :K]^T + bias). `a`/`out`/`residual` are base tensors whose data_ptr is row 0
    (slices of a bigger buffer are fine); leading dims default to the tensors' row strides. `qk_rope` selects UG_EPI_QKV_ROPE
    (with gelu_from_n > 0 the columns from there on still get GELU)."""
    dt = _act(a, "a")
    if qk_rope is not None:
        if dt != torch.bfloat16:
            raise ValueError("the fused q/k RMSNorm + RoPE epilogue is bf16 only")
        epilogue = L.EPI_QKV_ROPE
    _chk(w, "w", dt)
    _chk(out, "out", torch.float32 if epilogue == L.EPI_F32 else dt)
    if bias is not None:
        _chk(bias, "bias", dt)
    N, K = (w.shape[-2], w.shape[-1])
    d = L.GemmDesc()
    d.A, d.lda, d.a_rpb, d.a_bstride = a.data_ptr(), (lda if lda is not None else a.stride(-2)), a_map.rpb, a_map.bstride
    d.W, d.ldw = w.data_ptr(), w.stride(-2)
    d.bias = _p(bias)
    d.C, d.ldc, d.c_rpb, d.c_bstride = out.data_ptr(), (ldc if ldc is not None else out.stride(-2)), c_map.rpb, c_map.bstride
    if residual is not None:
        _chk(residual, "residual", dt)
        d.R, d.ldr, d.r_rpb, d.r_bstride = residual.data_ptr(), (ldr if ldr is not None else residual.stride(-2)), r_map.rpb, r_map.bstride
    if gate is not None:
        _chk(gate, "gate", dt)
        d.gate, d.gate_ld, d.rows_per_sample = gate.data_ptr(), gate_ld, rows_per_sample
    d.alpha, d.epilogue = alpha, epilogue
    d.M, d.N, d.K = M, N, K
    d.groups, d.a_gstride, d.w_gstride, d.bias_gstride, d.c_gstride = groups, a_gstride, w_gstride, bias_gstride, c_gstride
    d.r_gstride, d.gate_gstride = r_gstride, gate_gstride
    d.gelu_from_n, d.c_shift_from_n, d.c_shift = gelu_from_n, c_shift_from_n, c_shift
    if qk_rope is not None:
        q = qk_rope
        _chk(q.wq, "qk_rope.wq", dt); _chk(q.wk, "qk_rope.wk", dt)
        if q.wq.numel() != q.dh or q.wk.numel() != q.dh:
            raise ValueError(f"qk_rope: weights must be [{q.dh}]")
        if q.cs is not None:
            _chk(q.cs, "qk_rope.cs", torch.float32)
            if q.cs.dim() != 3 or tuple(q.cs.shape[1:]) != (q.dh // 2, 2) or not q.cs.is_contiguous():
                raise ValueError(f"qk_rope: the pair table must be [positions, {q.dh // 2}, 2] contiguous")
            if q.cs.shape[0] < q.pos0 + (q.rpb if q.rpb else M):
                raise ValueError(f"qk_rope: the pair table has {q.cs.shape[0]} positions, rows reach {q.pos0 + (q.rpb if q.rpb else M)}")
        elif q.dh != 64:
            raise ValueError("qk_rope: the (cos, sin) table may only be omitted at head width 64")
        d.qk_wq, d.qk_wk, d.rope_cs = q.wq.data_ptr(), q.wk.data_ptr(), _p(q.cs)
        d.rope_rpb, d.rope_pos0, d.qk_until_n, d.qk_eps, d.qk_dh = q.rpb, q.pos0, q.until_n, q.eps, q.dh
    stream = _stream()
    ws = _gemm_workspace(a.device, stream)
    d.workspace, d.workspace_bytes = ws.data_ptr(), ws.numel()
    if lora_t is not None:
        _chk(lora_t, "lora_t", dt); _chk(lora_b, "lora_b", dt)
        d.lora_T, d.ldt, d.lora_B, d.ldb, d.lora_r = lora_t.data_ptr(), lora_t.stride(-2), lora_b.data_ptr(), lora_b.stride(-2), lora_b.shape[-1]
    ev = _timer.begin("gemm") if _timer is not None else None
    L.check(_fn("ug_gemm_bf16", dt)(C.byref(d), stream), "ug_gemm_bf16")
    if ev is not None:
        _timer.end("gemm", 2.0 * M * N * (K + (d.lora_r or 0)) * max(groups, 1), ev, tag=(M, N, K, max(groups, 1), epilogue))
    return out


def small_linear(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], out: torch.Tensor, *, silu_in: bool = False,
                 residual: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out[m] = residual[m] + bf16(act(x[m]) @ w^T + bias), M = x.shape[0] (chunks of <= 16 rows per launch)."""
    dt = _act(x, "x")
    _chk(w, "w", dt); _chk(out, "out", dt)
    if bias is not None:
        _chk(bias, "bias", dt)
    if residual is not None:
        _chk(residual, "residual", dt)
    M, K = x.shape
    N = w.shape[0]
    fn = _fn("ug_small_linear_bf16", dt)
    step = 16 if dt == bf16 else 64     # the bf16 kernel keeps up to 16 rows of x in LDS; more rows re-stream the weights
    for m0 in range(0, M, step):
        mm = min(step, M - m0)
        r = residual[m0:] if residual is not None else None
        L.check(fn(x[m0:].data_ptr(), x.stride(0), w.data_ptr(), w.stride(0), _p(bias), _p(r),
                   residual.stride(0) if residual is not None else 0, out[m0:].data_ptr(), out.stride(0),
                   mm, N, K, 1 if silu_in else 0, _stream()), "ug_small_linear_bf16")
    return out


def adaln_modulate(x: torch.Tensor, shift: torch.Tensor, scale: torch.Tensor, out: torch.Tensor, *, rows: int, D: int,
                   rows_per_sample: int, mod_ld: int, ldx: Optional[int] = None, x_map: RowMap = IDENT, eps: float = 1e-6) -> torch.Tensor:
    dt = _act(x, "x")
    _chk(shift, "shift", dt); _chk(scale, "scale", dt); _chk(out, "out", dt)
    L.check(_fn("ug_adaln_modulate", dt)(x.data_ptr(), ldx if ldx is not None else x.stride(-2), x_map.rpb, x_map.bstride,
                                       shift.data_ptr(), scale.data_ptr(), mod_ld, rows_per_sample, out.data_ptr(),
                                       out.stride(-2), rows, D, eps, _stream()), "ug_adaln_modulate")
    return out


def qk_rmsnorm_rope(buf: torch.Tensor, *, batches: int, rows_per_batch: int, ld: int, q_off: int, k_off: int, heads: int, dh: int,
                    batch_stride_rows: Optional[int] = None, pos_offset: int = 0, wq_a=None, wk_a=None, wq_b=None, wk_b=None,
                    split: int = 0, cos: Optional[torch.Tensor] = None, sin: Optional[torch.Tensor] = None, eps: float = 1e-6) -> torch.Tensor:
    """buf.data_ptr() is row 0 of batch 0 of the processed row range."""
    dt = _act(buf, "buf")
    for wt in (wq_a, wk_a, wq_b, wk_b):
        if wt is not None:
            _chk(wt, "qk-norm weight", dt)
    if batch_stride_rows is None:
        batch_stride_rows = rows_per_batch
    if cos is not None:
        _chk(cos, "cos", torch.float32); _chk(sin, "sin", torch.float32)
        assert cos.shape[0] >= pos_offset + rows_per_batch and cos.shape[1] == dh and cos.is_contiguous() and sin.is_contiguous(), \
            (cos.shape, pos_offset, rows_per_batch, dh)
    L.check(_fn("ug_qk_rmsnorm_rope", dt)(buf.data_ptr(), ld, batches, rows_per_batch, batch_stride_rows, pos_offset, q_off, k_off, heads, dh,
                                        _p(wq_a), _p(wk_a), _p(wq_b), _p(wk_b), split, _p(cos), _p(sin), eps, _stream()), "ug_qk_rmsnorm_rope")
    return buf


def flash_attn(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, out: torch.Tensor, *, batches: int, heads: int, dh: int,
               Lq: int, Lkv: int, q_strides, k_strides, v_strides, o_strides, scale: Optional[float] = None,
               lse: Optional[torch.Tensor] = None) -> torch.Tensor:
    """q/k/v/out are base tensors (data_ptr = element [batch 0, row 0, head 0, 0]); *_strides = (row_stride, batch_stride).
    lse (bf16 path only): fp32 [batches, heads, >= Lq], receives the rows' base-2 log-sum-exp for flash_attn_bwd."""
    dt = _act(q, "q")
    _chk(k, "k", dt); _chk(v, "v", dt); _chk(out, "out", dt)
    if scale is None:
        scale = dh ** -0.5
    ev = _timer.begin("attn") if _timer is not None else None
    if lse is not None:
        _chk(lse, "lse", torch.float32)
        assert dt == bf16 and lse.is_contiguous() and lse.shape[0] == batches and lse.shape[1] == heads and lse.shape[2] >= Lq
        L.check(L.load().ug_flash_attn_fwd_lse(q.data_ptr(), q_strides[0], q_strides[1], k.data_ptr(), k_strides[0], k_strides[1],
                                               v.data_ptr(), v_strides[0], v_strides[1], out.data_ptr(), o_strides[0], o_strides[1],
                                               batches, heads, Lq, Lkv, dh, scale, lse.data_ptr(), lse.shape[2], _stream()), "ug_flash_attn_fwd_lse")
    else:
        L.check(_fn("ug_flash_attn_fwd", dt)(q.data_ptr(), q_strides[0], q_strides[1], k.data_ptr(), k_strides[0], k_strides[1],
                                           v.data_ptr(), v_strides[0], v_strides[1], out.data_ptr(), o_strides[0], o_strides[1],
                                           batches, heads, Lq, Lkv, dh, scale, _stream()), "ug_flash_attn_fwd")
    if ev is not None:
        _timer.end("attn", 4.0 * batches * heads * Lq * Lkv * dh, ev, tag=(batches, heads, Lq, Lkv, dh))
    return out


def timestep_embed(t: torch.Tensor, out: torch.Tensor) -> torch.Tensor:
    _chk(t, "t", torch.float32)
    dt = _act(out, "out")
    L.check(_fn("ug_timestep_embed", dt)(t.data_ptr(), out.data_ptr(), out.stride(0), t.shape[0], out.shape[1], _stream()), "ug_timestep_embed")
    return out


def euler_step(x: torch.Tensor, v: torch.Tensor, dt: float) -> torch.Tensor:
    adt = _act(x, "x")
    _chk(v, "v", adt)
    assert x.is_contiguous() and v.is_contiguous() and x.numel() == v.numel()
    L.check(_fn("ug_euler_step", adt)(x.data_ptr(), v.data_ptr(), dt, x.numel(), _stream()), "ug_euler_step")
    return x


def cfg_combine(uncond: torch.Tensor, text: torch.Tensor, guidance_scale: float, out: torch.Tensor) -> torch.Tensor:
    dt = _act(uncond, "uncond")
    _chk(text, "text", dt); _chk(out, "out", dt)
    assert uncond.is_contiguous() and text.is_contiguous() and out.is_contiguous() and uncond.numel() == text.numel() == out.numel()
    L.check(_fn("ug_cfg_combine", dt)(uncond.data_ptr(), text.data_ptr(), guidance_scale, out.data_ptr(), out.numel(), _stream()), "ug_cfg_combine")
    return out


def add(a: torch.Tensor, b: torch.Tensor, out: torch.Tensor) -> torch.Tensor:
    dt = _act(a, "a")
    _chk(b, "b", dt); _chk(out, "out", dt)
    D = a.shape[-1]
    rows = a.numel() // D
    a2, b2, o2 = a.reshape(rows, D), b.reshape(rows, D), out.view(rows, D)
    L.check(_fn("ug_add_bf16", dt)(a2.data_ptr(), a2.stride(0), b2.data_ptr(), b2.stride(0), o2.data_ptr(), o2.stride(0), rows, D, _stream()), "ug_add_bf16")
    return out


def add_rowbcast_f32(x: torch.Tensor, table: torch.Tensor, rows_per_batch: int) -> torch.Tensor:
    """x[r] = bf16(x[r] + table[r % rows_per_batch]) in place; x [rows, D] bf16, table [rows_per_batch, D] fp32."""
    dt = _act(x, "x")
    _chk(table, "table", torch.float32)
    rows, D = x.shape
    assert table.shape == (rows_per_batch, D)
    L.check(_fn("ug_add_rowbcast_f32", dt)(x.data_ptr(), x.stride(0), table.data_ptr(), table.stride(0), rows, rows_per_batch, D, _stream()), "ug_add_rowbcast_f32")
    return x


def gather_rows(src: torch.Tensor, idx: torch.Tensor, out: torch.Tensor) -> torch.Tensor:
    """out[i] = src[idx[i]] (zeros where idx[i] < 0); src [R, W], idx int32 [n], out [n, W]."""
    dt = _act(src, "src")
    _chk(idx, "idx", torch.int32); _chk(out, "out", dt)
    n, W = out.shape
    assert idx.numel() == n and src.shape[1] == W and idx.is_contiguous()
    L.check(_fn("ug_gather_rows", dt)(src.data_ptr(), src.stride(0), idx.data_ptr(), out.data_ptr(), out.stride(0), n, W, _stream()), "ug_gather_rows")
    return out


def moe_gate_top1(x: torch.Tensor, c: torch.Tensor, wg: torch.Tensor, gates: torch.Tensor, idx: torch.Tensor) -> None:
    dt = _act(x, "x")
    _chk(c, "c", dt); _chk(wg, "wg", dt); _chk(gates, "gates", torch.float32); _chk(idx, "idx", torch.int32)
    S, D = x.shape
    E = wg.shape[0]
    assert x.stride(0) == c.stride(0) and wg.is_contiguous() and gates.is_contiguous()
    L.check(_fn("ug_moe_gate_top1", dt)(x.data_ptr(), c.data_ptr(), x.stride(0), wg.data_ptr(), S, D, E, gates.data_ptr(), idx.data_ptr(), _stream()),
            "ug_moe_gate_top1")


def moe_gate_bwd(gates: torch.Tensor, dgates: torch.Tensor, x: torch.Tensor, c: torch.Tensor, wg: torch.Tensor):
    """Backward of moe_gate_top1: gates, d gates [S, E] fp32; x, c [S, D]; wg [E, D] -> (d(x + c) [S, D], d wg [E, D] in wg's dtype)."""
    dt = _act(x, "x")
    _chk(c, "c", dt); _chk(wg, "wg", dt); _chk(gates, "gates", torch.float32); _chk(dgates, "dgates", torch.float32)
    S, D = x.shape
    E = wg.shape[0]
    assert x.stride(0) == c.stride(0) and wg.is_contiguous() and gates.is_contiguous() and dgates.is_contiguous() and gates.shape == dgates.shape == (S, E)
    dx = torch.empty(S, D, device=x.device, dtype=dt)
    part = torch.empty(int(L.load().ug_moe_gate_bwd_slices(S)), E, D, device=x.device, dtype=torch.float32)
    L.check(_fn("ug_moe_gate_bwd", dt)(gates.data_ptr(), dgates.data_ptr(), x.data_ptr(), c.data_ptr(), x.stride(0), wg.data_ptr(), S, D, E, dx.data_ptr(), D,
                                     part.data_ptr(), _stream()), "ug_moe_gate_bwd")
    return dx, part.sum(0).to(dt)          # the kernel's per-slice fp32 partials, added in a fixed order


def moe_capacity_rts(gates, idx, uniform, capacity: int, slot, token_of_slot, exp_counts, l_aux) -> None:
    _chk(gates, "gates", torch.float32); _chk(idx, "idx", torch.int32); _chk(uniform, "uniform", torch.float32)
    _chk(slot, "slot", torch.int32); _chk(token_of_slot, "token_of_slot", torch.int32)
    _chk(exp_counts, "exp_counts", torch.int64); _chk(l_aux, "l_aux", torch.float32)
    S, E = gates.shape
    assert uniform.shape == (S, E) and uniform.is_contiguous() and token_of_slot.numel() == E * capacity
    L.check(L.load().ug_moe_capacity_rts(gates.data_ptr(), idx.data_ptr(), uniform.data_ptr(), S, E, capacity, slot.data_ptr(),
                                         token_of_slot.data_ptr(), exp_counts.data_ptr(), l_aux.data_ptr(), _stream()), "ug_moe_capacity_rts")


def moe_gate_top2(x: torch.Tensor, c: torch.Tensor, wg: torch.Tensor, noise: Optional[torch.Tensor], gates: torch.Tensor, idx: torch.Tensor) -> None:
    """top2gating's gate: gates [S, E] fp32, idx int32 [2, S] (first choice = argmax, second = argmax of logits + noise over the others)."""
    dt = _act(x, "x")
    _chk(c, "c", dt); _chk(wg, "wg", dt); _chk(gates, "gates", torch.float32); _chk(idx, "idx", torch.int32)
    S, D = x.shape
    E = wg.shape[0]
    assert x.stride(0) == c.stride(0) and wg.is_contiguous() and gates.is_contiguous() and idx.is_contiguous() and idx.shape == (2, S)
    if noise is not None:
        _chk(noise, "noise", torch.float32); assert noise.shape == (S, E) and noise.is_contiguous()
    L.check(_fn("ug_moe_gate_top2", dt)(x.data_ptr(), c.data_ptr(), x.stride(0), wg.data_ptr(), S, D, E, _p(noise), gates.data_ptr(), idx.data_ptr(), _stream()),
            "ug_moe_gate_top2")


def moe_capacity_top2(gates, idx, capacity: int, slot, token_of_slot, weights, exp_counts, l_aux) -> None:
    _chk(gates, "gates", torch.float32); _chk(idx, "idx", torch.int32); _chk(slot, "slot", torch.int32); _chk(token_of_slot, "token_of_slot", torch.int32)
    _chk(weights, "weights", torch.float32); _chk(exp_counts, "exp_counts", torch.int64); _chk(l_aux, "l_aux", torch.float32)
    S, E = gates.shape
    assert idx.shape == slot.shape == weights.shape == (2, S) and idx.is_contiguous() and slot.is_contiguous() and weights.is_contiguous()
    assert token_of_slot.numel() == E * capacity and token_of_slot.is_contiguous() and gates.is_contiguous()
    L.check(L.load().ug_moe_capacity_top2(gates.data_ptr(), idx.data_ptr(), S, E, capacity, slot.data_ptr(), token_of_slot.data_ptr(), weights.data_ptr(),
                                          exp_counts.data_ptr(), l_aux.data_ptr(), _stream()), "ug_moe_capacity_top2")


def moe_gate_topk(x: torch.Tensor, c: torch.Tensor, wg: torch.Tensor, K: int, gates: torch.Tensor, logits: torch.Tensor, idx: torch.Tensor) -> None:
    """topkgating's gate (k > 2): gates, logits [S, E] fp32, idx int32 [K, S] (the token's K choices by descending logit)."""
    dt = _act(x, "x")
    _chk(c, "c", dt); _chk(wg, "wg", dt); _chk(gates, "gates", torch.float32); _chk(logits, "logits", torch.float32); _chk(idx, "idx", torch.int32)
    S, D = x.shape
    E = wg.shape[0]
    assert c.shape == x.shape and wg.shape[1] == D and gates.shape == (S, E) and logits.shape == (S, E) and idx.shape == (K, S) and idx.is_contiguous()
    assert x.stride(0) == c.stride(0) and gates.is_contiguous() and logits.is_contiguous()
    L.check(_fn("ug_moe_gate_topk", dt)(x.data_ptr(), c.data_ptr(), x.stride(0), wg.data_ptr(), S, D, E, K, gates.data_ptr(), logits.data_ptr(), idx.data_ptr(),
                                        _stream()), "ug_moe_gate_topk")


def moe_capacity_topk(gates, logits, idx, capacity: int, slot, token_of_slot, weights, exp_counts, l_aux) -> None:
    """topkgating's capacity rule + combine weights + l_aux (include/unigen_hip.h): idx / slot / weights [K, S]."""
    S, E = gates.shape
    K = idx.shape[0]
    for t, dt in ((gates, torch.float32), (logits, torch.float32), (idx, torch.int32), (slot, torch.int32), (token_of_slot, torch.int32), (weights, torch.float32),
                  (exp_counts, torch.int64), (l_aux, torch.float32)):
        assert t.is_cuda and t.dtype == dt and t.is_contiguous()
    assert logits.shape == (S, E) and idx.shape == (K, S) and slot.shape == (K, S) and weights.shape == (K, S) and token_of_slot.shape == (E, capacity)
    L.check(L.load().ug_moe_capacity_topk(gates.data_ptr(), logits.data_ptr(), idx.data_ptr(), S, E, K, capacity, slot.data_ptr(), token_of_slot.data_ptr(),
                                          weights.data_ptr(), exp_counts.data_ptr(), l_aux.data_ptr(), _stream()), "ug_moe_capacity_topk")


def moe_combine_topk(yh, yc, weights, idx, slot, out, *, E: int, capacity: int, xs=None, cs=None, s_map: RowMap = IDENT, accumulate: bool = False) -> torch.Tensor:
    """weights fp32 / idx / slot [K, S'] (S' >= S: column slices of longer arrays keep their row stride); otherwise as moe_combine."""
    dt = _act(yh, "yh")
    _chk(yc, "yc", dt); _chk(out, "out", dt); _chk(weights, "weights", torch.float32); _chk(idx, "idx", torch.int32); _chk(slot, "slot", torch.int32)
    S, D = out.shape[-2], out.shape[-1]
    K = weights.shape[0]
    ks = weights.stride(0) if K > 1 else max(S, 1)
    assert weights.shape == idx.shape == slot.shape == (K, S) and weights.stride(1) == idx.stride(1) == slot.stride(1) == 1
    assert K == 1 or (idx.stride(0) == ks and slot.stride(0) == ks)
    if xs is not None:
        _chk(xs, "xs", dt); _chk(cs, "cs", dt); assert xs.stride(-2) == cs.stride(-2)
    L.check(_fn("ug_moe_combine_topk", dt)(yh.data_ptr(), yc.data_ptr(), weights.data_ptr(), idx.data_ptr(), slot.data_ptr(), K, ks, E, capacity, _p(xs), _p(cs),
                                            xs.stride(-2) if xs is not None else 0, s_map.rpb, s_map.bstride, out.data_ptr(), out.stride(-2), S, D,
                                            1 if accumulate else 0, _stream()), "ug_moe_combine_topk")
    return out


def moe_dispatch_modulate(x, add_, mod, token_of_slot, out, *, E: int, capacity: int, tokens_per_sample: int,
                          mod_estride: int = 0, mod_bstride: int = 0) -> torch.Tensor:
    """mod: the expert-modulation vectors, row of (expert e, sample b) at element offset e * mod_estride + b * mod_bstride."""
    dt = _act(x, "x")
    _chk(out, "out", dt); _chk(token_of_slot, "token_of_slot", torch.int32)
    D = x.shape[-1]
    assert out.is_contiguous() and out.numel() == E * capacity * D
    if mod is not None:
        _chk(mod, "mod", dt)
    if add_ is not None:
        _chk(add_, "add", dt); assert add_.is_contiguous()
    L.check(_fn("ug_moe_dispatch_modulate", dt)(x.data_ptr(), x.stride(-2), _p(add_), _p(mod), mod_estride, mod_bstride, token_of_slot.data_ptr(),
                                                 E, capacity, tokens_per_sample, D, out.data_ptr(), _stream()), "ug_moe_dispatch_modulate")
    return out


def moe_combine(yh, yc, gates, idx, slot, out, *, E: int, capacity: int, xs=None, cs=None, s_map: RowMap = IDENT, accumulate: bool = False) -> torch.Tensor:
    """xs / cs: base tensors of the shared experts' image / condition streams; token s is row s_map(s) of them."""
    dt = _act(yh, "yh")
    _chk(yc, "yc", dt); _chk(out, "out", dt)
    S, D = out.shape[-2], out.shape[-1]
    if xs is not None:
        _chk(xs, "xs", dt); _chk(cs, "cs", dt); assert xs.stride(-2) == cs.stride(-2)
    L.check(_fn("ug_moe_combine", dt)(yh.data_ptr(), yc.data_ptr(), gates.data_ptr(), idx.data_ptr(), slot.data_ptr(), E, capacity, _p(xs), _p(cs),
                                       xs.stride(-2) if xs is not None else 0, s_map.rpb, s_map.bstride, out.data_ptr(), out.stride(-2), S, D,
                                       1 if accumulate else 0, _stream()), "ug_moe_combine")
    return out


def pack_latents(latents: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """FluxPipeline._pack_latents: [B, C, H, W] -> [B, (H/2)(W/2), 4C] (src/UniGenPipeline.py:641)."""
    dt = _act(latents, "latents")
    B, Cc, H, W = latents.shape
    latents = latents.contiguous()
    if out is None:
        out = torch.empty(B, (H // 2) * (W // 2), Cc * 4, device=latents.device, dtype=dt)
    _chk(out, "out", dt); assert out.is_contiguous() and out.numel() == latents.numel()
    L.check(_fn("ug_pack_latents", dt)(latents.data_ptr(), out.data_ptr(), B, Cc, H, W, _stream()), "ug_pack_latents")
    return out


def unpack_latents(packed: torch.Tensor, H: int, W: int, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """FluxPipeline._unpack_latents: [B, (H/2)(W/2), 4C] -> [B, C, H, W] (H, W = latent height / width; src/UniGenPipeline.py:796)."""
    dt = _act(packed, "packed")
    B, n, ch = packed.shape
    assert n == (H // 2) * (W // 2) and ch % 4 == 0, (packed.shape, H, W)
    packed = packed.contiguous()
    if out is None:
        out = torch.empty(B, ch // 4, H, W, device=packed.device, dtype=dt)
    _chk(out, "out", dt); assert out.is_contiguous()
    L.check(_fn("ug_unpack_latents", dt)(packed.data_ptr(), out.data_ptr(), B, ch // 4, H, W, _stream()), "ug_unpack_latents")
    return out


def probe_mfma_peak(device, shape: int = 1, seconds: float = 0.4) -> float:
    """Measured dense bf16 MFMA rate (TFLOP/s) of a bare register-operand MFMA loop on every CU (ug_probe_mfma_bf16), timed with HIP
    events on the current stream over ~`seconds` of back-to-back launches so the chip settles at the clock it holds under matrix load.
    shape 0 = 32x32x16, 1 = 16x16x32 (the shape the GEMM uses)."""
    props = torch.cuda.get_device_properties(device)
    blocks = props.multi_processor_count
    scratch = torch.empty(blocks * 256, dtype=torch.float32, device=device)
    fl = C.c_double(0.0)
    iters = 100_000
    lib = L.load()
    launch = lambda: L.check(lib.ug_probe_mfma_bf16(shape, blocks, iters, scratch.data_ptr(), C.byref(fl), _stream()), "ug_probe_mfma_bf16")
    launch()
    torch.cuda.synchronize(device)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); launch(); e1.record(); e1.synchronize()
    n = max(3, int(seconds * 1e3 / max(e0.elapsed_time(e1), 1e-3)))
    for _ in range(n):          # settle the clock
        launch()
    e0.record()
    for _ in range(n):
        launch()
    e1.record(); e1.synchronize()
    return fl.value * n / (e0.elapsed_time(e1) * 1e-3) / 1e12


# ---------------------------------------------------------------------------------------------------------------------
# AutoencoderKL ops (unigen_amd/vae.py): NHWC activations as 2-D tensors [B * H * W, C]
# ---------------------------------------------------------------------------------------------------------------------
_zero_pages: dict = {}


def _zero_page(device) -> torch.Tensor:
    z = _zero_pages.get(device)
    if z is None:
        z = torch.zeros(4096, dtype=torch.uint8, device=device)      # >= 2 * (Cin + 64) bytes for Cin <= 1984: the 256^2 convolution path
        _zero_pages[device] = z
    return z


def conv2d_nhwc(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], out: torch.Tensor, *, B: int, H: int, W: int, Ho: int, Wo: int, KH: int, KW: int,
                stride: int = 1, pad_t: int = 1, pad_l: int = 1, up: int = 0, residual: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out[B*Ho*Wo, Cout] = residual + conv(x[B*H*W, Cin]; w[Cout, KH, KW, Cin]) + bias (ug_conv2d_nhwc: implicit GEMM, see include/unigen_hip.h)."""
    dt = _act(x, "x")
    _chk(w, "w", dt); _chk(out, "out", dt)
    Cin, Cout = x.shape[-1], out.shape[-1]
    assert x.is_contiguous() and w.is_contiguous() and out.is_contiguous() and w.numel() == Cout * KH * KW * Cin, (x.shape, w.shape, out.shape)
    assert x.numel() == B * H * W * Cin and out.numel() == B * Ho * Wo * Cout
    d = L.ConvDesc()
    d.x, d.B, d.H, d.W, d.Cin = x.data_ptr(), B, H, W, Cin
    d.w, d.bias = w.data_ptr(), _p(bias)
    if bias is not None:
        _chk(bias, "bias", dt)
    if residual is not None:
        _chk(residual, "residual", dt); assert residual.is_contiguous() and residual.numel() == out.numel()
        d.R = residual.data_ptr()
    d.out, d.Ho, d.Wo, d.Cout = out.data_ptr(), Ho, Wo, Cout
    d.KH, d.KW, d.stride, d.pad_t, d.pad_l, d.up = KH, KW, stride, pad_t, pad_l, up
    zp = _zero_page(x.device)
    d.zero_page, d.zero_page_bytes = zp.data_ptr(), zp.numel()
    L.check(_fn("ug_conv2d_nhwc", dt)(C.byref(d), _stream()), "ug_conv2d_nhwc")
    return out


_gn_ws: dict = {}


def groupnorm_nhwc(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, out: torch.Tensor, *, B: int, HW: int, groups: int, eps: float = 1e-6,
                   silu: bool = False) -> torch.Tensor:
    dt = _act(x, "x")
    _chk(gamma, "gamma", dt); _chk(beta, "beta", dt); _chk(out, "out", dt)
    Cc = x.shape[-1]
    assert x.is_contiguous() and out.is_contiguous() and x.numel() == B * HW * Cc
    need = int(L.load().ug_groupnorm_workspace_bytes(B, HW, groups))
    key = (x.device, _stream())
    ws = _gn_ws.get(key)
    if ws is None or ws.numel() < need:
        ws = torch.empty(need, dtype=torch.uint8, device=x.device)
        _gn_ws[key] = ws
    L.check(_fn("ug_groupnorm_nhwc", dt)(x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), out.data_ptr(), ws.data_ptr(), ws.numel(), B, HW, Cc, groups, eps,
                                          1 if silu else 0, _stream()), "ug_groupnorm_nhwc")
    return out


def softmax_rows(scores: torch.Tensor, probs: torch.Tensor, scale: float) -> torch.Tensor:
    """probs[r] = softmax(scale * scores[r]); scores fp32 [rows, cols], probs bf16 (or fp32: verification) [rows, cols]."""
    _chk(scores, "scores", torch.float32)
    dt = _act(probs, "probs")
    rows, cols = scores.shape
    L.check(_fn("ug_softmax_rows", dt)(scores.data_ptr(), scores.stride(0), probs.data_ptr(), probs.stride(0), rows, cols, scale, _stream()), "ug_softmax_rows")
    return probs


def nchw_to_nhwc(x: torch.Tensor, Cp: int, div: float = 0.0, add: float = 0.0) -> torch.Tensor:
    """[B, C, H, W] -> [B*H*W, Cp] (channels >= C zero); div != 0: v -> (v / div) + add on the way (latents / scaling_factor + shift_factor)."""
    dt = _act(x, "x")
    B, Cc, H, W = x.shape
    x = x.contiguous()
    out = torch.empty(B * H * W, Cp, device=x.device, dtype=dt)
    L.check(_fn("ug_nchw_to_nhwc", dt)(x.data_ptr(), out.data_ptr(), B, Cc, H * W, Cp, div, add, _stream()), "ug_nchw_to_nhwc")
    return out


def nhwc_to_nchw(x: torch.Tensor, B: int, Cc: int, H: int, W: int) -> torch.Tensor:
    """[B*H*W, Cp] -> [B, C, H, W] (first C channels)."""
    dt = _act(x, "x")
    assert x.is_contiguous() and x.shape[0] == B * H * W and x.shape[1] >= Cc
    out = torch.empty(B, Cc, H, W, device=x.device, dtype=dt)
    L.check(_fn("ug_nhwc_to_nchw", dt)(x.data_ptr(), out.data_ptr(), B, Cc, H * W, x.shape[1], _stream()), "ug_nhwc_to_nchw")
    return out


def vae_sample(moments: torch.Tensor, noise: torch.Tensor, *, B: int, latent: int, H: int, W: int, shift: float = 0.0, scale: float = 1.0) -> torch.Tensor:
    """z [B, L, H, W] = ((mean + exp(0.5 clamp(logvar)) * noise) - shift) * scale from NHWC moments [B*H*W, Cp >= 2L] and NCHW noise."""
    dt = _act(moments, "moments")
    _chk(noise, "noise", dt)
    assert moments.is_contiguous() and noise.is_contiguous() and noise.numel() == B * latent * H * W
    z = torch.empty(B, latent, H, W, device=moments.device, dtype=dt)
    L.check(_fn("ug_vae_sample", dt)(moments.data_ptr(), moments.shape[1], noise.data_ptr(), z.data_ptr(), B, latent, H * W, shift, scale, _stream()), "ug_vae_sample")
    return z


# ---------------------------------------------------------------------------------------------------------------------
# backward-pass entry points (include/unigen_hip.h "Backward pass of the control-module training step")
# ---------------------------------------------------------------------------------------------------------------------
def transpose(src: torch.Tensor, rows_pad: Optional[int] = None) -> torch.Tensor:
    """src [..., rows, cols] (any row stride, uniform batch stride) -> [..., cols, rows_pad] contiguous, zero beyond `rows`."""
    dt = _act(src, "src")
    rows, cols = src.shape[-2], src.shape[-1]
    rows_pad = rows if rows_pad is None else rows_pad
    lead = src.shape[:-2]
    batch = 1
    for d in lead:
        batch *= d
    if len(lead) > 1:
        raise ValueError("transpose: at most one batch dimension")
    bstride = src.stride(0) if lead else 0
    out = torch.empty(*lead, cols, rows_pad, device=src.device, dtype=dt)
    L.check(_fn("ug_transpose", dt)(src.data_ptr(), src.stride(-2), bstride, out.data_ptr(), rows_pad, cols * rows_pad, batch, rows, cols, rows_pad, _stream()),
            "ug_transpose")
    return out


def colsum(a: torch.Tensor, b: Optional[torch.Tensor] = None, *, rows_per_group: Optional[int] = None, alpha: float = 1.0) -> torch.Tensor:
    """a (and b) [rows, cols] -> [rows / rows_per_group, cols]: per-group column sums of a (* b), fp32 accumulation."""
    dt = _act(a, "a")
    rows, cols = a.shape
    g = rows if rows_per_group is None else rows_per_group
    if b is not None:
        _chk(b, "b", dt)
    out = torch.empty(rows // g, cols, device=a.device, dtype=dt)
    ws = torch.empty(int(L.load().ug_colsum_workspace_bytes(rows, cols, g)), device=a.device, dtype=torch.uint8)
    L.check(_fn("ug_colsum", dt)(a.data_ptr(), a.stride(0), _p(b), b.stride(0) if b is not None else 0, out.data_ptr(), cols, rows, cols, g, alpha,
                                 ws.data_ptr(), ws.numel(), _stream()), "ug_colsum")
    return out


def gelu_tanh(x: torch.Tensor) -> torch.Tensor:
    dt = _act(x, "x")
    x = x.contiguous()
    out = torch.empty_like(x)
    L.check(_fn("ug_gelu_tanh", dt)(x.data_ptr(), out.data_ptr(), x.numel(), _stream()), "ug_gelu_tanh")
    return out


def gelu_tanh_bwd(x: torch.Tensor, dy: torch.Tensor) -> torch.Tensor:
    dt = _act(x, "x")
    x, dy = x.contiguous(), dy.contiguous()
    _chk(dy, "dy", dt)
    out = torch.empty_like(x)
    L.check(_fn("ug_gelu_tanh_bwd", dt)(x.data_ptr(), dy.data_ptr(), out.data_ptr(), x.numel(), _stream()), "ug_gelu_tanh_bwd")
    return out


def adaln_modulate_bwd(x: torch.Tensor, dy: torch.Tensor, scale: torch.Tensor, *, rows_per_sample: int, eps: float = 1e-6):
    """x, dy [rows, D]; scale [samples, D] (row stride = its stride(0)) -> (dx [rows, D], d shift [samples, D], d scale [samples, D])."""
    dt = _act(x, "x")
    _chk(dy, "dy", dt); _chk(scale, "scale", dt)
    rows, D = x.shape
    samples = rows // rows_per_sample
    dx = torch.empty(rows, D, device=x.device, dtype=dt)
    part = torch.empty(samples, int(L.load().ug_adaln_modulate_bwd_partials(rows, rows_per_sample)), 2, D, device=x.device, dtype=torch.float32)
    L.check(_fn("ug_adaln_modulate_bwd", dt)(x.data_ptr(), x.stride(0), dy.data_ptr(), dy.stride(0), scale.data_ptr(), scale.stride(0), rows_per_sample,
                                           dx.data_ptr(), D, part.data_ptr(), rows, D, eps, _stream()), "ug_adaln_modulate_bwd")
    sums = part.sum(1).to(dt) if part.shape[1] > 1 else part[:, 0].to(dt)         # the kernel's per-sample partials, added in a fixed order
    return dx, sums[:, 0], sums[:, 1]


def qk_rmsnorm_rope_bwd(x: torch.Tensor, dy: torch.Tensor, w: Optional[torch.Tensor], cos: Optional[torch.Tensor], sin: Optional[torch.Tensor], *,
                        rows_per_batch: int, pos_offset: int, heads: int, dh: int, eps: float = 1e-6):
    """x, dy [rows, heads * dh] (any row stride) -> (dx [rows, heads * dh], d weight [dh] in x's dtype or None)."""
    dt = _act(x, "x")
    _chk(dy, "dy", dt)
    rows = x.shape[0]
    dx = torch.empty(rows, heads * dh, device=x.device, dtype=dt)
    dwx = torch.empty(int(L.load().ug_qk_rmsnorm_rope_bwd_partials(rows, heads)), dh, device=x.device, dtype=torch.float32) if w is not None else None
    if cos is not None:
        _chk(cos, "cos", torch.float32); _chk(sin, "sin", torch.float32)
        assert cos.shape[0] >= pos_offset + rows_per_batch and cos.shape[1] == dh and cos.is_contiguous() and sin.is_contiguous()
    L.check(_fn("ug_qk_rmsnorm_rope_bwd", dt)(x.data_ptr(), x.stride(0), dy.data_ptr(), dy.stride(0), dx.data_ptr(), heads * dh, _p(dwx), _p(w), _p(cos),
                                            _p(sin), rows, rows_per_batch, pos_offset, heads, dh, eps, _stream()), "ug_qk_rmsnorm_rope_bwd")
    return dx, (dwx.sum(0).to(dt) if dwx is not None else None)       # <= 2048 fp32 partial rows of the kernel, added in a fixed order


def row_lse(S: torch.Tensor, scale: float, valid_cols: Optional[int] = None) -> torch.Tensor:
    _chk(S, "S", torch.float32)
    rows, cols = S.shape
    out = torch.empty(rows, device=S.device, dtype=torch.float32)
    L.check(L.load().ug_row_lse(S.data_ptr(), S.stride(0), out.data_ptr(), rows, valid_cols or cols, scale, _stream()), "ug_row_lse")
    return out


def attn_prob(S: torch.Tensor, lse: torch.Tensor, scale: float, dtype, valid_cols: Optional[int] = None) -> torch.Tensor:
    rows, cols = S.shape
    P = torch.empty(rows, cols, device=S.device, dtype=dtype)
    L.check(_fn("ug_attn_prob", dtype)(S.data_ptr(), S.stride(0), lse.data_ptr(), P.data_ptr(), cols, rows, cols, valid_cols or cols, scale, _stream()),
            "ug_attn_prob")
    return P


def attn_dscore(P: torch.Tensor, dP: torch.Tensor, delta: torch.Tensor, scale: float) -> torch.Tensor:
    dt = _act(P, "P")
    rows, cols = P.shape
    dS = torch.empty_like(P)
    L.check(_fn("ug_attn_dscore", dt)(P.data_ptr(), P.stride(0), dP.data_ptr(), dP.stride(0), delta.data_ptr(), dS.data_ptr(), cols, rows, cols, scale, _stream()),
            "ug_attn_dscore")
    return dS


def rowdot(a: torch.Tensor, b: torch.Tensor, groups: int) -> torch.Tensor:
    """a, b [rows, groups * cols] -> [groups, rows] fp32: per-group row dot products."""
    dt = _act(a, "a")
    _chk(b, "b", dt)
    rows = a.shape[0]
    cols = a.shape[1] // groups
    out = torch.empty(groups, rows, device=a.device, dtype=torch.float32)
    L.check(_fn("ug_rowdot", dt)(a.data_ptr(), a.stride(0), b.data_ptr(), b.stride(0), out.data_ptr(), rows, groups, cols, _stream()), "ug_rowdot")
    return out


def flash_attn_bwd(q, k, v, o, do, *, heads: int, lse: Optional[torch.Tensor] = None):
    """q, o, do [B, Lq, H * dh], k, v [B, Lkv, H * dh] (any row / batch strides, bf16) -> (dq, dk, dv) contiguous.
    lse: what flash_attn(..., lse=) wrote, fp32 [B, H, Lq rounded up to 64] with zero padding (None: recomputed)."""
    _chk(q, "q"); _chk(k, "k"); _chk(v, "v"); _chk(o, "o"); _chk(do, "do")
    B, Lq, HD = q.shape
    Lkv, dh = k.shape[1], HD // heads
    dq, dk, dv = torch.empty(B, Lq, HD, device=q.device, dtype=bf16), torch.empty(B, Lkv, HD, device=q.device, dtype=bf16), torch.empty(B, Lkv, HD, device=q.device, dtype=bf16)
    lib = L.load()
    ws = torch.empty(int(lib.ug_flash_attn_bwd_workspace_bytes(B, heads, Lq)), device=q.device, dtype=torch.uint8)
    st = lambda t: (t.data_ptr(), t.stride(1), t.stride(0))
    ev = _timer.begin("attn_bwd") if _timer is not None else None
    if lse is not None:
        assert lse.dtype == torch.float32 and lse.is_contiguous() and tuple(lse.shape) == (B, heads, (Lq + 63) // 64 * 64), lse.shape
    L.check(lib.ug_flash_attn_bwd(*st(q), *st(k), *st(v), *st(o), *st(do), *st(dq), *st(dk), *st(dv), B, heads, Lq, Lkv, dh, dh ** -0.5, _p(lse),
                                  ws.data_ptr(), ws.numel(), _stream()), "ug_flash_attn_bwd")
    if ev is not None:
        _timer.end("attn_bwd", 10.0 * B * heads * Lq * Lkv * dh, ev)
    return dq, dk, dv


def gate_residual(x: Optional[torch.Tensor], a: torch.Tensor, gate: torch.Tensor, rows_per_sample: int) -> torch.Tensor:
    """a (and x) [rows, D]; gate [samples, D] -> x + gate[row // rows_per_sample] * a (x None: the product alone)."""
    dt = _act(a, "a")
    _chk(gate, "gate", dt)
    if x is not None:
        _chk(x, "x", dt)
    rows, D = a.shape
    y = torch.empty(rows, D, device=a.device, dtype=dt)
    L.check(_fn("ug_gate_residual", dt)(_p(x), x.stride(0) if x is not None else 0, a.data_ptr(), a.stride(0), gate.data_ptr(), gate.stride(0), rows_per_sample,
                                        y.data_ptr(), D, rows, D, _stream()), "ug_gate_residual")
    return y
