"""Per-condition LoRA switch on MI355X.

The reference ships `src/lora_switching_module.py` (`module_active_adapters`, `enable_lora`) on top of PEFT 0.15 tuner layers, but
never invokes it (imported, not called; inherited from UniCombine). This module supplies the same API over a PEFT-free
LoRA-capable linear whose forward is ONE fused GEMM in libunigen_hip.so:   y = x W^T + b + sum_a scaling_a * (x A_a^T) B_a^T
(the adapter product is a second K-segment of the base GEMM's accumulators, `ug_gemm_desc.lora_T / lora_B`).

Semantics restated from PEFT 0.15 `LoraLayer`: `scaling[a] = lora_alpha / r`; `set_scale(a, s)` sets `scaling[a] = s * lora_alpha / r`.
`enable_lora.__exit__` therefore "restores" through `set_scale(saved_scaling)`, i.e. multiplies by lora_alpha / r again
(idempotent only when alpha == r) - kept as the reference has it.
"""
from __future__ import annotations

from typing import Any, Dict, List, Optional, Type

import torch
from torch import nn

from . import lib as L
from . import ops

BF = torch.bfloat16
_RANK_PAD = 64    # the GEMM's K-tile: adapter ranks are zero-padded to a multiple of it


class _AdapterWeight(nn.Module):
    """One side of one adapter: holds `.weight` so that the state-dict keys read `<proj>.lora_A.<adapter>.weight` / `<proj>.lora_B.<adapter>.weight`
    as PEFT's `ModuleDict` of `nn.Linear` does. Never called: the arithmetic is the K-segment of ug_gemm_bf16."""

    def __init__(self, w: torch.Tensor):
        super().__init__()
        self.weight = nn.Parameter(w, requires_grad=False)


class LoRALayer:
    """The duck-typed stand-in for a PEFT `BaseTunerLayer` / `LoraLayer` (peft 0.15 tuners/lora/layer.py): `lora_A`, `lora_B`, `r`, `lora_alpha`,
    `scaling`, `active_adapters`, `set_scale`. Mixed into `LoRALinear` (a stand-alone layer) and into the parameter holders of the engines'
    projections (`HipModule.add_lora`), so that `enable_lora(list(model.modules()), [...])` - the reference's call shape - finds them."""

    def _init_lora(self) -> None:
        self.lora_A, self.lora_B = nn.ModuleDict(), nn.ModuleDict()
        self.r: Dict[str, int] = {}
        self.lora_alpha: Dict[str, float] = {}
        self.scaling: Dict[str, float] = {}
        self.active_adapters: List[str] = []

    def add_adapter(self, name: str, r: int, lora_alpha: float, A: Optional[torch.Tensor] = None, B: Optional[torch.Tensor] = None,
                    generator: Optional[torch.Generator] = None) -> None:
        """peft `update_layer`: A [r, in] (default: uniform(-1/sqrt(in), 1/sqrt(in)), PEFT's kaiming_uniform(a = sqrt 5)), B [out, r] (default: zeros)."""
        w = self.weight
        N, K = w.shape
        if A is None:
            bound = 1.0 / (K ** 0.5)
            A = torch.empty(r, K, dtype=torch.float32).uniform_(-bound, bound, generator=generator)
        if B is None:
            B = torch.zeros(N, r)
        if tuple(A.shape) != (r, K) or tuple(B.shape) != (N, r):
            raise ValueError(f"adapter {name!r}: A must be [{r}, {K}] and B [{N}, {r}], got {tuple(A.shape)} / {tuple(B.shape)}")
        self.lora_A[name] = _AdapterWeight(A.detach().to(w.device, w.dtype).contiguous())
        self.lora_B[name] = _AdapterWeight(B.detach().to(w.device, w.dtype).contiguous())
        self.r[name], self.lora_alpha[name] = int(r), float(lora_alpha)
        self.scaling[name] = float(lora_alpha) / r
        if name not in self.active_adapters:
            self.active_adapters.append(name)

    def set_scale(self, adapter: str, scale: float) -> None:          # peft LoraLayer.set_scale
        if adapter in self.scaling:
            self.scaling[adapter] = scale * self.lora_alpha[adapter] / self.r[adapter]

    def scale_layer(self, scale: float) -> None:                      # peft LoraLayer.scale_layer (diffusers scale_lora_layers calls it)
        if scale == 1:
            return
        for a in self.active_adapters:
            if a in self.lora_A:
                self.scaling[a] *= scale

    def unscale_layer(self, scale=None) -> None:                      # peft LoraLayer.unscale_layer
        for a in self.active_adapters:
            if a not in self.lora_A:
                continue
            if scale is None:
                self.scaling[a] = self.lora_alpha[a] / self.r[a]
            else:
                self.scaling[a] /= scale

    def live_adapters(self) -> List[str]:
        """The adapters that contribute to this layer's output right now: active, known, scaling != 0."""
        return [a for a in self.active_adapters if a in self.scaling and self.scaling[a] != 0.0]


def fuse_adapters(layers: List[Optional["LoRALayer"]], widths: List[int], dtype, device):
    """The adapter operands of ONE GEMM launch over the row-concatenated weights of `layers` (None = a projection without adapters; widths = the
    projections' output widths):  A_cat [R, K] = every live adapter's A stacked along the rank, B_bd [sum widths, R] = block-diagonal, projection i's rows
    holding scaling_a * B_a in the columns of ITS adapters, zero elsewhere; R zero-padded to the GEMM's K-tile. (None, None) when nothing is live."""
    blocks, R = [], 0
    for i, lay in enumerate(layers):
        for a in (lay.live_adapters() if lay is not None else ()):
            blocks.append((i, lay.lora_A[a].weight.data, lay.lora_B[a].weight.data, lay.scaling[a], R))
            R += lay.r[a]
    if not blocks:
        return None, None
    Rp = (R + _RANK_PAD - 1) // _RANK_PAD * _RANK_PAD
    K = blocks[0][1].shape[1]
    A = torch.zeros(Rp, K, device=device, dtype=dtype)
    Bm = torch.zeros(sum(widths), Rp, device=device, dtype=dtype)
    row0 = [sum(widths[:i]) for i in range(len(widths))]
    for i, a_w, b_w, sc, c0 in blocks:
        r = a_w.shape[0]
        A[c0:c0 + r] = a_w
        Bm[row0[i]:row0[i] + widths[i], c0:c0 + r] = (b_w.float() * sc).to(dtype)
    return A, Bm


class LoRALinear(nn.Module, LoRALayer):
    """nn.Linear-shaped layer (weight [N, K], bias [N]) with named LoRA adapters, stand-alone (outside a transformer)."""

    def __init__(self, in_features: int, out_features: int, bias: bool = True, device=None, dtype=BF):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        self.weight = nn.Parameter(torch.empty(out_features, in_features, device=device, dtype=dtype), requires_grad=False)
        self.bias = nn.Parameter(torch.zeros(out_features, device=device, dtype=dtype), requires_grad=False) if bias else None
        self._init_lora()

    def _fused_adapters(self):
        """Concatenate the live adapters along the rank: A_cat [R, K], B_cat [N, R] with scaling folded into B, R padded to 64."""
        return fuse_adapters([self], [self.out_features], self.weight.dtype, self.weight.device)

    @torch.no_grad()
    def forward(self, x: torch.Tensor) -> torch.Tensor:
        shp = x.shape
        x2 = x.reshape(-1, shp[-1])
        M = x2.shape[0]
        dt = self.weight.dtype                                            # bf16 (product) or fp32 (verification twins, ug_gemm_f32)
        out = torch.empty(M, self.out_features, device=x.device, dtype=dt)
        A, Bm = self._fused_adapters()
        if A is None:
            ops.gemm(x2, self.weight, self.bias, out, M=M)
        else:
            t = torch.empty(M, A.shape[0], device=x.device, dtype=dt)
            ops.gemm(x2, A, None, t, M=M)                              # T = x A^T
            ops.gemm(x2, self.weight, self.bias, out, M=M, lora_t=t, lora_b=Bm)   # base GEMM + LoRA K-segment in one accumulator
        return out.view(*shp[:-1], self.out_features)


def module_active_adapters(module) -> List[str]:
    """src/lora_switching_module.py:4-9."""
    if hasattr(module, "active_adapters"):
        return [i for i in module.active_adapters if i in module.scaling.keys()]
    return []


class enable_lora:
    """src/lora_switching_module.py:11-38: inside the context every active adapter NOT in `enable_adapters` has its scale set to 0."""

    def __init__(self, lora_modules: List[Any], enable_adapters: List) -> None:
        self.lora_modules = [each for each in lora_modules if isinstance(each, LoRALayer)]      # PEFT: isinstance(each, BaseTunerLayer)
        self.active_adapter_scales = [{a: m.scaling[a] for a in module_active_adapters(m)} for m in self.lora_modules]
        self.enable_adapters = enable_adapters

    def __enter__(self) -> None:
        for m in self.lora_modules:
            for a in module_active_adapters(m):
                if a not in self.enable_adapters:
                    m.set_scale(a, 0)

    def __exit__(self, exc_type: Optional[Type[BaseException]], exc_val: Optional[BaseException], exc_tb: Optional[Any]) -> None:
        for i, m in enumerate(self.lora_modules):
            for a in module_active_adapters(m):
                m.set_scale(a, self.active_adapter_scales[i][a])
