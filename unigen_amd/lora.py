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


class LoRALinear(nn.Module):
    """nn.Linear-shaped layer (weight [N, K], bias [N]) with named LoRA adapters; the duck-typed stand-in for a PEFT BaseTunerLayer
    (`active_adapters`, `scaling`, `set_scale`, `lora_A`, `lora_B`)."""

    def __init__(self, in_features: int, out_features: int, bias: bool = True, device=None, dtype=BF):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        self.weight = nn.Parameter(torch.empty(out_features, in_features, device=device, dtype=dtype), requires_grad=False)
        self.bias = nn.Parameter(torch.zeros(out_features, device=device, dtype=dtype), requires_grad=False) if bias else None
        self.lora_A, self.lora_B = nn.ParameterDict(), nn.ParameterDict()
        self.r: Dict[str, int] = {}
        self.lora_alpha: Dict[str, float] = {}
        self.scaling: Dict[str, float] = {}
        self.active_adapters: List[str] = []

    def add_adapter(self, name: str, r: int, lora_alpha: float, A: Optional[torch.Tensor] = None, B: Optional[torch.Tensor] = None) -> None:
        dev, dt = self.weight.device, self.weight.dtype
        self.lora_A[name] = nn.Parameter((A if A is not None else torch.randn(r, self.in_features) * 0.01).to(dev, dt), requires_grad=False)
        self.lora_B[name] = nn.Parameter((B if B is not None else torch.zeros(self.out_features, r)).to(dev, dt), requires_grad=False)
        self.r[name], self.lora_alpha[name] = r, float(lora_alpha)
        self.scaling[name] = float(lora_alpha) / r
        if name not in self.active_adapters:
            self.active_adapters.append(name)

    def set_scale(self, adapter: str, scale: float) -> None:          # peft LoraLayer.set_scale
        if adapter in self.scaling:
            self.scaling[adapter] = scale * self.lora_alpha[adapter] / self.r[adapter]

    def _fused_adapters(self):
        """Concatenate the active adapters along the rank: A_cat [R, K], B_cat [N, R] with scaling folded into B, R padded to 64."""
        names = [a for a in self.active_adapters if a in self.scaling and self.scaling[a] != 0.0]
        if not names:
            return None, None
        A = torch.cat([self.lora_A[a] for a in names], 0)
        Bm = torch.cat([(self.lora_B[a].float() * self.scaling[a]).to(self.weight.dtype) for a in names], 1)
        R = A.shape[0]
        Rp = (R + _RANK_PAD - 1) // _RANK_PAD * _RANK_PAD
        if Rp != R:
            A = torch.cat([A, A.new_zeros(Rp - R, A.shape[1])], 0)
            Bm = torch.cat([Bm, Bm.new_zeros(Bm.shape[0], Rp - R)], 1)
        return A.contiguous(), Bm.contiguous()

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
        self.lora_modules = [each for each in lora_modules if isinstance(each, LoRALinear)]
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
