"""Shared host-side machinery of the HIP engines (UniGenFlux, UniGenSD3): parameter containers under the reference's state-dict
names, weight packing into fused buffers (parameters become views), a workspace cache, token-stream descriptors and the
block routines that orchestrate calls through the C ABI (unigen_amd/ops.py). No arithmetic happens in torch here."""
from __future__ import annotations

import math
from typing import Dict, Optional, Sequence, Tuple

import torch
from torch import nn

from . import lib as L
from . import ops
from .lora import LoRALayer, fuse_adapters
from .ops import RowMap

BF = torch.bfloat16


class _Holder(nn.Module):
    """Parameter container; the arithmetic lives in libunigen_hip.so, so calling it is an error."""

    def forward(self, *a, **k):  # pragma: no cover
        raise L.UniGenHipError("parameter holder: unigen_amd modules are executed by the HIP engine, not called directly")


class _LoRAHolder(_Holder, LoRALayer):
    """The holder of a projection that carries LoRA adapters (HipModule.add_lora): the same `weight` / `bias` parameters plus
    `lora_A.<adapter>.weight`, `lora_B.<adapter>.weight` and the PEFT layer attributes `enable_lora` reads."""

    def __init__(self, plain: _Holder):
        super().__init__()
        for n, p_ in plain._parameters.items():         # the SAME Parameter objects: packed views, state-dict keys and optimiser references survive
            self.register_parameter(n, p_)
        for n, m in plain._modules.items():
            self.add_module(n, m)
        self._init_lora()


def _register(root: nn.Module, name: str, shape: Tuple[int, ...], device, dtype) -> nn.Parameter:
    parts = name.split(".")
    mod = root
    for part in parts[:-1]:
        child = mod._modules.get(part)
        if child is None:
            child = _Holder()
            mod.add_module(part, child)
        mod = child
    p = nn.Parameter(torch.empty(shape, device=device, dtype=dtype), requires_grad=False)
    mod.register_parameter(parts[-1], p)
    return p


def reference_init_(root: nn.Module, names: Sequence[str], copies: Dict[str, str], zero_prefixes: Sequence[str], expert_prefix: str) -> None:
    """Initial values of the control modules as the reference's constructors leave them (src/UniGenTransformer.py:727-773, 833-842, 26-128):
      * `copies` {dst prefix: src prefix}: `copy.deepcopy` of a base module (Flux: control_time_text_embed, control_condition_embed <-
        time_text_embed; control_x_embedder <- x_embedder, :731-742);
      * `zero_prefixes`: `zero_module(nn.Linear)` - the zero-res projections controlnet_add_* only (:757,771; :119-123 for SD3);
      * RMSNorm weights (1-D `*.weight`) = 1; every other `nn.Linear` / `nn.Conv2d` = torch's default reset_parameters():
        weight ~ U(-1/sqrt(fan_in), 1/sqrt(fan_in)) (kaiming_uniform_, a = sqrt(5)), bias ~ U(-1/sqrt(fan_in), 1/sqrt(fan_in));
      * deepspeed `Experts` deep-copies ONE expert module num_experts times: every expert starts from expert 0's values.
    Draws come from torch's global generator on the parameters' device, as the reference's do. (Round 2 zeroed all of it: a from-scratch
    train.py run then starts from a dead network - tied gate, q = k = 0, most gradients exactly zero.)"""
    params = dict(root.named_parameters())
    e0 = expert_prefix + "0."
    with torch.no_grad():
        for name in names:
            p = params[name]
            src = next((s + name[len(d):] for d, s in copies.items() if name.startswith(d)), None)
            if src is not None:
                p.copy_(params[src])
            elif any(name.startswith(z) for z in zero_prefixes):
                p.zero_()
            elif name.startswith(expert_prefix) and not name.startswith(e0):
                continue                                                    # filled from expert 0 below
            elif name.endswith(".weight") and p.dim() == 1:
                p.fill_(1.0)
            elif name.endswith(".weight"):
                bound = 1.0 / math.sqrt(p[0].numel())
                p.copy_(torch.empty(p.shape, device=p.device, dtype=torch.float32).uniform_(-bound, bound))
            elif name.endswith(".bias"):
                w = params.get(name[:-len("bias")] + "weight")
                bound = 1.0 / math.sqrt(w[0].numel()) if w is not None and w.dim() > 1 else 0.0
                p.copy_(torch.empty(p.shape, device=p.device, dtype=torch.float32).uniform_(-bound, bound))
            else:
                raise L.UniGenHipError(f"reference_init_: no rule for parameter {name}")
        for name in names:
            if name.startswith(expert_prefix) and not name.startswith(e0):
                rest = name[len(expert_prefix):].split(".", 1)[1]
                params[name].copy_(params[e0 + rest])


class _Workspace:
    def __init__(self):
        self._bufs: Dict[Tuple, torch.Tensor] = {}

    def get(self, name: str, shape: Sequence[int], dtype, device) -> torch.Tensor:
        key = (name, tuple(shape), dtype, str(device))
        t = self._bufs.get(key)
        if t is None:
            t = torch.empty(tuple(shape), dtype=dtype, device=device)
            self._bufs[key] = t
        return t

    def clear(self):
        self._bufs.clear()


class _Stream:
    """A token stream [B, Ls, D] living in some buffer: `base` is a 2-D view whose row 0 is token (0, 0); logical row
    m = b * Ls + r sits at physical row b * bstride + r (RowMap), leading dimension `ld`."""
    __slots__ = ("base", "ld", "map", "Ls")

    def __init__(self, base: torch.Tensor, Ls: int, bstride: Optional[int] = None):
        self.base, self.ld, self.Ls = base, base.stride(0), Ls
        self.map = RowMap() if (bstride is None or bstride == Ls) else RowMap(Ls, bstride)


class _Routing:
    """One gate decision in index form: K choices per token; idx / slot are [S] (K = 1) or [2, S]; tos [E, C]; weights [2, S] (K = 2)."""
    __slots__ = ("K", "E", "C", "gates", "idx", "slot", "tos", "weights", "exp_counts", "l_aux")

    def __init__(self, K, E, C, gates, idx, slot, tos, weights, exp_counts, l_aux):
        self.K, self.E, self.C, self.gates, self.idx, self.slot, self.tos = K, E, C, gates, idx, slot, tos
        self.weights, self.exp_counts, self.l_aux = weights, exp_counts, l_aux


class HipModule(nn.Module):
    """Base of the drop-in transformer classes: holds parameters, never computes in torch."""

    inner_dim: int
    _heads: int
    _head_dim: int

    def __init__(self):
        super().__init__()
        self._ws = _Workspace()
        self._packed: Dict[str, torch.Tensor] = {}
        self._emb_tab: Dict[str, torch.Tensor] = {}       # AdaLN linear outputs of the current step, by module prefix (_adaln_group)
        self.trainable_control_modules: Dict[str, nn.Module] = {}
        self._lora_sites: Dict[str, _LoRAHolder] = {}     # projection prefix -> its adapter-carrying holder (add_lora)
        self._lora_fused: Dict[Tuple, Tuple] = {}         # fused adapter operands per launch (prefixes + adapter state), see _lora_operands

    def _check_dtype(self, *inputs: torch.Tensor):
        """bf16 = the product path; fp32 = the verification path (every C-ABI call goes to its `_f32` twin, unigen_amd/ops.py).
        Returns the activation dtype; inputs are cast to it by the caller (bf16 -> fp32 is exact)."""
        dt = self.dtype
        if dt not in (BF, torch.float32):
            raise TypeError("the HIP engine computes in bf16 (or fp32 for verification): cast the model to torch.bfloat16")
        for t in inputs:
            if t is not None and t.dtype not in (BF, torch.float32):
                raise TypeError(f"inputs must be torch.bfloat16 (or torch.float32), got {t.dtype}")
        return dt

    def _probe(self) -> torch.Tensor:
        return next(self.parameters())

    @property
    def dtype(self):
        return self._probe().dtype

    @property
    def device(self):
        return self._probe().device

    def init_trainable_param(self):
        for module in self.trainable_control_modules.values():
            module.requires_grad_(True)

    def enable_gradient_checkpointing(self):
        """train.py:317: transformer blocks are recomputed in the backward (torch.utils.checkpoint around each block of unigen_amd/training.py)."""
        self._grad_checkpoint = True

    def disable_gradient_checkpointing(self):
        self._grad_checkpoint = False

    def enable_xformers_memory_efficient_attention(self, *args, **kwargs):
        """train.py:337 (behind `--enable_xformers_memory_efficient_attention`): nothing to switch - attention here is always the fused flash kernel
        (ug_flash_attn_fwd / _bwd); kept so that the flag does not turn into an AttributeError."""
        return None

    def init_synthetic_(self, seed: int = 0, std: float = 0.02, bias_std: float = 0.0) -> "HipModule":
        """Seeded N(0, std^2) weights, zero (or N(0, bias_std^2)) biases, unit RMSNorm weights; the zero-res projections are
        randomised too so the control path contributes. Generated on the parameters' device."""
        g = torch.Generator(device=self.device).manual_seed(seed)
        with torch.no_grad():
            for name, p in self.named_parameters():
                if ".norm_q." in name or ".norm_k." in name or ".norm_added_q." in name or ".norm_added_k." in name:
                    p.fill_(1.0)
                elif name.endswith(".bias"):
                    if bias_std > 0:
                        p.copy_(torch.randn(p.shape, generator=g, device=p.device, dtype=torch.float32) * bias_std)
                    else:
                        p.zero_()
                else:
                    # chunked to bound the fp32 temporary for the 12288 x 3072 matrices
                    p.copy_(torch.randn(p.shape, generator=g, device=p.device, dtype=torch.float32) * std)
        return self

    # ------------------------------------------------------------------ LoRA (A12) -------------------------------------
    # The projections whose GEMM launches take the adapter K-segment (every attention / feed-forward projection of a joint or single block, in
    # the base, control, shared-expert and consistency blocks alike). PEFT matches `target_modules` by module-name suffix; a suffix that hits a
    # parameter the engine runs another way (AdaLN linears, embedders, experts, zero-res projections) is refused rather than silently ignored.
    LORA_CAPABLE = ("attn.to_q", "attn.to_k", "attn.to_v", "attn.add_q_proj", "attn.add_k_proj", "attn.add_v_proj", "attn.to_out.0", "attn.to_add_out",
                    "attn2.to_q", "attn2.to_k", "attn2.to_v", "attn2.to_out.0", "ff.net.0.proj", "ff.net.2", "ff_context.net.0.proj", "ff_context.net.2",
                    "proj_mlp", "proj_out")

    def add_lora(self, target_suffixes: Sequence[str], adapter_name: str, r: int, lora_alpha: float, A=None, B=None, prefix: str = "",
                 init_lora_weights: bool = True, seed: int = 0) -> Sequence[str]:
        """Attach the LoRA adapter `adapter_name` (rank r, scaling lora_alpha / r) to every projection whose module name ends with one of
        `target_suffixes` (PEFT's `target_modules` rule: name == key or name.endswith("." + key)) and starts with `prefix` (e.g.
        "control_joint_trans_blocks." for the per-condition adapters of the control branch). The projection's holder becomes a PEFT-shaped layer
        (`lora_A`, `lora_B`, `scaling`, `active_adapters`, `set_scale`) that `enable_lora(list(model.modules()), [...])`
        (src/lora_switching_module.py:11-38) switches; the engines' GEMM launches for that weight take the live adapters as a K-segment of the base
        product (`ug_gemm_desc.lora_T / lora_B`). A / B: None (PEFT's initial values: A uniform, B zero - or B random too with
        init_lora_weights=False, PEFT's testing mode), or {module name: tensor}. Returns the module names that received the adapter.
        No adapter attached -> nothing changes anywhere (bit-identical outputs, same launches)."""
        hits = []
        for name, mod in list(self.named_modules()):
            if not name.startswith(prefix) or not any(name == t or name.endswith("." + t) for t in target_suffixes):
                continue
            w = mod._parameters.get("weight")
            if w is None or w.dim() != 2:
                continue
            if not any(name == c or name.endswith("." + c) for c in self.LORA_CAPABLE) or name == "proj_out" or ".experts." in name:
                raise L.UniGenHipError(f"add_lora: {name} is not a projection the HIP engine can extend with an adapter K-segment "
                                       f"(supported suffixes: {', '.join(self.LORA_CAPABLE)} inside transformer blocks)")
            hits.append(name)
        if not hits:
            raise ValueError(f"add_lora: no module under prefix {prefix!r} matches {list(target_suffixes)}")
        g = torch.Generator().manual_seed(seed)
        for name in hits:
            parent_name, _, leaf = name.rpartition(".")
            parent = self.get_submodule(parent_name) if parent_name else self
            mod = parent._modules[leaf]
            if not isinstance(mod, _LoRAHolder):
                mod = _LoRAHolder(mod)
                parent._modules[leaf] = mod
                self._lora_sites[name] = mod
            a = A.get(name) if isinstance(A, dict) else A
            b = B.get(name) if isinstance(B, dict) else B
            if b is None and not init_lora_weights:
                b = torch.randn(mod.weight.shape[0], r, generator=g) * 0.02
            mod.add_adapter(adapter_name, r, lora_alpha, A=a, B=b, generator=g)
        self._pname_cache = None
        self._lora_fused.clear()
        return hits

    def _lora_scaled(self, joint_attention_kwargs):
        """`scale_lora_layers(self, lora_scale)` ... `unscale_lora_layers(self, lora_scale)` around a forward (src/UniGenTransformer.py:1200-1208,
        1266-1269; diffusers 0.32.2 utils/peft_utils.py): `joint_attention_kwargs["scale"]` multiplies every adapter's scaling for the duration of
        the forward; 1.0 (or no adapter-carrying projection) is a no-op, 0 is undone by `set_scale(adapter, 1.0)`."""
        import contextlib

        @contextlib.contextmanager
        def cm():
            w = (joint_attention_kwargs or {}).get("scale", 1.0)
            on = bool(self._lora_sites) and w is not None and w != 1.0
            if on:
                for m in self._lora_sites.values():
                    m.scale_layer(w)
            try:
                yield
            finally:
                if on:
                    for m in self._lora_sites.values():
                        if w != 0:
                            m.unscale_layer(w)
                        else:
                            for a in m.active_adapters:
                                m.set_scale(a, 1.0)
        return cm()

    def _refuse_lora_in_training(self):
        if self._lora_sites and any(m.live_adapters() for m in self._lora_sites.values()):
            raise NotImplementedError("the differentiable forward (unigen_amd/training.py) does not carry LoRA adapters: run under torch.no_grad(), "
                                      "or disable them (enable_lora(modules, []))")

    def _lora_live(self, prefixes: Sequence[str]) -> bool:
        """Does any projection of this launch carry an adapter that contributes right now?"""
        if not self._lora_sites:
            return False
        return any(p in self._lora_sites and self._lora_sites[p].live_adapters() for p in prefixes)

    def _lora_operands(self, prefixes: Sequence[str], a: torch.Tensor, M: int, tag: str, lda: Optional[int] = None, a_map: RowMap = ops.IDENT):
        """(T, B) for ONE launch over the row-concatenated weights of `prefixes` (the fused q|k|v[|proj_mlp] launch, or a single projection):
        T = a A_cat^T [M, R] - ONE GEMM for all live adapters of all projections of the launch - and the block-diagonal, scaling-folded B [N, R]
        (lora.fuse_adapters). (None, None) when no adapter is live: the caller's launch is then exactly the adapter-free one."""
        if not self._lora_live(prefixes):
            return None, None
        sites = [self._lora_sites.get(p) for p in prefixes]
        state = tuple((p, tuple((n, s.scaling[n], s.lora_A[n].weight._version, s.lora_B[n].weight._version, s.lora_A[n].weight.data_ptr())
                                for n in s.live_adapters())) for p, s in zip(prefixes, sites) if s is not None)
        key = (tuple(prefixes), self.dtype)
        hit = self._lora_fused.get(key)
        if hit is None or hit[0] != state:
            widths = [self.get_parameter(p + ".weight").shape[0] for p in prefixes]
            hit = (state,) + fuse_adapters(sites, widths, self.dtype, self.device)
            self._lora_fused[key] = hit
        A_cat, B_bd = hit[1], hit[2]
        t = self._w("lora_t_" + tag, (M, A_cat.shape[0]))
        ops.gemm(a, A_cat, None, t, M=M, lda=lda, a_map=a_map)
        return t, B_bd

    # ------------------------------------------------------------------ weight packing --------------------------------
    @staticmethod
    def _storage_moved():
        """Parameters were re-pointed at new storage: the backward's cached weight transposes (autograd._wt_cache, keyed by address) are stale."""
        from . import autograd
        autograd.clear_caches()

    def _apply(self, fn, *a, **k):
        r = super()._apply(fn, *a, **k)          # .to() / .cuda() / .float(): every parameter gets new storage
        self._storage_moved()
        self._lora_fused.clear()
        return r

    def _P(self, name: str) -> torch.Tensor:
        return self.get_parameter(name).data

    def _is_packed(self, t: Optional[torch.Tensor], params, stacked: bool) -> bool:
        """EVERY member still a view of `t` at its own offset (another grouping that re-pointed a later member must not pass as packed)."""
        if t is None or params[0].device != t.device:
            return False
        off = 0
        esz = t.element_size()
        for i, p in enumerate(params):
            want = t.data_ptr() + (i * t.stride(0) if stacked else off * t.stride(0)) * esz
            if p.data.data_ptr() != want or not p.data.is_contiguous():
                return False
            off += p.data.shape[0]
        return True

    def _drop_superseded(self, key: str, params) -> None:
        """A new pack of these parameters replaces every older pack that still holds one of them (a second grouping of the same weights would
        otherwise keep a stale duplicate of their storage alive in self._packed)."""
        ptrs = {p.data.untyped_storage().data_ptr() for p in params}
        dropped = {t.untyped_storage().data_ptr() for k, t in self._packed.items() if k != key and t.untyped_storage().data_ptr() in ptrs}
        if not dropped:
            return
        for k in [k for k, t in self._packed.items() if k != key and t.untyped_storage().data_ptr() in dropped]:
            del self._packed[k]
        # members of a dropped pack that the new grouping does not take (proj_mlp after a three-way q|k|v re-pack of the four-way pack) would keep
        # the WHOLE old buffer alive next to the new copy: give them storage of their own
        mine = {id(p) for p in params}
        for p in self.parameters():
            if id(p) not in mine and p.data.untyped_storage().data_ptr() in dropped:
                p.data = p.data.clone()

    def _pack(self, key: str, names: Sequence[str]) -> torch.Tensor:
        """Concatenate parameters along dim 0 into one buffer and re-point them at views of it (no duplicate storage)."""
        params = [self.get_parameter(n) for n in names]
        t = self._packed.get(key)
        if self._is_packed(t, params, stacked=False):
            return t
        self._drop_superseded(key, params)
        t = torch.cat([p.data for p in params], dim=0).contiguous()
        off = 0
        for p in params:
            n = p.data.shape[0]
            p.data = t[off:off + n]
            off += n
        self._packed[key] = t
        self._storage_moved()
        return t

    def _pack_stack(self, key: str, names: Sequence[str]) -> torch.Tensor:
        params = [self.get_parameter(n) for n in names]
        t = self._packed.get(key)
        if self._is_packed(t, params, stacked=True):
            return t
        self._drop_superseded(key, params)
        t = torch.stack([p.data for p in params], dim=0).contiguous()
        for i, p in enumerate(params):
            p.data = t[i]
        self._packed[key] = t
        self._storage_moved()
        return t

    def _attn_qkv(self, p: str) -> Tuple[torch.Tensor, torch.Tensor]:
        return (self._pack(p + ".qkv.w", [f"{p}.to_q.weight", f"{p}.to_k.weight", f"{p}.to_v.weight"]),
                self._pack(p + ".qkv.b", [f"{p}.to_q.bias", f"{p}.to_k.bias", f"{p}.to_v.bias"]))

    def _single_qkv_mlp(self, p: str) -> Optional[Tuple[torch.Tensor, torch.Tensor]]:
        """ONE grouping of a single block's projections for the inference engine and the training forward alike: the four-way pack
        [to_q | to_k | to_v | proj_mlp] when the q|k|v / mlp split falls on a 256-column tile boundary (3 D % 256 == 0: the fused launch of
        flux._single_block), otherwise None and the caller uses the three-way _attn_qkv pack. (Two groupings of the same weights made
        alternating train / eval forwards re-pack on every call.)"""
        if (3 * self.inner_dim) % 256 != 0:
            return None
        a = p + ".attn"
        names = [f"{a}.to_q", f"{a}.to_k", f"{a}.to_v", f"{p}.proj_mlp"]
        return (self._pack(p + ".qkv_mlp.w", [x + ".weight" for x in names]), self._pack(p + ".qkv_mlp.b", [x + ".bias" for x in names]))

    def _attn_add_qkv(self, p: str) -> Tuple[torch.Tensor, torch.Tensor]:
        return (self._pack(p + ".aqkv.w", [f"{p}.add_q_proj.weight", f"{p}.add_k_proj.weight", f"{p}.add_v_proj.weight"]),
                self._pack(p + ".aqkv.b", [f"{p}.add_q_proj.bias", f"{p}.add_k_proj.bias", f"{p}.add_v_proj.bias"]))

    # ------------------------------------------------------------------ small pieces ----------------------------------
    def _w(self, name, shape, dtype=None):
        return self._ws.get(name, shape, dtype if dtype is not None else self.dtype, self.device)

    # ------------------------------------------------------------------ CoMoE routing ---------------------------------
    def _route(self, x: torch.Tensor, c: torch.Tensor, E: int, top_k: int, draw: Optional[torch.Tensor]) -> "_Routing":
        """TopKGate.forward on (x + c) [S, D] + deepspeed's top1gating (k = 1: capacity ceil(S / E), Random Token Selection; `draw` = its
        Uniform(0, 1) sample [S, E]) or top2gating (k = 2: capacity ceil(2 S / E), second choice by Gumbel-max; `draw` = its Gumbel(0, 1) sample)
        in index form (src/UniGenUtils.py:99; include/unigen_hip.h), or topkgating (k > 2: no draw). A missing draw is taken from the device RNG, as the reference does."""
        S, dev = x.shape[0], x.device
        wg = self._P("moe.moe_layer.gate.wg.weight")
        gates = self._w("moe_gates", (S, E), torch.float32)
        exp_counts = torch.empty(E, device=dev, dtype=torch.int64)
        l_aux = torch.empty(1, device=dev, dtype=torch.float32)
        if top_k == 1:
            C = max(int(math.ceil(S / E)), 4)          # deepspeed _capacity(capacity_factor=1, min_capacity=4)
            idx, slot, tos = self._w("moe_idx", (S,), torch.int32), self._w("moe_slot", (S,), torch.int32), self._w("moe_tos", (E, C), torch.int32)
            ops.moe_gate_top1(x, c, wg, gates, idx)
            if draw is None:
                draw = torch.rand(S, E, device=dev, dtype=torch.float32)   # RTS draw; the reference consumes the global device RNG too
            ops.moe_capacity_rts(gates, idx, draw.contiguous(), C, slot, tos, exp_counts, l_aux)
            return _Routing(1, E, C, gates, idx, slot, tos, None, exp_counts, l_aux)
        if top_k > 2:
            # deepspeed topkgating (k > 2; used by no shipped configuration of the reference, restated from its published source, include/unigen_hip.h):
            # capacity ceil(k S / E), the K largest logits per token, per expert the `capacity` largest of [chosen logit | 0], renormalised weights
            if top_k > E:
                raise ValueError(f"top_num {top_k} exceeds the {E} experts")
            C = max(int(math.ceil((S / E) * float(top_k))), 4)
            K = top_k
            idx, slot, tos = self._w("moe_idxk", (K, S), torch.int32), self._w("moe_slotk", (K, S), torch.int32), self._w("moe_tos", (E, C), torch.int32)
            weights, logits = self._w("moe_wk", (K, S), torch.float32), self._w("moe_logits", (S, E), torch.float32)
            ops.moe_gate_topk(x, c, wg, K, gates, logits, idx)
            ops.moe_capacity_topk(gates, logits, idx, C, slot, tos, weights, exp_counts, l_aux)
            return _Routing(K, E, C, gates, idx, slot, tos, weights, exp_counts, l_aux)
        C = max(int(math.ceil((S / E) * 2.0)), 4)      # top2gating: _capacity(gates, capacity_factor * 2, min_capacity)
        idx, slot, tos = self._w("moe_idx2", (2, S), torch.int32), self._w("moe_slot2", (2, S), torch.int32), self._w("moe_tos", (E, C), torch.int32)
        weights = self._w("moe_w2", (2, S), torch.float32)
        if draw is None:                               # gumbel_rsample: torch.distributions.gumbel.Gumbel(0, 1).rsample on the device
            draw = torch.distributions.gumbel.Gumbel(torch.tensor(0.0, device=dev), torch.tensor(1.0, device=dev)).rsample((S, E))
        ops.moe_gate_top2(x, c, wg, draw.to(torch.float32).contiguous(), gates, idx)
        ops.moe_capacity_top2(gates, idx, C, slot, tos, weights, exp_counts, l_aux)
        return _Routing(2, E, C, gates, idx, slot, tos, weights, exp_counts, l_aux)

    def _combine(self, rt: "_Routing", yh: torch.Tensor, yc: torch.Tensor, out: torch.Tensor, rows: Optional[slice] = None, **kw) -> None:
        """einsum("sec,ecm->sm") with the routing's combine weights, fused with the CoMoE residual sums (ops.moe_combine / moe_combine_topk);
        rows: a slice of the token axis (one sample)."""
        sl = rows if rows is not None else slice(None)
        if rt.K == 1:
            ops.moe_combine(yh, yc, rt.gates[sl], rt.idx[sl], rt.slot[sl], out, E=rt.E, capacity=rt.C, **kw)
        else:
            ops.moe_combine_topk(yh, yc, rt.weights[:, sl], rt.idx[:, sl], rt.slot[:, sl], out, E=rt.E, capacity=rt.C, **kw)

    def _time_text_embed(self, prefix: str, t_f32: torch.Tensor, pooled: torch.Tensor, g_f32: Optional[torch.Tensor], tag: str) -> torch.Tensor:
        """CombinedTimestep(Guidance)TextProjEmbeddings (diffusers embeddings.py)."""
        B, D = pooled.shape[0], self.inner_dim
        tp = ops.timestep_embed(t_f32, self._w("tp", (B, 256)))
        h1 = ops.small_linear(tp, self._P(prefix + ".timestep_embedder.linear_1.weight"), self._P(prefix + ".timestep_embedder.linear_1.bias"), self._w("tte_h", (B, D)))
        emb = ops.small_linear(h1, self._P(prefix + ".timestep_embedder.linear_2.weight"), self._P(prefix + ".timestep_embedder.linear_2.bias"),
                               self._w("tte_t" + tag, (B, D)), silu_in=True)
        if g_f32 is not None:
            gp = ops.timestep_embed(g_f32, self._w("gp", (B, 256)))
            h1 = ops.small_linear(gp, self._P(prefix + ".guidance_embedder.linear_1.weight"), self._P(prefix + ".guidance_embedder.linear_1.bias"), self._w("tte_h", (B, D)))
            emb = ops.small_linear(h1, self._P(prefix + ".guidance_embedder.linear_2.weight"), self._P(prefix + ".guidance_embedder.linear_2.bias"),
                                   self._w("tte_g" + tag, (B, D)), silu_in=True, residual=emb)
        h2 = ops.small_linear(pooled, self._P(prefix + ".text_embedder.linear_1.weight"), self._P(prefix + ".text_embedder.linear_1.bias"), self._w("tte_h2", (B, D)))
        return ops.small_linear(h2, self._P(prefix + ".text_embedder.linear_2.weight"), self._P(prefix + ".text_embedder.linear_2.bias"),
                                self._w("temb_" + tag, (B, D)), silu_in=True, residual=emb)

    def _adaln_emb(self, prefix: str, temb: torch.Tensor, nchunks: int, tag: str) -> torch.Tensor:
        """AdaLayerNormZero*.linear(silu(emb)) -> [B, nchunks * D]: a column slice of this step's grouped table when the module was
        part of an _adaln_group call, otherwise its own launch."""
        hit = self._emb_tab.get(prefix)
        if hit is not None:
            assert hit.shape[1] == nchunks * self.inner_dim, (prefix, hit.shape, nchunks)
            return hit
        B, D = temb.shape[0], self.inner_dim
        return ops.small_linear(temb, self._P(prefix + ".linear.weight"), self._P(prefix + ".linear.bias"), self._w("emb_" + tag, (B, nchunks * D)), silu_in=True)

    def _adaln_group(self, key: str, temb: torch.Tensor, prefixes: Sequence[str]) -> None:
        """Every AdaLN linear that reads the same vector (`temb`, `condition_temb` or `control_temb`: only three distinct inputs per
        step, src/UniGenTransformer.py:1048-1049,1222) as ONE launch over their stacked weights [sum N_i, D] (the parameters become views
        of the stack): ~170 per-block launches per forward become 3-4, and the 6.5 GB of AdaLN weights stream in one go. The per-module
        results are column slices of the [B, sum N_i] table (leading dimension sum N_i), looked up by _adaln_emb."""
        prefixes = list(prefixes)
        w = self._pack("adaln." + key + ".w", [p + ".linear.weight" for p in prefixes])
        b = self._pack("adaln." + key + ".b", [p + ".linear.bias" for p in prefixes])
        tab = ops.small_linear(temb, w, b, self._w("adaln_tab_" + key, (temb.shape[0], w.shape[0])), silu_in=True)
        off = 0
        for p in prefixes:
            n = self.get_parameter(p + ".linear.weight").shape[0]
            self._emb_tab[p] = tab[:, off:off + n]
            off += n

    def _modulate(self, s: _Stream, emb: torch.Tensor, shift_chunk: int, scale_chunk: int, B: int, tag: str) -> torch.Tensor:
        D = self.inner_dim
        out = self._w("norm_" + tag, (B * s.Ls, D))
        ops.adaln_modulate(s.base, emb[:, shift_chunk * D:], emb[:, scale_chunk * D:], out, rows=B * s.Ls, D=D, rows_per_sample=s.Ls,
                           mod_ld=emb.stride(0), ldx=s.ld, x_map=s.map)
        return out

    # ------------------------------------------------------------------ blocks ----------------------------------------
    def _double_block(self, p: str, B: int, s_in: _Stream, s_out: _Stream, c_in: _Stream, c_out: Optional[_Stream], temb: torch.Tensor,
                      rope: Optional[Tuple[torch.Tensor, torch.Tensor]], tag: str, ctx_cached: bool = False, dual: bool = False,
                      ctx_continuous: bool = False) -> None:
        """One MM-DiT joint block: diffusers FluxTransformerBlock (and its undefined control twin) or SD3's
        JointTransformerBlock as restated at src/UniGenUtils.py:440-522. Joint layout [context | sample] (attention is
        order-free; RoPE tables, when given, follow this order).
        c_out None : the context stream only contributes K/V (its own output is discarded by every caller, or the block is
                     `context_pre_only`); ctx_continuous selects AdaLayerNormContinuous (scale first) for that stream.
        ctx_cached : the context K/V already sit in this tag's qkv workspace (same block, same step).
        dual       : SD3.5 dual attention (SD35AdaLayerNormZeroX: 9 chunks; attn2 = self-attention over the sample stream)."""
        D, H, dh = self.inner_dim, self._heads, self._head_dim
        Ls, Lc = s_in.Ls, c_in.Ls
        Lj = Lc + Ls
        a = p + ".attn"
        qkv = self._w("qkv_" + tag, (B, Lj, 3 * D))
        qkv2 = qkv.view(B * Lj, 3 * D)
        cos, sin = rope if rope is not None else (None, None)
        opt = lambda n: self._P(n) if n in self._pnames() else None
        wq, wk = opt(a + ".norm_q.weight"), opt(a + ".norm_k.weight")
        waq, wak = opt(a + ".norm_added_q.weight"), opt(a + ".norm_added_k.weight")
        touch = cos is not None or wq is not None           # q/k need a pass at all (qk-norm and/or RoPE)
        # sample stream: AdaLN-Zero -> fused QKV
        emb_s = self._adaln_emb(p + ".norm1", temb, 9 if dual else 6, tag + "s")
        ns = self._modulate(s_in, emb_s, 0, 1, B, "s")
        # SD35AdaLayerNormZeroX (src/UniGenUtils.py:340-352): norm_hidden_states2 is derived from the block's INPUT. It is taken here,
        # before the to_out GEMM below overwrites the sample stream (s_in and s_out are the same buffer for in-place blocks).
        n2 = self._modulate(s_in, emb_s, 6, 7, B, "s2") if dual else None
        w_qkv, b_qkv = self._attn_qkv(a)
        # sample rows: q / k RMSNorm + RoPE in the projection's epilogue when the shapes allow it (whole 256^2 tiles, heads of 128)
        cs = getattr(rope, "cs", None)
        # live LoRA adapters of to_q / to_k / to_v: T = ns A_cat^T in one GEMM, then the K-segment of this launch (the q/k RMSNorm + RoPE then
        # runs as its own kernel: UG_EPI_QKV_ROPE is not LoRA-extended)
        lt, lb = self._lora_operands([a + ".to_q", a + ".to_k", a + ".to_v"], ns, B * Ls, "s")
        fused = (lt is None and wq is not None and (cs is not None or (cos is None and dh == 64)) and ops.qk_rope_fusable(B * Ls, 3 * D, 2 * D, dh, ns.dtype))
        ops.gemm(ns, w_qkv, b_qkv, qkv2[Lc:], M=B * Ls, ldc=3 * D, c_map=RowMap(Ls, Lj),
                 qk_rope=ops.QkRope(wq, wk, cs, Ls, Lc, 2 * D, dh=dh) if fused else None, lora_t=lt, lora_b=lb)
        # context stream
        emb_c = None
        if c_out is not None or not ctx_cached:
            if ctx_continuous:                                # AdaLayerNormContinuous: chunks (scale, shift)
                emb_c = self._adaln_emb(p + ".norm1_context", temb, 2, tag + "c")
                nc = self._modulate(c_in, emb_c, 1, 0, B, "c")
            else:
                emb_c = self._adaln_emb(p + ".norm1_context", temb, 6, tag + "c")
                nc = self._modulate(c_in, emb_c, 0, 1, B, "c")
            w_a, b_a = self._attn_add_qkv(a)
            if c_out is not None:
                lt, lb = self._lora_operands([a + ".add_q_proj", a + ".add_k_proj", a + ".add_v_proj"], nc, B * Lc, "c")
                ops.gemm(nc, w_a, b_a, qkv2, M=B * Lc, ldc=3 * D, c_map=RowMap(Lc, Lj), lora_t=lt, lora_b=lb)
            else:
                lt, lb = self._lora_operands([a + ".add_k_proj", a + ".add_v_proj"], nc, B * Lc, "c")
                ops.gemm(nc, w_a[D:], b_a[D:], qkv2[0, D:], M=B * Lc, ldc=3 * D, c_map=RowMap(Lc, Lj), lora_t=lt, lora_b=lb)
        if touch:
            if c_out is not None and not fused:
                ops.qk_rmsnorm_rope(qkv2, batches=B, rows_per_batch=Lj, ld=3 * D, q_off=0, k_off=D, heads=H, dh=dh, wq_a=waq, wk_a=wak,
                                    wq_b=wq, wk_b=wk, split=Lc, cos=cos, sin=sin)
            elif c_out is not None:                           # the context rows only
                ops.qk_rmsnorm_rope(qkv2, batches=B, rows_per_batch=Lc, batch_stride_rows=Lj, pos_offset=0, ld=3 * D, q_off=0, k_off=D,
                                    heads=H, dh=dh, wq_a=waq, wk_a=wak, split=Lc, cos=cos, sin=sin)
            else:
                if not fused:
                    ops.qk_rmsnorm_rope(qkv2[Lc:], batches=B, rows_per_batch=Ls, batch_stride_rows=Lj, pos_offset=Lc, ld=3 * D, q_off=0,
                                        k_off=D, heads=H, dh=dh, wq_b=wq, wk_b=wk, split=0, cos=cos, sin=sin)
                if not ctx_cached:
                    ops.qk_rmsnorm_rope(qkv2, batches=B, rows_per_batch=Lc, batch_stride_rows=Lj, pos_offset=0, ld=3 * D, q_off=-1, k_off=D,
                                        heads=H, dh=dh, wk_a=wak, split=Lc, cos=cos, sin=sin)
        # attention over the joint sequence; queries = every row, or the sample rows only
        st = (3 * D, Lj * 3 * D)
        if c_out is not None:
            att = self._w("att_" + tag, (B * Lj, D))
            ops.flash_attn(qkv2, qkv2[0, D:], qkv2[0, 2 * D:], att, batches=B, heads=H, dh=dh, Lq=Lj, Lkv=Lj, q_strides=st, k_strides=st,
                           v_strides=st, o_strides=(D, Lj * D))
            att_s, att_map = att[Lc:], RowMap(Ls, Lj)
        else:
            att = self._w("att_" + tag, (B * Ls, D))
            ops.flash_attn(qkv2[Lc:], qkv2[0, D:], qkv2[0, 2 * D:], att, batches=B, heads=H, dh=dh, Lq=Ls, Lkv=Lj, q_strides=st, k_strides=st,
                           v_strides=st, o_strides=(D, Ls * D))
            att_s, att_map = att, RowMap()
        # sample: x = x + gate_msa * to_out(attn)
        lt, lb = self._lora_operands([a + ".to_out.0"], att_s, B * Ls, "o", lda=D, a_map=att_map)
        ops.gemm(att_s, self._P(a + ".to_out.0.weight"), self._P(a + ".to_out.0.bias"), s_out.base, M=B * Ls, epilogue=L.EPI_RES_GATE, lda=D,
                 a_map=att_map, ldc=s_out.ld, c_map=s_out.map, residual=s_in.base, ldr=s_in.ld, r_map=s_in.map, gate=emb_s[:, 2 * D:],
                 gate_ld=emb_s.stride(0), rows_per_sample=Ls, lora_t=lt, lora_b=lb)
        if dual:
            # x = x + gate_msa2 * attn2(LN(x_in) * (1 + scale_msa2) + shift_msa2): self-attention over the sample tokens only
            a2 = p + ".attn2"
            q2 = self._w("qkv2_" + tag, (B * Ls, 3 * D))
            w2, b2 = self._attn_qkv(a2)
            w2q, w2k = opt(a2 + ".norm_q.weight"), opt(a2 + ".norm_k.weight")
            lt, lb = self._lora_operands([a2 + ".to_q", a2 + ".to_k", a2 + ".to_v"], n2, B * Ls, "s")
            fused2 = lt is None and w2q is not None and dh == 64 and ops.qk_rope_fusable(B * Ls, 3 * D, 2 * D, dh, n2.dtype)     # attn2 has no RoPE
            ops.gemm(n2, w2, b2, q2, M=B * Ls, qk_rope=ops.QkRope(w2q, w2k, None, Ls, 0, 2 * D, dh=dh) if fused2 else None, lora_t=lt, lora_b=lb)
            if w2q is not None and not fused2:
                ops.qk_rmsnorm_rope(q2, batches=B, rows_per_batch=Ls, ld=3 * D, q_off=0, k_off=D, heads=H, dh=dh, wq_b=w2q, wk_b=w2k, split=0)
            att2 = self._w("att2_" + tag, (B * Ls, D))
            st2 = (3 * D, Ls * 3 * D)
            ops.flash_attn(q2, q2[0, D:], q2[0, 2 * D:], att2, batches=B, heads=H, dh=dh, Lq=Ls, Lkv=Ls, q_strides=st2, k_strides=st2,
                           v_strides=st2, o_strides=(D, Ls * D))
            lt, lb = self._lora_operands([a2 + ".to_out.0"], att2, B * Ls, "o")
            ops.gemm(att2, self._P(a2 + ".to_out.0.weight"), self._P(a2 + ".to_out.0.bias"), s_out.base, M=B * Ls, epilogue=L.EPI_RES_GATE,
                     ldc=s_out.ld, c_map=s_out.map, residual=s_out.base, ldr=s_out.ld, r_map=s_out.map, gate=emb_s[:, 8 * D:],
                     gate_ld=emb_s.stride(0), rows_per_sample=Ls, lora_t=lt, lora_b=lb)
        # x = x + gate_mlp * ff(norm2(x) * (1 + scale_mlp) + shift_mlp)
        self._ff(p + ".ff", B, s_out, emb_s, "s")
        if c_out is not None:
            lt, lb = self._lora_operands([a + ".to_add_out"], att, B * Lc, "o", lda=D, a_map=RowMap(Lc, Lj))
            ops.gemm(att, self._P(a + ".to_add_out.weight"), self._P(a + ".to_add_out.bias"), c_out.base, M=B * Lc, epilogue=L.EPI_RES_GATE, lda=D,
                     a_map=RowMap(Lc, Lj), ldc=c_out.ld, c_map=c_out.map, residual=c_in.base, ldr=c_in.ld, r_map=c_in.map, gate=emb_c[:, 2 * D:],
                     gate_ld=emb_c.stride(0), rows_per_sample=Lc, lora_t=lt, lora_b=lb)
            self._ff(p + ".ff_context", B, c_out, emb_c, "c")

    def _pnames(self):
        names = getattr(self, "_pname_cache", None)
        if names is None:        # subclasses reset the cache when they register parameters (init_condition_block)
            names = {n for n, _ in self.named_parameters()}
            self._pname_cache = names
        return names

    def _ff(self, p: str, B: int, s: _Stream, emb: torch.Tensor, tag: str) -> None:
        D = self.inner_dim
        n2 = self._modulate(s, emb, 3, 4, B, tag)
        hid = self._w("ffh_" + tag, (B * s.Ls, 4 * D))
        lt, lb = self._lora_operands([p + ".net.0.proj"], n2, B * s.Ls, "f0")
        ops.gemm(n2, self._P(p + ".net.0.proj.weight"), self._P(p + ".net.0.proj.bias"), hid, M=B * s.Ls, epilogue=L.EPI_BIAS_GELU, lora_t=lt, lora_b=lb)
        lt, lb = self._lora_operands([p + ".net.2"], hid, B * s.Ls, "f2")
        ops.gemm(hid, self._P(p + ".net.2.weight"), self._P(p + ".net.2.bias"), s.base, M=B * s.Ls, epilogue=L.EPI_RES_GATE, ldc=s.ld, c_map=s.map,
                 residual=s.base, ldr=s.ld, r_map=s.map, gate=emb[:, 5 * D:], gate_ld=emb.stride(0), rows_per_sample=s.Ls, lora_t=lt, lora_b=lb)

