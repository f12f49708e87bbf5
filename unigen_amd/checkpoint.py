"""Wire formats of UniGen control checkpoints -> one state dict -> `model.load_state_dict(..., strict=False)`.

The reference resolves `--transformer` three ways (infer.py:124-140) and `train.py` saves a fourth (src/hook.py:10-27):

  1. a DeepSpeed ZeRO checkpoint directory (has a `latest` file): the reference calls
     `deepspeed.utils.zero_to_fp32.get_fp32_state_dict_from_zero_checkpoint` (infer.py:124-128). A consolidated `pytorch_model_fp32.bin`
     (script/infer.sh:44-46) is read when it sits in the directory; otherwise the per-rank shards of `<dir>/<tag>/` are merged here
     (`merge_zero_checkpoint`: the algorithm of deepspeed 0.16.5 zero_to_fp32.py restated, stages 1/2 and 3, frozen parameters, shared
     parameters, buffers) - deepspeed itself is neither needed nor imported;
  2. a single `torch.save`d state dict (`*.bin` / `*.pt`);
  3. a directory of `*.safetensors` shards;
  4. `{module}_weights_{idx}.bin` files written by `save_all_model_hook`: one partial state dict per trainable module family.

Tensors are returned on the CPU in their stored dtype; the model's `load_state_dict` casts to bf16 and writes through its packed
(fused QKV / stacked expert) views. No network access. Plain state-dict files load with `weights_only=True`; ZeRO shard files pickle
DeepSpeed objects (loss scaler, config) beside the tensors, so they are read with an unpickler that resolves an exact allowlist of
(module, name) pairs - tensor rebuild functions, storages, dtypes, plain containers - and replaces every other global by an inert stub:
nothing from the checkpoint is imported or executed (tests/test_host_cpu.py::test_zero_unpickler_runs_no_payload).

Parity note: no DeepSpeed checkpoint exists in this container (no deepspeed, no weights), so the merge is pinned only by a writer that
restates the SAVE side of the same release (tests/test_host_cpu.py::test_zero_shard_merge) - format parity unpinned.
"""
from __future__ import annotations

import glob
import os
from typing import Dict

import torch


def _torch_load(path: str) -> Dict[str, torch.Tensor]:
    sd = torch.load(path, map_location="cpu", weights_only=True)
    if isinstance(sd, dict) and "state_dict" in sd and isinstance(sd["state_dict"], dict):
        sd = sd["state_dict"]
    if not isinstance(sd, dict) or not all(isinstance(v, torch.Tensor) for v in sd.values()):
        raise ValueError(f"{path}: not a flat name -> tensor state dict")
    return sd


def read_control_state_dict(path: str) -> Dict[str, torch.Tensor]:
    """The reference's `--transformer` resolution order (infer.py:124-140), plus the hook format."""
    if os.path.isdir(path):
        if os.path.exists(os.path.join(path, "latest")):                                   # 1. ZeRO checkpoint directory
            for name in ("pytorch_model_fp32.bin", "pytorch_model.bin"):
                f = os.path.join(path, name)
                if os.path.exists(f):
                    return _torch_load(f)
            return merge_zero_checkpoint(path)
        sd: Dict[str, torch.Tensor] = {}
        st = sorted(glob.glob(os.path.join(path, "*.safetensors")))
        if st:                                                                              # 3. safetensors shards
            from safetensors.torch import load_file
            for f in st:
                sd.update(load_file(f))
            return sd
        hooks = sorted(glob.glob(os.path.join(path, "*_weights_*.bin")))
        if hooks:                                                                           # 4. save_all_model_hook files
            for f in hooks:
                part = _torch_load(f)
                dup = set(part) & set(sd)
                if dup:
                    raise ValueError(f"{f}: {len(dup)} keys already defined by an earlier file (e.g. {sorted(dup)[0]})")
                sd.update(part)
            return sd
        raise OSError(f"no control checkpoint under {path} (looked for latest, *.safetensors, *_weights_*.bin)")
    if os.path.exists(path):                                                                # 2. single state-dict file
        if path.endswith(".safetensors"):
            from safetensors.torch import load_file
            return load_file(path)
        return _torch_load(path)
    raise OSError(f"{path}: no such checkpoint")


# ----------------------------------------------------------------------------------------------------------------------
# DeepSpeed ZeRO shard merge (deepspeed 0.16.5 utils/zero_to_fp32.py, restated)
# ----------------------------------------------------------------------------------------------------------------------

class _Stub:
    """Stands in for any class a ZeRO shard pickles that is not a tensor / container (LossScaler, DeepSpeedConfig, ...)."""

    def __init__(self, *a, **k):
        pass

    def __setstate__(self, state):
        self.__dict__["_state"] = state

    def __call__(self, *a, **k):
        return self


# Exact (module, name) pairs a ZeRO shard may resolve. Everything else - the rest of `builtins` (eval, exec, getattr, __import__ ...), of
# `torch` (torch.hub, torch.load ...) and of `numpy` included - becomes an inert _Stub: a `__reduce__` that names it calls _Stub(...), which
# does nothing. (allowlisting whole module roots let `builtins.eval` through.)
_SAFE_BUILTINS = ("dict", "list", "tuple", "set", "frozenset", "int", "float", "bool", "bytes", "bytearray", "str", "slice", "complex", "range")
_SAFE_TORCH_STORAGES = ("FloatStorage", "DoubleStorage", "HalfStorage", "BFloat16Storage", "LongStorage", "IntStorage", "ShortStorage",
                        "CharStorage", "ByteStorage", "BoolStorage", "UntypedStorage", "ComplexFloatStorage", "ComplexDoubleStorage")
_SAFE_TORCH_DTYPES = ("float32", "float64", "float16", "bfloat16", "int64", "int32", "int16", "int8", "uint8", "bool", "complex64", "complex128",
                      "float", "double", "half", "long", "int", "short")


def _safe_globals():
    import collections
    ok = {("collections", "OrderedDict"): collections.OrderedDict, ("torch", "Size"): torch.Size, ("torch", "device"): torch.device,
          ("torch._utils", "_rebuild_tensor_v2"): torch._utils._rebuild_tensor_v2, ("torch._utils", "_rebuild_parameter"): torch._utils._rebuild_parameter,
          ("torch._utils", "_rebuild_tensor"): torch._utils._rebuild_tensor, ("torch.serialization", "_get_layout"): torch.serialization._get_layout,
          ("_codecs", "encode"): __import__("_codecs").encode}           # bytes objects of protocol-2 pickles
    import builtins
    for n in _SAFE_BUILTINS:
        ok[("builtins", n)] = ok[("__builtin__", n)] = getattr(builtins, n)
    for n in _SAFE_TORCH_STORAGES:
        if hasattr(torch, n):
            ok[("torch", n)] = getattr(torch, n)
    ok[("torch.storage", "UntypedStorage")] = torch.UntypedStorage
    ok[("torch.storage", "_load_from_bytes")] = _Stub                    # carries a nested pickle: never evaluated
    for n in _SAFE_TORCH_DTYPES:
        ok[("torch", n)] = getattr(torch, n)
    return ok


class _SafePickle:
    """pickle_module for torch.load of a ZeRO shard: only the exact globals of `_safe_globals()` resolve, everything else becomes _Stub."""
    import pickle as _p
    __name__ = "unigen_amd.checkpoint._SafePickle"
    load, loads, dump, dumps = _p.load, _p.loads, _p.dump, _p.dumps
    HIGHEST_PROTOCOL, DEFAULT_PROTOCOL, PickleError, UnpicklingError = _p.HIGHEST_PROTOCOL, _p.DEFAULT_PROTOCOL, _p.PickleError, _p.UnpicklingError
    Pickler = _p.Pickler

    class Unpickler(_p.Unpickler):
        _OK = None

        def find_class(self, module, name):
            cls = type(self)
            if cls._OK is None:
                cls._OK = _safe_globals()
            hit = cls._OK.get((module, name))
            if hit is not None:
                return hit
            return type(str(name).split(".")[-1] or "Stub", (_Stub,), {"__module__": module})


def _zero_load(path: str):
    return torch.load(path, map_location="cpu", weights_only=False, pickle_module=_SafePickle)


def _has_stub(v, depth: int = 0):
    """The first blocked global inside a value the merge needs (a _Stub instance or class), or None."""
    if isinstance(v, _Stub) or (isinstance(v, type) and issubclass(v, _Stub)):
        return v if isinstance(v, type) else type(v)
    if depth < 3:
        if isinstance(v, dict):
            for x in list(v.keys())[:64] + list(v.values())[:64]:
                hit = _has_stub(x, depth + 1)
                if hit is not None:
                    return hit
        elif isinstance(v, (list, tuple)):
            for x in v[:64]:
                hit = _has_stub(x, depth + 1)
                if hit is not None:
                    return hit
    return None


def _need(d, key, where, leaf: bool = True):
    """A field the merge arithmetic depends on: present, and not replaced by the unpickler's inert stub (a shard that stores e.g. its
    partition_count as a numpy scalar or its tensors as a subclass resolves globals outside the allowlist; failing HERE names the blocked
    (module, name) instead of a later `int(_Stub)` TypeError)."""
    if not isinstance(d, dict) or key not in d:
        raise ValueError(f"{where}: not a DeepSpeed ZeRO shard this reader understands (no '{key}' entry; written by deepspeed 0.14-0.16?)")
    hit = _has_stub(d[key]) if leaf else None          # containers such as optimizer_state_dict legitimately hold stubbed objects (loss scaler, config)
    if hit is not None:
        raise ValueError(f"{where}: field '{key}' needs the global {hit.__module__}.{hit.__name__}, which the shard reader does not resolve "
                         "(only tensors, storages, dtypes and plain containers are allow-listed); convert the checkpoint with zero_to_fp32.py instead")
    return d[key]


def _natural(files):
    import re
    return sorted(files, key=lambda f: [int(t) if t.isdigit() else t for t in re.split(r"(\d+)", os.path.basename(f))])


def merge_zero_checkpoint(checkpoint_dir: str, tag: str = None) -> Dict[str, torch.Tensor]:
    """get_fp32_state_dict_from_zero_checkpoint(checkpoint_dir, tag): fp32 master weights of every trainable parameter re-assembled from
    the per-rank optimizer shards, plus frozen parameters, buffers and shared-parameter aliases."""
    if tag is None:
        with open(os.path.join(checkpoint_dir, "latest")) as f:
            tag = f.read().strip()
    ds_dir = os.path.join(checkpoint_dir, tag)
    if not os.path.isdir(ds_dir):
        raise OSError(f"{ds_dir}: ZeRO checkpoint tag directory not found")
    optim_files = _natural(glob.glob(os.path.join(ds_dir, "*_optim_states.pt")))
    if not optim_files:
        raise OSError(f"{ds_dir}: no *_optim_states.pt shards")
    optim = [_need(_zero_load(f), "optimizer_state_dict", f, leaf=False) for f in optim_files]
    stage = int(_need(optim[0], "zero_stage", optim_files[0]))
    world = _need(optim[0], "partition_count", optim_files[0])
    world = int(max(world)) if isinstance(world, (list, tuple)) else int(world)
    if world != len(optim_files):
        raise ValueError(f"{ds_dir}: {len(optim_files)} optimizer shards but partition_count = {world}")
    model_files = _natural(glob.glob(os.path.join(ds_dir, "zero_pp_rank_*_mp_rank_00_model_states.pt"))) if stage == 3 else \
        _natural(glob.glob(os.path.join(ds_dir, "mp_rank_00_model_states.pt")))
    if not model_files:
        raise OSError(f"{ds_dir}: no *_model_states.pt for ZeRO stage {stage}")
    models = [_zero_load(f) for f in model_files]
    m0 = models[0]
    param_shapes = _need(m0, "param_shapes", model_files[0])   # list (one dict name -> shape per optimizer param group)
    if isinstance(param_shapes, dict):
        param_shapes = [param_shapes]
    numel = lambda shp: int(torch.Size(shp).numel())
    sd: Dict[str, torch.Tensor] = {}
    # buffers (fp32 copies)
    buffer_names = set(m0.get("buffer_names", []))
    for k, v in (m0.get("module") or {}).items():
        if k in buffer_names:
            sd[k] = v.float()
    # frozen parameters
    frozen_shapes = (_need(m0, "frozen_param_shapes", model_files[0]) if m0.get("frozen_param_shapes") is not None else None) or {}
    if frozen_shapes:
        for m, f in zip(models, model_files):
            _need(m, "frozen_param_fragments", f)
    for name, shp in frozen_shapes.items():
        if stage == 3:
            frag = torch.cat([m["frozen_param_fragments"][name].reshape(-1) for m in models], 0)
            sd[name] = frag.narrow(0, 0, numel(shp)).view(torch.Size(shp)).float()
        else:
            sd[name] = m0["frozen_param_fragments"][name].float()
    if stage == 3:
        flats = [torch.cat([g.reshape(-1) for g in _need(o, "fp32_flat_groups", f)], 0) for o, f in zip(optim, optim_files)]      # per rank: its partitions of every param, in order
        offset = 0
        for shapes in param_shapes:
            for name, shp in shapes.items():
                n = numel(shp)
                part = -(-n // world)                                                           # ceil: each rank holds `part` elements (zero padded)
                sd[name] = torch.cat([fl.narrow(0, offset, part) for fl in flats], 0).narrow(0, 0, n).view(torch.Size(shp))
                offset += part
        if offset > flats[0].numel():
            raise ValueError(f"{ds_dir}: shards hold {flats[0].numel()} elements per rank, the parameter list needs {offset}")
    elif stage in (1, 2):
        align = 2 * world
        up = lambda x: align * -(-x // align)
        ngroups = len(_need(optim[0], "single_partition_of_fp32_groups", optim_files[0]))
        if ngroups != len(param_shapes):
            raise ValueError(f"{ds_dir}: {ngroups} flat groups but {len(param_shapes)} parameter groups")
        for gi, shapes in enumerate(param_shapes):
            full = torch.cat([o["single_partition_of_fp32_groups"][gi].reshape(-1) for o in optim], 0)
            offset = 0
            for name, shp in shapes.items():
                n = numel(shp)
                sd[name] = full.narrow(0, offset, n).view(torch.Size(shp))
                offset += n
            if up(offset) != up(full.numel()):
                raise ValueError(f"{ds_dir}: group {gi} consumed {offset} of {full.numel()} elements")
    else:
        raise ValueError(f"{ds_dir}: unknown zero stage {stage}")
    for pair in m0.get("shared_params") or []:
        if pair[1] in sd:
            sd[pair[0]] = sd[pair[1]]
    return sd


def load_control_checkpoint(model, path: str):
    """`transformer.load_state_dict(read(path), strict=False)` as infer.py does; returns the (missing, unexpected) result."""
    return model.load_state_dict(read_control_state_dict(path), strict=False)
