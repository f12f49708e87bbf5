"""Wire formats of UniGen control checkpoints -> one state dict -> `model.load_state_dict(..., strict=False)`.

The reference resolves `--transformer` three ways (infer.py:124-140) and `train.py` saves a fourth (src/hook.py:10-27):

  1. a DeepSpeed ZeRO checkpoint directory (has a `latest` file): the reference calls
     `deepspeed.utils.zero_to_fp32.get_fp32_state_dict_from_zero_checkpoint`. Its documented offline product is one consolidated
     `pytorch_model_fp32.bin` (script/infer.sh:44-46); this loader reads that file when it sits in the directory and otherwise says how
     to make it - re-implementing the ZeRO shard merge is out of scope;
  2. a single `torch.save`d state dict (`*.bin` / `*.pt`);
  3. a directory of `*.safetensors` shards;
  4. `{module}_weights_{idx}.bin` files written by `save_all_model_hook`: one partial state dict per trainable module family.

Tensors are returned on the CPU in their stored dtype; the model's `load_state_dict` casts to bf16 and writes through its packed
(fused QKV / stacked expert) views. No network access, no pickle code execution (`weights_only=True`).
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
            raise OSError(f"{path} is a raw DeepSpeed ZeRO checkpoint; consolidate it first (python zero_to_fp32.py {path} "
                          f"{path}/pytorch_model_fp32.bin, as script/infer.sh of the reference does) and point here again")
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


def load_control_checkpoint(model, path: str):
    """`transformer.load_state_dict(read(path), strict=False)` as infer.py does; returns the (missing, unexpected) result."""
    return model.load_state_dict(read_control_state_dict(path), strict=False)
