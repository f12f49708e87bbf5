"""Drop-in import path of the reference's `src/lora_switching_module.py` (never invoked by the reference itself)."""
from unigen_amd.lora import LoRALinear, enable_lora, module_active_adapters  # noqa: F401

__all__ = ["enable_lora", "module_active_adapters", "LoRALinear"]
