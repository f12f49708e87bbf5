"""Drop-in import path of the reference: `getattr(importlib.import_module("src.UniGenPipeline"), args.pipeline)` (reference infer.py:146)."""
from unigen_amd.pipeline import UniGenFLUXPipeline, UniGenSD3Pipeline  # noqa: F401

__all__ = ["UniGenFLUXPipeline", "UniGenSD3Pipeline"]
