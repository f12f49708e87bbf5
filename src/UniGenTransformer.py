"""Drop-in import path of the reference: `getattr(importlib.import_module("src.UniGenTransformer"), args.basemodel)`
(reference infer.py:115, train.py:302). The classes are the MI355X-native implementations in unigen_amd/."""
from unigen_amd.flux import MultiCondtionUniGenFlux, UniGenFlux  # noqa: F401
from unigen_amd.sd3 import UniGenSD3  # noqa: F401

__all__ = ["UniGenFlux", "MultiCondtionUniGenFlux", "UniGenSD3"]
