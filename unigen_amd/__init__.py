"""unigen_amd: MI355X-native (gfx950) implementation of UniGen's condition-weaving + expert-modulation forward pass.

Layout: csrc/ (HIP kernels + C ABI -> libunigen_hip.so), lib.py (ctypes binding), ops.py (tensor front end),
flux.py (UniGenFlux / MultiCondtionUniGenFlux host engine), pipeline.py (denoise loop / UniGenFLUXPipeline surface)."""
__version__ = "0.1.0"
