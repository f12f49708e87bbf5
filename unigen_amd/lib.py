"""ctypes binding of libunigen_hip.so — the C ABI declared in include/unigen_hip.h.

The product path has no fallback: if the shared library is missing or does not load, importing the ops raises.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("UG_LIB_PATH") or os.path.join(_HERE, "libunigen_hip.so")      # UG_LIB_PATH: an alternative build (A/B tools only)

UG_OK, UG_ERR_BAD_SHAPE, UG_ERR_BAD_ALIGN, UG_ERR_UNSUPPORTED, UG_ERR_HIP = 0, -1, -2, -3, -4
EPI_BIAS, EPI_BIAS_GELU, EPI_RES_GATE, EPI_RES_SCALE, EPI_F32, EPI_QKV_ROPE = 0, 1, 2, 3, 4, 5

i64, i32, f32, vp = C.c_int64, C.c_int32, C.c_float, C.c_void_p


class GemmDesc(C.Structure):
    """Mirror of struct ug_gemm_desc (include/unigen_hip.h)."""

    _fields_ = [
        ("A", vp), ("lda", i64), ("a_rpb", i64), ("a_bstride", i64),
        ("W", vp), ("ldw", i64),
        ("bias", vp),
        ("C", vp), ("ldc", i64), ("c_rpb", i64), ("c_bstride", i64),
        ("R", vp), ("ldr", i64), ("r_rpb", i64), ("r_bstride", i64),
        ("gate", vp), ("gate_ld", i64), ("rows_per_sample", i64),
        ("alpha", f32),
        ("epilogue", i32),
        ("M", i64), ("N", i64), ("K", i64),
        ("groups", i32), ("_pad0", i32),
        ("a_gstride", i64), ("w_gstride", i64), ("bias_gstride", i64), ("c_gstride", i64),
        ("lora_T", vp), ("ldt", i64),
        ("lora_B", vp), ("ldb", i64),
        ("lora_r", i32), ("_pad1", i32),
        ("r_gstride", i64), ("gate_gstride", i64),
        ("workspace", vp), ("workspace_bytes", i64),
        ("gelu_from_n", i64), ("c_shift_from_n", i64), ("c_shift", i64),
        ("qk_wq", vp), ("qk_wk", vp), ("rope_cs", vp),
        ("rope_rpb", i64), ("rope_pos0", i64), ("qk_until_n", i64),
        ("qk_eps", f32), ("qk_dh", i32),
    ]


class ConvDesc(C.Structure):
    """Mirror of struct ug_conv_desc (include/unigen_hip.h)."""

    _fields_ = [("x", vp), ("B", i64), ("H", i64), ("W", i64), ("Cin", i64), ("w", vp), ("bias", vp), ("R", vp), ("out", vp), ("Ho", i64), ("Wo", i64),
                ("Cout", i64), ("KH", i32), ("KW", i32), ("stride", i32), ("pad_t", i32), ("pad_l", i32), ("up", i32), ("zero_page", vp), ("zero_page_bytes", i64)]


# name -> (restype, argtypes); must list every symbol declared in include/unigen_hip.h
SIGNATURES = {
    "ug_version": (i32, []),
    "ug_last_error": (C.c_char_p, []),
    "ug_gemm_workspace_bytes": (i64, []),
    "ug_gemm_bf16": (i32, [C.POINTER(GemmDesc), vp]),
    "ug_small_linear_bf16": (i32, [vp, i64, vp, i64, vp, vp, i64, vp, i64, i64, i64, i64, i32, vp]),
    "ug_adaln_modulate": (i32, [vp, i64, i64, i64, vp, vp, i64, i64, vp, i64, i64, i64, f32, vp]),
    "ug_qk_rmsnorm_rope": (i32, [vp, i64, i64, i64, i64, i64, i64, i64, i32, i32, vp, vp, vp, vp, i64, vp, vp, f32, vp]),
    "ug_flash_attn_fwd": (i32, [vp, i64, i64, vp, i64, i64, vp, i64, i64, vp, i64, i64, i64, i32, i64, i64, i32, f32, vp]),
    "ug_timestep_embed": (i32, [vp, vp, i64, i64, i32, vp]),
    "ug_euler_step": (i32, [vp, vp, f32, i64, vp]),
    "ug_cfg_combine": (i32, [vp, vp, f32, vp, i64, vp]),
    "ug_add_bf16": (i32, [vp, i64, vp, i64, vp, i64, i64, i64, vp]),
    "ug_add_rowbcast_f32": (i32, [vp, i64, vp, i64, i64, i64, i64, vp]),
    "ug_gather_rows": (i32, [vp, i64, vp, vp, i64, i64, i64, vp]),
    "ug_moe_gate_top1": (i32, [vp, vp, i64, vp, i64, i64, i32, vp, vp, vp]),
    "ug_moe_capacity_rts": (i32, [vp, vp, vp, i64, i32, i64, vp, vp, vp, vp, vp]),
    "ug_moe_gate_top2": (i32, [vp, vp, i64, vp, i64, i64, i32, vp, vp, vp, vp]),
    "ug_moe_capacity_top2": (i32, [vp, vp, i64, i32, i64, vp, vp, vp, vp, vp, vp]),
    "ug_moe_gate_topk": (i32, [vp, vp, i64, vp, i64, i64, i32, i32, vp, vp, vp, vp]),
    "ug_moe_capacity_topk": (i32, [vp, vp, vp, i64, i32, i32, i64, vp, vp, vp, vp, vp, vp]),
    "ug_moe_combine_topk": (i32, [vp, vp, vp, vp, vp, i32, i64, i32, i64, vp, vp, i64, i64, i64, vp, i64, i64, i64, i32, vp]),
    "ug_moe_dispatch_modulate": (i32, [vp, i64, vp, vp, i64, i64, vp, i32, i64, i64, i64, vp, vp]),
    "ug_moe_combine": (i32, [vp, vp, vp, vp, vp, i32, i64, vp, vp, i64, i64, i64, vp, i64, i64, i64, i32, vp]),
    "ug_conv2d_nhwc": (i32, [C.POINTER(ConvDesc), vp]),
    "ug_groupnorm_workspace_bytes": (i64, [i64, i64, i32]),
    "ug_groupnorm_nhwc": (i32, [vp, vp, vp, vp, vp, i64, i64, i64, i64, i32, f32, i32, vp]),
    "ug_softmax_rows": (i32, [vp, i64, vp, i64, i64, i64, f32, vp]),
    "ug_nchw_to_nhwc": (i32, [vp, vp, i64, i64, i64, i64, f32, f32, vp]),
    "ug_nhwc_to_nchw": (i32, [vp, vp, i64, i64, i64, i64, vp]),
    "ug_vae_sample": (i32, [vp, i64, vp, vp, i64, i64, i64, f32, f32, vp]),
    "ug_probe_mfma_bf16": (i32, [i32, i64, i64, vp, C.POINTER(C.c_double), vp]),
    "ug_pack_latents": (i32, [vp, vp, i64, i64, i64, i64, vp]),
    "ug_unpack_latents": (i32, [vp, vp, i64, i64, i64, i64, vp]),
}
# fp32 verification twins: `<name>_f32` has the signature of the function it mirrors (include/unigen_hip.h, last section)
SIGNATURES.update({
    "ug_transpose": (i32, [vp, i64, i64, vp, i64, i64, i64, i64, i64, i64, vp]),
    "ug_colsum": (i32, [vp, i64, vp, i64, vp, i64, i64, i64, i64, f32, vp, i64, vp]),
    "ug_colsum_workspace_bytes": (i64, [i64, i64, i64]),
    "ug_gate_residual": (i32, [vp, i64, vp, i64, vp, i64, i64, vp, i64, i64, i64, vp]),
    "ug_moe_gate_bwd": (i32, [vp, vp, vp, vp, i64, vp, i64, i64, i32, vp, i64, vp, vp]),
    "ug_moe_gate_bwd_slices": (i64, [i64]),
    "ug_gelu_tanh": (i32, [vp, vp, i64, vp]),
    "ug_gelu_tanh_bwd": (i32, [vp, vp, vp, i64, vp]),
    "ug_adaln_modulate_bwd": (i32, [vp, i64, vp, i64, vp, i64, i64, vp, i64, vp, i64, i64, f32, vp]),
    "ug_adaln_modulate_bwd_partials": (i64, [i64, i64]),
    "ug_qk_rmsnorm_rope_bwd": (i32, [vp, i64, vp, i64, vp, i64, vp, vp, vp, vp, i64, i64, i64, i32, i32, f32, vp]),
    "ug_qk_rmsnorm_rope_bwd_partials": (i64, [i64, i32]),
    "ug_row_lse": (i32, [vp, i64, vp, i64, i64, f32, vp]),
    "ug_attn_prob": (i32, [vp, i64, vp, vp, i64, i64, i64, i64, f32, vp]),
    "ug_attn_dscore": (i32, [vp, i64, vp, i64, vp, vp, i64, i64, i64, f32, vp]),
    "ug_rowdot": (i32, [vp, i64, vp, i64, vp, i64, i64, i64, vp]),
    "ug_flash_attn_bwd_workspace_bytes": (i64, [i64, i32, i64]),
    "ug_flash_attn_bwd": (i32, [vp, i64, i64] * 8 + [i64, i32, i64, i64, i32, f32, vp, vp, i64, vp]),
    "ug_flash_attn_fwd_lse": (i32, [vp, i64, i64, vp, i64, i64, vp, i64, i64, vp, i64, i64, i64, i32, i64, i64, i32, f32, vp, i64, vp]),
})
_F32_TWINS = {"ug_gate_residual_f32": "ug_gate_residual", "ug_moe_gate_bwd_f32": "ug_moe_gate_bwd", "ug_transpose_f32": "ug_transpose", "ug_colsum_f32": "ug_colsum", "ug_gelu_tanh_f32": "ug_gelu_tanh", "ug_gelu_tanh_bwd_f32": "ug_gelu_tanh_bwd",
              "ug_adaln_modulate_bwd_f32": "ug_adaln_modulate_bwd", "ug_qk_rmsnorm_rope_bwd_f32": "ug_qk_rmsnorm_rope_bwd",
              "ug_attn_prob_f32": "ug_attn_prob", "ug_attn_dscore_f32": "ug_attn_dscore", "ug_rowdot_f32": "ug_rowdot",
              "ug_gemm_f32": "ug_gemm_bf16", "ug_small_linear_f32": "ug_small_linear_bf16", "ug_adaln_modulate_f32": "ug_adaln_modulate",
              "ug_qk_rmsnorm_rope_f32": "ug_qk_rmsnorm_rope", "ug_flash_attn_fwd_f32": "ug_flash_attn_fwd", "ug_timestep_embed_f32": "ug_timestep_embed",
              "ug_euler_step_f32": "ug_euler_step", "ug_cfg_combine_f32": "ug_cfg_combine", "ug_add_f32": "ug_add_bf16",
              "ug_add_rowbcast_f32_f32": "ug_add_rowbcast_f32", "ug_gather_rows_f32": "ug_gather_rows", "ug_moe_gate_top1_f32": "ug_moe_gate_top1",
              "ug_moe_dispatch_modulate_f32": "ug_moe_dispatch_modulate", "ug_moe_combine_f32": "ug_moe_combine",
              "ug_moe_gate_top2_f32": "ug_moe_gate_top2", "ug_moe_gate_topk_f32": "ug_moe_gate_topk", "ug_moe_combine_topk_f32": "ug_moe_combine_topk",
              "ug_pack_latents_f32": "ug_pack_latents", "ug_unpack_latents_f32": "ug_unpack_latents",
              "ug_conv2d_nhwc_f32": "ug_conv2d_nhwc", "ug_groupnorm_nhwc_f32": "ug_groupnorm_nhwc", "ug_softmax_rows_f32": "ug_softmax_rows",
              "ug_nchw_to_nhwc_f32": "ug_nchw_to_nhwc", "ug_nhwc_to_nchw_f32": "ug_nhwc_to_nchw", "ug_vae_sample_f32": "ug_vae_sample"}
for _twin, _base in _F32_TWINS.items():
    SIGNATURES[_twin] = SIGNATURES[_base]

_lib = None


class UniGenHipError(RuntimeError):
    pass


def load() -> C.CDLL:
    """Load the HIP library (once). Raises if it has not been built: there is no CPU fallback."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        # a clean clone carries no binaries: build the product library once, in-tree (about 30 s); there is no CPU fallback for the hot path
        if os.environ.get("UG_LIB_PATH"):
            raise UniGenHipError(f"UG_LIB_PATH={LIB_PATH} not found (probe library: `python -m unigen_amd.build --probe`)")
        try:
            from . import build as _build
            _build.build()          # serialised across processes by a file lock; outputs appear atomically (os.replace)
        except Exception as e:      # no hipcc, compile error: fail loudly, never fall back
            raise UniGenHipError(f"{LIB_PATH} not found and `python -m unigen_amd.build` (hipcc --offload-arch=gfx950) failed: {e}. "
                                 "unigen_amd has no CPU fallback for the hot path.") from e
    # The library links libamdhip64 by SONAME. PyTorch-ROCm ships its own copy of the HIP runtime: it has to be in the process first so
    # that both resolve to ONE runtime (loaded the other way round, the system runtime and torch's each keep their own device state and
    # launches fail with "no ROCm-capable device is detected").
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc: int, what: str) -> None:
    if rc != UG_OK:
        msg = load().ug_last_error()
        raise UniGenHipError(f"{what} failed (code {rc}): {msg.decode() if msg else '?'}")
