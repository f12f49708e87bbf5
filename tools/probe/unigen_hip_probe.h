/* Entry points that exist only in the PROBE library (python -m unigen_amd.build --probe -> tools/probe/libunigen_hip_probe.so): kernels that were
 * built, measured and dropped from the product path, kept for A/B measurements. Not part of the product C ABI (include/unigen_hip.h). */
#pragma once
#include "../../include/unigen_hip.h"
#ifdef __cplusplus
extern "C" {
#endif
/* C[I][J] = sum_r A[r][I] * B[r][J], bf16 in / out, fp32 accumulation: dW = dY^T X of a Linear layer (A = dY [rows][out_features],
 * B = X [rows][in_features]) straight from the row-major operands - both MFMA fragments come from transposing LDS reads, no transposed copies.
 * I, J, lda, ldb multiples of 8. Measured 6 % slower per training step than two ug_transpose + the 256^2 ug_gemm_bf16 (round 2). */
int ug_gemm_tn_bf16(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, int64_t R, int64_t I, int64_t J, ug_stream_t stream);
#ifdef __cplusplus
}
#endif
