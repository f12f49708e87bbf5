// Shared device/host helpers for libunigen_hip.so (gfx950 only, wave64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/unigen_hip.h"

typedef unsigned short bf16_t;  // raw bf16 bits
typedef __attribute__((ext_vector_type(8))) short bf16x8;   // MFMA A/B fragment (4 VGPRs)
typedef __attribute__((ext_vector_type(4))) short bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

#define UG_WAVE 64

// ---- error plumbing (host) -------------------------------------------------------------
void ug_set_error(const char* fmt, ...);
#define UG_FAIL(code, ...) do { ug_set_error(__VA_ARGS__); return (code); } while (0)
#define UG_REQUIRE(cond, code, ...) do { if (!(cond)) UG_FAIL(code, __VA_ARGS__); } while (0)
#define UG_CHECK_LAUNCH(name) do { hipError_t e_ = hipGetLastError(); \
    if (e_ != hipSuccess) UG_FAIL(UG_ERR_HIP, "%s: launch failed: %s", name, hipGetErrorString(e_)); } while (0)

static inline bool ug_aligned(const void* p, size_t a) { return (reinterpret_cast<uintptr_t>(p) % a) == 0; }

// ---- bf16 <-> f32 (device) -------------------------------------------------------------
__device__ __forceinline__ float bf2f(bf16_t u) { return __uint_as_float(((unsigned)u) << 16); }
__device__ __forceinline__ bf16_t f2bf(float f) {  // RNE, NaN-preserving (v_cvt_pk_bf16_f32)
    __bf16 b = (__bf16)f;
    return __builtin_bit_cast(bf16_t, b);
}
__device__ __forceinline__ float rbf(float f) { return bf2f(f2bf(f)); }  // round through bf16
__device__ __forceinline__ unsigned pack2bf(float lo, float hi) {
    typedef __attribute__((ext_vector_type(2))) __bf16 bf2v;
    typedef __attribute__((ext_vector_type(2))) float f2v;
    f2v x = {lo, hi};
    bf2v y = __builtin_convertvector(x, bf2v);
    return __builtin_bit_cast(unsigned, y);
}
__device__ __forceinline__ float bflo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bfhi(unsigned u) { return __uint_as_float(u & 0xffff0000u); }

// logical row -> physical row of a concatenated buffer (see unigen_hip.h "row map")
__device__ __forceinline__ int64_t ug_rowmap(int64_t m, int64_t rpb, int64_t bstride) {
    if (rpb <= 0) return m;
    int64_t b = m / rpb;
    return b * bstride + (m - b * rpb);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
