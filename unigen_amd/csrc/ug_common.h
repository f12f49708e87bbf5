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

// integer switch from the environment: cached per call site name, re-read on every call when UG_ENV_DYNAMIC=1 (A/B tools). The PRODUCT library
// uses it only where a switch selects between two SHIPPED kernels that the dispatcher otherwise picks by shape (tests pin each of them):
// UG_GEMM_FORCE_TILE, UG_ADALN_FAST, UG_GN_FAST, UG_SOFTMAX_FAST, UG_CONV256 / UG_CONV256_MIN_TILES.
int ug_env_int(const char* name, int dflt);
// Tuning constants and measured-and-dropped kernel variants: compile-time constants in the product library (one kernel per dispatch decision);
// `python -m unigen_amd.build --probe` builds tools/probe/libunigen_hip_probe.so with -DUG_PROBE_BUILD, where they are environment switches
// again and the dropped variants (one-wave-per-SIMD GEMMs, cross-tile stream, lock-step / 4-wave / one-wave-per-SIMD attention, register-staged
// backward, ...) are compiled in for A/B measurements (tools/probe/README.md).
#ifdef UG_PROBE_BUILD
#define UG_TUNE(name, dflt) ug_env_int(name, dflt)
#else
#define UG_TUNE(name, dflt) (dflt)
#endif

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

// ---- element traits: one kernel source for the product path (bf16 storage, the reference's bf16 rounding points) and the
// fp32 VERIFICATION path (`*_f32` entry points: fp32 storage, no intermediate rounding). 8 elements per access in both. ------
template <typename T> struct ElemT;
template <> struct ElemT<bf16_t> {
    static constexpr bool kF32 = false;
    static __device__ __forceinline__ void load8(const bf16_t* p, float* f) {
        const u32x4 v = *(const u32x4*)p;
        f[0] = bflo(v.x); f[1] = bfhi(v.x); f[2] = bflo(v.y); f[3] = bfhi(v.y);
        f[4] = bflo(v.z); f[5] = bfhi(v.z); f[6] = bflo(v.w); f[7] = bfhi(v.w);
    }
    static __device__ __forceinline__ void store8(bf16_t* p, const float* f) {
        u32x4 v;
        v.x = pack2bf(f[0], f[1]); v.y = pack2bf(f[2], f[3]); v.z = pack2bf(f[4], f[5]); v.w = pack2bf(f[6], f[7]);
        *(u32x4*)p = v;
    }
    static __device__ __forceinline__ void zero8(bf16_t* p) { *(u32x4*)p = (u32x4){0u, 0u, 0u, 0u}; }
    static __device__ __forceinline__ void load2(const bf16_t* p, float* f) { const unsigned v = *(const unsigned*)p; f[0] = bflo(v); f[1] = bfhi(v); }
    static __device__ __forceinline__ void store2(bf16_t* p, const float* f) { *(unsigned*)p = pack2bf(f[0], f[1]); }
    static __device__ __forceinline__ float rnd(float x) { return rbf(x); }          // a bf16 tensor op's output
    static __device__ __forceinline__ float ld(const bf16_t* p) { return bf2f(*p); }
    static __device__ __forceinline__ void st(bf16_t* p, float x) { *p = f2bf(x); }
};
template <> struct ElemT<float> {
    static constexpr bool kF32 = true;
    static __device__ __forceinline__ void load8(const float* p, float* f) {
        const f32x4 a = *(const f32x4*)p, b = *(const f32x4*)(p + 4);
        f[0] = a[0]; f[1] = a[1]; f[2] = a[2]; f[3] = a[3]; f[4] = b[0]; f[5] = b[1]; f[6] = b[2]; f[7] = b[3];
    }
    static __device__ __forceinline__ void store8(float* p, const float* f) {
        *(f32x4*)p = (f32x4){f[0], f[1], f[2], f[3]};
        *(f32x4*)(p + 4) = (f32x4){f[4], f[5], f[6], f[7]};
    }
    static __device__ __forceinline__ void zero8(float* p) { *(f32x4*)p = (f32x4){0.f, 0.f, 0.f, 0.f}; *(f32x4*)(p + 4) = (f32x4){0.f, 0.f, 0.f, 0.f}; }
    static __device__ __forceinline__ void load2(const float* p, float* f) { const float2 v = *(const float2*)p; f[0] = v.x; f[1] = v.y; }
    static __device__ __forceinline__ void store2(float* p, const float* f) { *(float2*)p = make_float2(f[0], f[1]); }
    static __device__ __forceinline__ float rnd(float x) { return x; }
    static __device__ __forceinline__ float ld(const float* p) { return *p; }
    static __device__ __forceinline__ void st(float* p, float x) { *p = x; }
};

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
