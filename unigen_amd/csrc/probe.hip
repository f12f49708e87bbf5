// Measured MFMA peak ("the builder must also report a measured MFMA-peak microbenchmark and use the measured
// sustained-clock peak as a second denominator"): a bare bf16 MFMA loop, operands in registers, one wave per SIMD on every CU,
// pseudo-random (full-range, mixed-sign) operand values - zero or trivial operands let the chip hold a higher clock than any real
// kernel sees (MI355X_MICROARCH.md, DVFS give-back). bench.py times it with HIP events: FLOP/s = what the matrix pipes deliver at
// the clock the chip holds under an MFMA-only load, i.e. the ceiling no LDS / HBM / VALU work can be added to for free.
#include "ug_common.h"

namespace {

__device__ __forceinline__ unsigned hash32(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
// 8 bf16 values in [-2, 2) with random mantissas and signs
__device__ __forceinline__ bf16x8 rand_frag(unsigned seed) {
    bf16x8 f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const unsigned h = hash32(seed * 8u + (unsigned)i);
        f[i] = (short)((h & 0x80ffu) | 0x3f00u | ((h >> 16) & 0x0080u));     // sign | exponent 126..127 | 7 mantissa bits
    }
    return f;
}

template <int SHAPE>   // 0: v_mfma_f32_32x32x16_bf16 (4 accumulators x 16 regs), 1: v_mfma_f32_16x16x32_bf16 (8 accumulators x 4 regs)
__global__ __launch_bounds__(256) void mfma_probe_kernel(float* __restrict__ out, int iters) {
    const unsigned tid = blockIdx.x * 256u + threadIdx.x;
    const bf16x8 a = rand_frag(tid * 3u), b0 = rand_frag(tid * 3u + 1u), b1 = rand_frag(tid * 3u + 2u);
    float s = 0.f;
    // inline asm: the chains stay exactly as written (as builtins hipcc merges accumulators that compute equal values and shuffles the
    // rest through v_accvgpr copies); back-to-back accumulate chains need no nops (MFMA -> same-shape MFMA taking the result as C)
    if constexpr (SHAPE == 0) {
        f32x16 c0, c1, c2, c3;
#pragma unroll
        for (int i = 0; i < 16; ++i) { c0[i] = 0.f; c1[i] = 0.f; c2[i] = 0.f; c3[i] = 0.f; }
        for (int it = 0; it < iters; ++it) {
            asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c0) : "v"(a), "v"(b0));
            asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c1) : "v"(a), "v"(b1));
            asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c2) : "v"(b0), "v"(a));
            asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c3) : "v"(b1), "v"(a));
        }
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
#pragma unroll
        for (int i = 0; i < 16; ++i) s += c0[i] + c1[i] + c2[i] + c3[i];
    } else {
        f32x4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = c0, c2 = c0, c3 = c0, c4 = c0, c5 = c0, c6 = c0, c7 = c0;
        for (int it = 0; it < iters; ++it) {
            asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c0) : "v"(a), "v"(b0));
            asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c1) : "v"(a), "v"(b1));
            asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c2) : "v"(b0), "v"(a));
            asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c3) : "v"(b1), "v"(a));
            asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c4) : "v"(b0), "v"(b1));
            asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c5) : "v"(b1), "v"(b0));
            asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c6) : "v"(a), "v"(a));
            asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c7) : "v"(b0), "v"(b0));
        }
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
        const f32x4 t = c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7;
        s = t[0] + t[1] + t[2] + t[3];
    }
    out[tid] = s;       // keeps the chain alive; never read
}

}  // namespace

// One launch of the probe on `blocks` workgroups of 4 waves (one per SIMD). FLOPs of the launch are returned through *flops_out
// (host pointer, may be NULL): blocks * 4 waves * iters * MFMAs per iteration * 2 * M * N * K. `scratch` needs blocks * 256 floats.
extern "C" int ug_probe_mfma_bf16(int32_t shape, int64_t blocks, int64_t iters, void* scratch, double* flops_out_host, ug_stream_t stream) {
    UG_REQUIRE((shape == 0 || shape == 1) && blocks > 0 && blocks < (1 << 20) && iters > 0 && iters < (1ll << 31) && scratch, UG_ERR_BAD_SHAPE,
               "ug_probe_mfma_bf16: bad arguments");
    if (shape == 0) hipLaunchKernelGGL(mfma_probe_kernel<0>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (float*)scratch, (int)iters);
    else hipLaunchKernelGGL(mfma_probe_kernel<1>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (float*)scratch, (int)iters);
    UG_CHECK_LAUNCH("ug_probe_mfma_bf16");
    if (flops_out_host) {
        const double per_iter = shape == 0 ? 4.0 * 2.0 * 32 * 32 * 16 : 8.0 * 2.0 * 16 * 16 * 32;
        *flops_out_host = (double)blocks * 4.0 * (double)iters * per_iter;
    }
    return UG_OK;
}
