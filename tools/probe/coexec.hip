// Probe: do one wave's MFMAs and another wave's VALU instructions on the SAME SIMD execute concurrently on gfx950?
// Waves 0-3 of a 512-thread workgroup (one per SIMD) run a chain of v_mfma_f32_32x32x16_bf16; waves 4-7 run a VALU stream of one kind.
// Each role is timed alone and together (s_memtime in the wave). Build: hipcc --offload-arch=gfx950 -O2 coexec.hip -o bin/coexec
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((__vector_size__(8 * sizeof(short)))) short bf16x8;
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16;
typedef __attribute__((__vector_size__(2 * sizeof(float)))) float f32x2;

template <int KIND>
__device__ __forceinline__ void valu_body(float (&r)[8]) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        if constexpr (KIND == 0) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(r[i]));
        if constexpr (KIND == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(r[i]));
        if constexpr (KIND == 2) asm volatile("v_mul_f32 %0, %0, %0" : "+v"(r[i]));
        if constexpr (KIND == 4) asm volatile("v_xor_b32 %0, %0, %0" : "+v"(r[i]));
    }
    if constexpr (KIND == 3) {
#pragma unroll
        for (int i = 0; i < 8; i += 2) {
            f32x2 p = {r[i], r[i + 1]};
            asm volatile("v_pk_mul_f32 %0, %0, %0" : "+v"(p));
            asm volatile("v_pk_mul_f32 %0, %0, %0" : "+v"(p));
            r[i] = p[0]; r[i + 1] = p[1];
        }
    }
    if constexpr (KIND == 5) {
#pragma unroll
        for (int i = 0; i < 8; i += 2) {
            unsigned w;
            asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(w) : "v"(r[i]), "v"(r[i + 1]));
            asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(w) : "v"(r[i + 1]), "v"(r[i]));
            r[i] = __uint_as_float(w);
        }
    }
}

// roles: bit 0 = waves 0-3 run MFMA, bit 1 = waves 4-7 run VALU; a role that is off exits at once
template <int KIND>
__global__ __launch_bounds__(512) void coexec(int roles, int iters, unsigned long long* out, float* sink) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const bool mf = wave < 4;
    if (mf && !(roles & 1)) return;
    if (!mf && !(roles & 2)) return;
    unsigned long long t0, t1;
    if (mf) {
        f32x16 a0 = {}, a1 = {};
        bf16x8 x = {1, 2, 3, 4, 5, 6, 7, 8}, y = {1, 1, 1, 1, 1, 1, 1, 1};
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(y, x, a1, 0, 0, 0);
            }
        }
        asm volatile("s_nop 15\n\ts_nop 15" : "+v"(a0), "+v"(a1));
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
        if (a0[0] + a1[0] == 12345.f) sink[lane] = a0[1];
    } else {
        float r[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) r[i] = 1.0f + 0.001f * (lane + i);
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 8; ++u) valu_body<KIND>(r);
        }
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
        float s = 0; for (int i = 0; i < 8; ++i) s += r[i];
        if (s == 12345.f) sink[lane] = s;
    }
    if (lane == 0 && blockIdx.x == 0) out[wave] = t1 - t0;
}

template <int KIND>
void run(const char* name, unsigned long long* dout, float* sink) {
    const int iters = 2000;
    double res[3][2];
    for (int roles = 1; roles <= 3; ++roles) {
        (void)hipMemset(dout, 0, 64);
        hipLaunchKernelGGL(coexec<KIND>, dim3(256), dim3(512), 0, 0, roles, iters, dout, sink);
        unsigned long long h[8];
        (void)hipMemcpy(h, dout, 64, hipMemcpyDeviceToHost);
        res[roles - 1][0] = (double)h[0] / (iters * 16.0);           // cycles per MFMA
        res[roles - 1][1] = (double)h[4] / (iters * 64.0);           // cycles per VALU instruction (64 per iteration)
    }
    printf("COEXEC %-18s  MFMA alone %6.1f ticks/MFMA | VALU alone %6.2f ticks/inst | together: MFMA %6.1f, VALU %6.2f\n", name, res[0][0], res[1][1], res[2][0], res[2][1]);
}

int main() {
    unsigned long long* dout; float* sink;
    (void)hipMalloc(&dout, 64); (void)hipMalloc(&sink, 1024);
    run<0>("v_fma_f32", dout, sink);
    run<1>("v_exp_f32", dout, sink);
    run<2>("v_mul_f32", dout, sink);
    run<3>("v_pk_mul_f32", dout, sink);
    run<4>("v_xor_b32", dout, sink);
    run<5>("v_cvt_pk_bf16_f32", dout, sink);
    return 0;
}
