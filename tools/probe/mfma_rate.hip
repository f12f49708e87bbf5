// Probe: cycles per v_mfma_f32_32x32x16_bf16 on one wave per SIMD for different operand register classes / accumulator rotations.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int MODE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void k(float* out, unsigned long long* cyc, int iters) {
    bf16x8 a, b0, b1;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b0[i] = (__bf16)(i * 0.5f); b1[i] = (__bf16)(i * 0.25f); }
    f32x16 c0, c1, c2, c3;
    for (int i = 0; i < 16; ++i) { c0[i] = 0; c1[i] = 0; c2[i] = 0; c3[i] = 0; }
    if (MODE == 1 || MODE == 3) { asm volatile("" : "+a"(b0), "+a"(b1)); }
    if (MODE == 4) { asm volatile("" : "+a"(c0), "+a"(c1), "+a"(c2), "+a"(c3)); }
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {          // asm, VGPR acc, VGPR B, 4 accumulators rotating
            asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c0) : "v"(a), "v"(b0));
            asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c1) : "v"(a), "v"(b1));
            asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c2) : "v"(a), "v"(b0));
            asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c3) : "v"(a), "v"(b1));
        } else if (MODE == 1) {   // asm, VGPR acc, AGPR B
            asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c0) : "v"(a), "a"(b0));
            asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c1) : "v"(a), "a"(b1));
            asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c2) : "v"(a), "a"(b0));
            asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c3) : "v"(a), "a"(b1));
        } else if (MODE == 2) {   // asm, VGPR, single accumulator chain
            asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c0) : "v"(a), "v"(b0));
            asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c0) : "v"(a), "v"(b1));
            asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c0) : "v"(a), "v"(b0));
            asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c0) : "v"(a), "v"(b1));
        } else if (MODE == 3) {   // asm, VGPR acc, AGPR B, two accumulators alternating
            asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c0) : "v"(a), "a"(b0));
            asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c1) : "v"(a), "a"(b1));
            asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c0) : "v"(a), "a"(b0));
            asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c1) : "v"(a), "a"(b1));
        } else if (MODE == 4) {   // asm, AGPR acc, VGPR B, 4 accumulators
            asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c0) : "v"(a), "v"(b0));
            asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c1) : "v"(a), "v"(b1));
            asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c2) : "v"(a), "v"(b0));
            asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c3) : "v"(a), "v"(b1));
        } else {                  // builtin
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b0, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b1, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b0, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b1, c3, 0, 0, 0);
        }
    }
    asm volatile("s_nop 15\n s_nop 15\n s_nop 15" ::: "memory");
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < 16; ++i) s += c0[i] + c1[i] + c2[i] + c3[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE> void run(const char* name, int nblocks) {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, nblocks * 256 * sizeof(float)); hipMalloc(&cyc, nblocks * sizeof(unsigned long long));
    const int iters = 2000;
    k<MODE><<<nblocks, 256>>>(out, cyc, iters);
    k<MODE><<<nblocks, 256>>>(out, cyc, iters);
    hipDeviceSynchronize();
    unsigned long long h[8]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    printf("%-46s blocks %3d: %.2f cycles / MFMA\n", name, nblocks, (double)h[0] / (4.0 * iters));
    hipFree(out); hipFree(cyc);
}
int main() {
    for (int nb : {8, 256}) {
        run<0>("asm VGPR acc, VGPR B, 4 accs", nb);
        run<1>("asm VGPR acc, AGPR B, 4 accs", nb);
        run<2>("asm VGPR acc, VGPR B, 1 acc chain", nb);
        run<3>("asm VGPR acc, AGPR B, 2 accs alternating", nb);
        run<4>("asm AGPR acc, VGPR B, 4 accs", nb);
        run<5>("builtin, 4 accs", nb);
    }
    return 0;
}
