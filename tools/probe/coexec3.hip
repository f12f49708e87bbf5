// Probe 3: the X|Y stagger in isolation. 8 waves; waves 0-3 (group A) and 4-7 (group B) alternate, one segment apart, between an
// X segment (24 MFMAs 32x32x16 fed by 16 ds_read_b128 + 16 ds_read_b64_tr_b16, software-pipelined two steps ahead) and a Y segment (the
// backward's P (dP - delta) -> bf16 arithmetic for 32 scores per lane), with an s_barrier after every segment - the loop structure of an
// attention kernel without its data movement. Prints ticks per tile (two segments) for: X only (Y empty), Y only, both, and both without
// the LDS reads. Flags add one feature of the real kernel at a time: 1 = two LDS-DMAs per wave per tile (issued in the Y segment by
// group B and in the X segment by group A, waited for two segments later), 2 = s_waitcnt lgkmcnt(0) ahead of each barrier.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((__vector_size__(8 * sizeof(short)))) short bf16x8;
typedef __attribute__((__vector_size__(4 * sizeof(short)))) short bf16x4;
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16;
typedef __attribute__((address_space(3))) bf16x4* lds_b64_ptr;
typedef __attribute__((address_space(3))) void* lptr_t;

__device__ __forceinline__ unsigned pack2bf(float a, float b) { unsigned r; asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ bf16x8 tr_pair(const unsigned char* lo, const unsigned char* hi) {
    const bf16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_b64_ptr)lo), b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_b64_ptr)hi);
    return (bf16x8){a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
}

template <bool DO_X, bool DO_Y, bool READS, int FLAGS>
__global__ __launch_bounds__(512, 2) void stagger(int iters, const unsigned char* gsrc, unsigned long long* out, float* sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 65536 / 4; i += 512) ((unsigned*)smem)[i] = 0x3f803f80u;
    __syncthreads();
    const bool groupA = wave < 4;
    f32x16 acc[2] = {}, x1[2] = {}, x2[2] = {};
    bf16x8 f1[4], f2[4], zf[2][2];
#pragma unroll
    for (int s = 0; s < 4; ++s) { f1[s] = (bf16x8){1, 1, 1, 1, 1, 1, 1, 1}; f2[s] = f1[s]; asm volatile("" : "+v"(f1[s]), "+v"(f2[s])); }
#pragma unroll
    for (int i = 0; i < 4; ++i) zf[i >> 1][i & 1] = f1[0];
    float c = 0.1f, lse = 0.3f, dl = 0.01f;
    asm volatile("" : "+v"(lse), "+v"(dl), "+s"(c));
    const unsigned char* rb = smem + lane * 16;
    const unsigned char* tb = smem + 32768 + lane * 8;
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lptr_t)smem) + 49152 + wave * 2048;
    const unsigned char* gp = gsrc + (size_t)blockIdx.x * 65536 + wave * 2048 + lane * 16;

    auto pin_x = [&]() {
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) { asm volatile("" : "+v"(x1[kb])); asm volatile("" : "+v"(x2[kb])); }
    };
    auto pin_z = [&]() {
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) { asm volatile("" : "+v"(zf[kb][0])); asm volatile("" : "+v"(zf[kb][1])); }
    };
    auto do_x = [&]() {
        if constexpr (!DO_X) { pin_x(); return; }
        bf16x8 tf[3][2], ab[3][2];
        auto rd = [&](int j) {
            if constexpr (READS) {
                if (j < 4) { tf[j % 3][0] = tr_pair(tb + j * 2048, tb + j * 2048 + 512); tf[j % 3][1] = tr_pair(tb + j * 2048 + 1024, tb + j * 2048 + 1536); }
                else { ab[j % 3][0] = *(const bf16x8*)(rb + (j - 4) * 2048); ab[j % 3][1] = *(const bf16x8*)(rb + (j - 4) * 2048 + 1024); }
            } else {
                if (j < 4) { tf[j % 3][0] = f1[0]; tf[j % 3][1] = f2[0]; } else { ab[j % 3][0] = f1[1]; ab[j % 3][1] = f2[1]; }
            }
        };
        rd(0); rd(1);
#pragma unroll
        for (int j = 0; j < 12; ++j) {
            if (j + 2 < 12) rd(j + 2);
            __builtin_amdgcn_sched_barrier(0);
            if (j < 4) {
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tf[j % 3][0], zf[j >> 1][j & 1], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tf[j % 3][1], zf[j >> 1][j & 1], acc[1], 0, 0, 0);
            } else {
                const int kb = (j - 4) / 4, s = (j - 4) % 4;
                if (s == 0) { x1[kb] = (f32x16){}; x2[kb] = (f32x16){}; }
                x1[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab[j % 3][0], f1[s], x1[kb], 0, 0, 0);
                x2[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab[j % 3][1], f2[s], x2[kb], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        pin_x();
    };
    auto do_y = [&]() {
        if constexpr (!DO_Y) { pin_z(); return; }
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            float z[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) { const float p = __builtin_amdgcn_exp2f(fmaf(x1[kb][i], c, -lse)); z[i] = p * (x2[kb][i] - dl); }
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                typedef __attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned u32x4;
                u32x4 w;
                w[0] = pack2bf(z[8 * s2 + 0], z[8 * s2 + 1]); w[1] = pack2bf(z[8 * s2 + 2], z[8 * s2 + 3]);
                w[2] = pack2bf(z[8 * s2 + 4], z[8 * s2 + 5]); w[3] = pack2bf(z[8 * s2 + 6], z[8 * s2 + 7]);
                zf[kb][s2] = __builtin_bit_cast(bf16x8, w);
            }
        }
        pin_z();
    };
    auto dma2 = [&]() {
        if constexpr (FLAGS & 1) {
            asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gp), "s"(lds0) : "memory");
            asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gp + 1024), "s"(lds0 + 1024) : "memory");
        }
    };
    auto dma_wait = [&]() { if constexpr (FLAGS & 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); };
    auto bar = [&]() {
        if constexpr (FLAGS & 2) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0);
    };
    unsigned long long t0, t1;
    bar();
    if (!groupA) bar();
    do_x();
    bar();
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int it = 0; it < iters; ++it) {
        if (!groupA) dma2();
        do_y();
        if (groupA) dma_wait();
        bar();
        if (groupA) dma2();
        do_x();
        if (!groupA) dma_wait();
        bar();
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    if (groupA) bar();
    float s = acc[0][0] + acc[1][0] + x1[0][0] + x2[1][3];
    if (s == 12345.f) sink[lane] = s;
    if (lane == 0 && blockIdx.x == 17) out[wave] = t1 - t0;
}

template <bool DO_X, bool DO_Y, bool READS, int FLAGS>
double run(const unsigned char* gsrc, unsigned long long* dout, float* sink) {
    const int iters = 400;
    (void)hipFuncSetAttribute((const void*)stagger<DO_X, DO_Y, READS, FLAGS>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    (void)hipMemset(dout, 0, 64);
    hipLaunchKernelGGL((stagger<DO_X, DO_Y, READS, FLAGS>), dim3(256), dim3(512), 65536, 0, iters, gsrc, dout, sink);
    unsigned long long h[8];
    (void)hipMemcpy(h, dout, 64, hipMemcpyDeviceToHost);
    return (double)h[0] / iters;
}

int main() {
    unsigned long long* dout; float* sink; unsigned char* gsrc;
    (void)hipMalloc(&dout, 64); (void)hipMalloc(&sink, 1024); (void)hipMalloc(&gsrc, 256 * 65536); (void)hipMemset(gsrc, 0, 256 * 65536);
    printf("COEXEC3 ticks per tile (two segments, two barriers; 24 MFMAs = 768 cycles and ~800 cycles of VALU per wave per tile)\n");
    printf("COEXEC3 no LDS reads : X only %6.0f | Y only %6.0f | X and Y %6.0f\n", run<true, false, false, 0>(gsrc, dout, sink), run<false, true, false, 0>(gsrc, dout, sink), run<true, true, false, 0>(gsrc, dout, sink));
    printf("COEXEC3 LDS-fed X    : X only %6.0f | Y only %6.0f | X and Y %6.0f\n", run<true, false, true, 0>(gsrc, dout, sink), run<false, true, true, 0>(gsrc, dout, sink), run<true, true, true, 0>(gsrc, dout, sink));
    printf("COEXEC3 + lgkmcnt(0) at barriers : X and Y %6.0f\n", run<true, true, true, 2>(gsrc, dout, sink));
    printf("COEXEC3 + 2 LDS-DMAs per wave per tile : X only %6.0f | Y only %6.0f | X and Y %6.0f\n", run<true, false, true, 3>(gsrc, dout, sink), run<false, true, true, 3>(gsrc, dout, sink), run<true, true, true, 3>(gsrc, dout, sink));
    return 0;
}
