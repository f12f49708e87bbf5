// Probe 4 (round 3): would the attention forward gain from v_mfma_f32_16x16x32_bf16 instead of v_mfma_f32_32x32x16_bf16? The two shapes have the
// same MACs per cycle on paper, but the bare-MFMA peaks measured on this chip differ (~2.05 vs ~1.85 PFLOP/s - power). This is
// the X | Y stagger of flash_attn_kernel<128> as a skeleton: 8 waves, groups A / B one segment apart; X = P.V of tile t then S^T of tile t + 1
// (16 transposed-read pairs + 16 ds_read_b128, fragments two steps ahead) with either 32 MFMAs 32x32x16 or 64 MFMAs 16x16x32 - the same MACs,
// the same LDS reads, the same registers; Y = the online-softmax arithmetic of 32 scores per lane; two barriers per tile; LDS and registers hold
// RANDOM bf16 (the power of an MFMA depends on its data). One workgroup per CU, long enough for the clock to settle; wall time per tile.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/probe/bin/coexec4 tools/probe/coexec4.hip ; run: tools/probe/bin/coexec4
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __attribute__((__vector_size__(8 * sizeof(short)))) short bf16x8;
typedef __attribute__((__vector_size__(4 * sizeof(short)))) short bf16x4;
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16;
typedef __attribute__((__vector_size__(4 * sizeof(float)))) float f32x4;
typedef __attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned u32x4;
typedef __attribute__((address_space(3))) bf16x4* lds_b64_ptr;

__device__ __forceinline__ unsigned pack2bf(float a, float b) { unsigned r; asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ bf16x8 tr_pair(const unsigned char* lo, const unsigned char* hi) {
    const bf16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_b64_ptr)lo), b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_b64_ptr)hi);
    return (bf16x8){a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
}

template <int SHAPE /* 32 | 16 */, bool DO_Y>
__global__ __launch_bounds__(512, 2) void stagger(int iters, const unsigned* gsrc, float* sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 65536 / 4; i += 512) ((unsigned*)smem)[i] = gsrc[i];       // random bf16 in (-2, 2)
    __syncthreads();
    const bool groupA = wave < 4;
    bf16x8 qf[8], pf[4];
#pragma unroll
    for (int s = 0; s < 8; ++s) qf[s] = *(const bf16x8*)(smem + ((lane * 16 + s * 1024 + wave * 8192) & 65535));
#pragma unroll
    for (int s = 0; s < 4; ++s) pf[s] = *(const bf16x8*)(smem + ((lane * 16 + s * 1024 + 32768) & 65535));
    float c = 0.05f, mc = 0.3f, l_run = 0.f;
    asm volatile("" : "+v"(mc), "+s"(c));
    const unsigned char* rb = smem + lane * 16;
    const unsigned char* tb = smem + 32768 + lane * 8;
    float sc[32];                                                  // the scores handed from X to Y
#pragma unroll
    for (int i = 0; i < 32; ++i) sc[i] = 0.f;
    constexpr int NACC = SHAPE == 32 ? 4 : 16;
    f32x16 o32[SHAPE == 32 ? 4 : 1];
    f32x4 o16[SHAPE == 16 ? 16 : 1];
#pragma unroll
    for (int i = 0; i < (SHAPE == 32 ? 4 : 1); ++i) o32[i] = (f32x16){};
#pragma unroll
    for (int i = 0; i < (SHAPE == 16 ? 16 : 1); ++i) o16[i] = (f32x4){};
    (void)NACC;
    auto do_x = [&]() __attribute__((always_inline)) {
        bf16x8 fr[3];
        auto rd = [&](int j) __attribute__((always_inline)) {
            if (j < 16) fr[j % 3] = tr_pair(tb + (j & 15) * 1024, tb + (j & 15) * 1024 + 512);
            else fr[j % 3] = *(const bf16x8*)(rb + ((j - 16) & 15) * 2048);
        };
        rd(0); rd(1);
        if constexpr (SHAPE == 32) {
            f32x16 s32[2] = {(f32x16){}, (f32x16){}};
#pragma unroll
            for (int j = 0; j < 32; ++j) {
                if (j + 2 < 32) rd(j + 2);
                __builtin_amdgcn_sched_barrier(0);
                if (j < 16) o32[j >> 2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[j % 3], pf[j & 3], o32[j >> 2], 0, 0, 0);
                else s32[(j - 16) >> 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[j % 3], qf[(j - 16) & 7], s32[(j - 16) >> 3], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int i = 0; i < 32; ++i) sc[i] = s32[i >> 4][i & 15];
        } else {
            f32x4 s16[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) s16[i] = (f32x4){};
#pragma unroll
            for (int j = 0; j < 32; ++j) {                         // one fragment read feeds TWO 16x16x32 MFMAs (both query column blocks)
                if (j + 2 < 32) rd(j + 2);
                __builtin_amdgcn_sched_barrier(0);
                if (j < 16) {
                    o16[2 * (j >> 1)] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr[j % 3], pf[(j & 1) * 2], o16[2 * (j >> 1)], 0, 0, 0);
                    o16[2 * (j >> 1) + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr[j % 3], pf[(j & 1) * 2 + 1], o16[2 * (j >> 1) + 1], 0, 0, 0);
                } else {
                    const int kb = (j - 16) >> 2, s = (j - 16) & 3;
                    s16[2 * kb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr[j % 3], qf[2 * s], s16[2 * kb], 0, 0, 0);
                    s16[2 * kb + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr[j % 3], qf[2 * s + 1], s16[2 * kb + 1], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int i = 0; i < 32; ++i) sc[i] = s16[i >> 2][i & 3];
        }
#pragma unroll
        for (int i = 0; i < 32; ++i) asm volatile("" : "+v"(sc[i]));
    };
    auto do_y = [&]() __attribute__((always_inline)) {
        if constexpr (!DO_Y) return;
        float tmax = sc[0];
#pragma unroll
        for (int i = 1; i < 32; i += 2) tmax = __builtin_fmaxf(__builtin_fmaxf(tmax, sc[i]), sc[(i + 1) & 31]);
        asm volatile("" : "+v"(tmax));
        float p[32];
#pragma unroll
        for (int i = 0; i < 32; ++i) { p[i] = __builtin_amdgcn_exp2f(fmaf(sc[i], c, -mc) * 1e-3f); l_run += p[i]; }
#pragma unroll
        for (int s2 = 0; s2 < 4; ++s2) {
            u32x4 w;
            w[0] = pack2bf(p[8 * s2 + 0], p[8 * s2 + 1]); w[1] = pack2bf(p[8 * s2 + 2], p[8 * s2 + 3]);
            w[2] = pack2bf(p[8 * s2 + 4], p[8 * s2 + 5]); w[3] = pack2bf(p[8 * s2 + 6], p[8 * s2 + 7]);
            bf16x8 np = __builtin_bit_cast(bf16x8, w);
            // keep the P operand random (the exponentials above are ~1): mix the new bits into the old operand
            pf[s2] = pf[s2] ^ (np & (bf16x8){1, 1, 1, 1, 1, 1, 1, 1});
            asm volatile("" : "+v"(pf[s2]));
        }
    };
    auto bar = [&]() { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); };
    bar();
    if (!groupA) bar();
    do_x();
    bar();
    for (int it = 0; it < iters; ++it) {
        do_y();
        bar();
        do_x();
        bar();
    }
    if (groupA) bar();
    float s = l_run + sc[3];
    if constexpr (SHAPE == 32) {               // every accumulator is live (an unused one would take its MFMAs with it)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) s += o32[i][e];
    } else {
#pragma unroll
        for (int i = 0; i < 16; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) s += o16[i][e];
    }
    if (s == 12345.f) sink[lane] = s;
}

template <int SHAPE, bool DO_Y>
double run(const unsigned* gsrc, float* sink, int iters) {
    (void)hipFuncSetAttribute((const void*)stagger<SHAPE, DO_Y>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((stagger<SHAPE, DO_Y>), dim3(256), dim3(512), 65536, 0, iters / 10, gsrc, sink);      // warm the clock
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((stagger<SHAPE, DO_Y>), dim3(256), dim3(512), 65536, 0, iters, gsrc, sink);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    return (double)ms * 1e6 / iters;       // ns per tile
}

int main() {
    std::vector<unsigned> h(65536 / 4);
    srand(1);
    for (auto& w : h) {           // two random bf16 in (-2, 2): sign, exponent 125..127, 7 random mantissa bits
        unsigned v = 0;
        for (int k = 0; k < 2; ++k) { const unsigned s = rand() & 1, e = 125 + rand() % 3, m = rand() & 127; v |= ((s << 15) | (e << 7) | m) << (16 * k); }
        w = v;
    }
    unsigned* g; float* sink;
    (void)hipMalloc(&g, 65536); (void)hipMalloc(&sink, 256);
    (void)hipMemcpy(g, h.data(), 65536, hipMemcpyHostToDevice);
    const int iters = 40000;
    const double flop_tile = 256.0 * 8 * 32 * 2.0 * 32 * 32 * 16;        // CUs x waves x MFMAs x MACs x 2
    for (int rep = 0; rep < 3; ++rep) {
        const double a = run<32, true>(g, sink, iters), b = run<16, true>(g, sink, iters);
        const double a0 = run<32, false>(g, sink, iters), b0 = run<16, false>(g, sink, iters);
        printf("ns per tile, 256 CUs: 32x32x16 %.1f (%.0f TFLOP/s)  16x16x32 %.1f (%.0f TFLOP/s)  [%+.1f %%]   | without the softmax arithmetic: %.1f (%.0f)  %.1f (%.0f)  [%+.1f %%]\n",
               a, flop_tile / a / 1e3, b, flop_tile / b / 1e3, (a / b - 1) * 100, a0, flop_tile / a0 / 1e3, b0, flop_tile / b0 / 1e3, (a0 / b0 - 1) * 100);
    }
    return 0;
}
