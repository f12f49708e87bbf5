// fp32 VERIFICATION kernels: ug_gemm_f32 and ug_flash_attn_fwd_f32.
//
// The product path computes in bf16 (gemm.hip, attention.hip) with the reference's bf16 rounding points, so a forward of a deep
// transformer can only agree with another bf16 evaluation to ~1e-2 (bf16 eps = 7.8e-3 per op). To show that the HOST ORCHESTRATION
// (which weights, which token streams, which order: src/UniGenTransformer.py:1106-1180, 969-1026) matches the reference to the
// north star's 1e-3, every entry point has an fp32 twin with the SAME descriptor / argument meaning, fp32 storage and no
// intermediate rounding; a model whose parameters are fp32 runs the identical host code through these twins and is compared with
// the oracle's fp32 evaluation (tests/test_verify_f32_gpu.py). Correctness-first kernels: plain FMA chains, no MFMA (gfx950 has no
// fp32-input fast path beyond the vector rate anyway), sized for the verification shapes, not for the benchmark.
#include "ug_common.h"
#include <math.h>

namespace {

__device__ __forceinline__ unsigned vmap32(unsigned m, unsigned rpb, unsigned bstride) {
    if (rpb == 0) return m;
    const unsigned b = m / rpb;
    return b * bstride + (m - b * rpb);
}

// C[m][n] = epi( sum_k A[m][k] W[n][k] (+ sum_r T[m][r] B[n][r]) + bias[n] ): 64 x 64 tile, BK = 16, 256 threads x (4 x 4) outputs.
constexpr int VT = 64, VK = 16;

__global__ __launch_bounds__(256) void gemm_f32_kernel(const ug_gemm_desc p) {
    __shared__ float As[VK][VT + 4];
    __shared__ float Ws[VK][VT + 4];
    const int tid = threadIdx.x;
    const int tx = tid & 15, ty = tid >> 4;             // outputs: rows ty*4..+3, cols tx*4..+3
    const int64_t m0 = (int64_t)blockIdx.y * VT, n0 = (int64_t)blockIdx.x * VT;
    const int g = blockIdx.z;
    float acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
    const int lr = tid >> 2, lk = (tid & 3) * 4;         // staging: row lr (0..63), k offset lk (0, 4, 8, 12)
    for (int seg = 0; seg < 2; ++seg) {
        const float* Ab; const float* Wb; int64_t lda, ldw, K; unsigned a_rpb, a_bs;
        if (seg == 0) {
            Ab = (const float*)p.A + (int64_t)g * p.a_gstride; lda = p.lda; a_rpb = (unsigned)p.a_rpb; a_bs = (unsigned)p.a_bstride;
            Wb = (const float*)p.W + (int64_t)g * p.w_gstride; ldw = p.ldw; K = p.K;
        } else {
            if (p.lora_r <= 0) break;
            Ab = (const float*)p.lora_T; lda = p.ldt; a_rpb = 0; a_bs = 0;
            Wb = (const float*)p.lora_B; ldw = p.ldb; K = p.lora_r;
        }
        const int64_t am = m0 + lr, wn = n0 + lr;
        const float* arow = am < p.M ? Ab + (int64_t)vmap32((unsigned)am, a_rpb, a_bs) * lda : nullptr;
        const float* wrow = wn < p.N ? Wb + wn * ldw : nullptr;
        for (int64_t k0 = 0; k0 < K; k0 += VK) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int64_t k = k0 + lk + e;
                As[lk + e][lr] = (arow && k < K) ? arow[k] : 0.f;
                Ws[lk + e][lr] = (wrow && k < K) ? wrow[k] : 0.f;
            }
            __syncthreads();
#pragma unroll
            for (int kk = 0; kk < VK; ++kk) {
                const f32x4 a = *(const f32x4*)&As[kk][ty * 4];
                const f32x4 w = *(const f32x4*)&Ws[kk][tx * 4];
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(a[i], w[j], acc[i][j]);
            }
            __syncthreads();
        }
    }
    const float* bias = p.bias ? (const float*)p.bias + (int64_t)g * p.bias_gstride : nullptr;
    const bool gelu_tile = p.epilogue == UG_EPI_BIAS_GELU && n0 >= p.gelu_from_n;
    const int64_t cshift = (p.c_shift_from_n > 0 && n0 >= p.c_shift_from_n) ? p.c_shift : 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int64_t m = m0 + ty * 4 + i;
        if (m >= p.M) continue;
        float* crow = (float*)p.C + (int64_t)g * p.c_gstride + (int64_t)vmap32((unsigned)m, (unsigned)p.c_rpb, (unsigned)p.c_bstride) * p.ldc + cshift;
        const float* rrow = nullptr; const float* grow = nullptr;
        if (p.epilogue == UG_EPI_RES_GATE || p.epilogue == UG_EPI_RES_SCALE)
            rrow = (const float*)p.R + (int64_t)g * p.r_gstride + (int64_t)vmap32((unsigned)m, (unsigned)p.r_rpb, (unsigned)p.r_bstride) * p.ldr;
        if (p.epilogue == UG_EPI_RES_GATE)
            grow = (const float*)p.gate + (int64_t)g * p.gate_gstride + (int64_t)((unsigned)m / (unsigned)p.rows_per_sample) * p.gate_ld;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int64_t n = n0 + tx * 4 + j;
            if (n >= p.N) continue;
            float v = acc[i][j] + (bias ? bias[n] : 0.f);
            if (gelu_tile) {
                const float u = 0.7978845608028654f * (v + 0.044715f * v * v * v);      // F.gelu(approximate="tanh")
                v = 0.5f * v * (1.0f + tanhf(u));
            } else if (p.epilogue == UG_EPI_RES_GATE) {
                v = rrow[n] + grow[n] * v;
            } else if (p.epilogue == UG_EPI_RES_SCALE) {
                v = rrow[n] + p.alpha * v;
            }
            crow[n] = v;
        }
    }
}

// O = softmax(Q K^T * scale) V in fp32: one thread per query row (64 rows of one (batch, head) per block), keys streamed with
// wave-uniform addresses, exact online softmax (running max updated per key), fp32 accumulators in registers, q in LDS.
template <int DH>
__global__ __launch_bounds__(64) void flash_attn_f32_kernel(
    const float* __restrict__ q, int64_t q_rs, int64_t q_bs, const float* __restrict__ k, int64_t k_rs, int64_t k_bs,
    const float* __restrict__ v, int64_t v_rs, int64_t v_bs, float* __restrict__ o, int64_t o_rs, int64_t o_bs,
    int heads, int Lq, int Lkv, int nQ, float scale) {
    __shared__ __attribute__((aligned(16))) float qs[64][DH + 4];
    const int lane = threadIdx.x;
    const int qt = blockIdx.x % nQ, bh = blockIdx.x / nQ;
    const int head = bh % heads, b = bh / heads;
    const int row = qt * 64 + lane;
    const int rl = row < Lq ? row : Lq - 1;
    const float* qp = q + (int64_t)b * q_bs + (int64_t)rl * q_rs + head * DH;
#pragma unroll
    for (int d = 0; d < DH; d += 4) *(f32x4*)&qs[lane][d] = *(const f32x4*)(qp + d);
    const float* Kb = k + (int64_t)b * k_bs + head * DH;
    const float* Vb = v + (int64_t)b * v_bs + head * DH;
    float acc[DH];
#pragma unroll
    for (int d = 0; d < DH; ++d) acc[d] = 0.f;
    float m = -INFINITY, l = 0.f;
    for (int j = 0; j < Lkv; ++j) {
        const float* kr = Kb + (int64_t)j * k_rs;
        float s = 0.f;
#pragma unroll
        for (int d = 0; d < DH; d += 4) {
            const f32x4 kv = *(const f32x4*)(kr + d);
            const f32x4 qv = *(const f32x4*)&qs[lane][d];
            s = fmaf(qv[0], kv[0], s); s = fmaf(qv[1], kv[1], s); s = fmaf(qv[2], kv[2], s); s = fmaf(qv[3], kv[3], s);
        }
        s *= scale;
        const float mn = fmaxf(m, s);
        const float alpha = expf(m - mn);            // exp(-inf) = 0 on the first key
        const float pj = expf(s - mn);
        l = l * alpha + pj;
        m = mn;
        const float* vr = Vb + (int64_t)j * v_rs;
#pragma unroll
        for (int d = 0; d < DH; d += 4) {
            const f32x4 vv = *(const f32x4*)(vr + d);
            acc[d] = fmaf(pj, vv[0], acc[d] * alpha); acc[d + 1] = fmaf(pj, vv[1], acc[d + 1] * alpha);
            acc[d + 2] = fmaf(pj, vv[2], acc[d + 2] * alpha); acc[d + 3] = fmaf(pj, vv[3], acc[d + 3] * alpha);
        }
    }
    if (row < Lq) {
        const float inv = 1.0f / l;
        float* op = o + (int64_t)b * o_bs + (int64_t)row * o_rs + head * DH;
#pragma unroll
        for (int d = 0; d < DH; d += 4) *(f32x4*)(op + d) = (f32x4){acc[d] * inv, acc[d + 1] * inv, acc[d + 2] * inv, acc[d + 3] * inv};
    }
}

}  // namespace

extern "C" int ug_gemm_f32(const ug_gemm_desc* dp, ug_stream_t stream) {
    UG_REQUIRE(dp != nullptr, UG_ERR_BAD_SHAPE, "ug_gemm_f32: null descriptor");
    ug_gemm_desc d = *dp;
    if (d.groups <= 0) d.groups = 1;
    UG_REQUIRE(d.M >= 0 && d.N > 0 && d.K > 0, UG_ERR_BAD_SHAPE, "ug_gemm_f32: bad M/N/K %lld/%lld/%lld", (long long)d.M, (long long)d.N, (long long)d.K);
    if (d.M == 0) return UG_OK;
    UG_REQUIRE(d.M < (1ll << 31) && d.N < (1ll << 31), UG_ERR_UNSUPPORTED, "ug_gemm_f32: row counts must fit 31 bits");
    UG_REQUIRE(d.A && d.W && d.C, UG_ERR_BAD_SHAPE, "ug_gemm_f32: null operand");
    UG_REQUIRE(d.lda >= d.K && d.ldw >= d.K && d.ldc >= d.N, UG_ERR_BAD_SHAPE, "ug_gemm_f32: leading dims too small");
    UG_REQUIRE(d.epilogue >= UG_EPI_BIAS && d.epilogue <= UG_EPI_F32, UG_ERR_UNSUPPORTED, "ug_gemm_f32: unknown epilogue %d", d.epilogue);
    UG_REQUIRE(d.gelu_from_n >= 0 && d.c_shift_from_n >= 0 && d.gelu_from_n % 64 == 0 && d.c_shift_from_n % 64 == 0 &&
               (d.c_shift_from_n > 0 || d.c_shift == 0), UG_ERR_BAD_SHAPE, "ug_gemm_f32: column split boundaries must be multiples of 64");
    if (d.epilogue == UG_EPI_RES_GATE || d.epilogue == UG_EPI_RES_SCALE) UG_REQUIRE(d.R != nullptr, UG_ERR_BAD_SHAPE, "ug_gemm_f32: residual missing");
    if (d.epilogue == UG_EPI_RES_GATE) UG_REQUIRE(d.gate && d.rows_per_sample > 0, UG_ERR_BAD_SHAPE, "ug_gemm_f32: gate missing");
    if (d.lora_r > 0) UG_REQUIRE(d.lora_T && d.lora_B && d.groups == 1, UG_ERR_BAD_SHAPE, "ug_gemm_f32: LoRA operands missing (or grouped)");
    dim3 grid((unsigned)((d.N + VT - 1) / VT), (unsigned)((d.M + VT - 1) / VT), (unsigned)d.groups);
    hipLaunchKernelGGL(gemm_f32_kernel, grid, dim3(256), 0, (hipStream_t)stream, d);
    UG_CHECK_LAUNCH("ug_gemm_f32");
    return UG_OK;
}

extern "C" int ug_flash_attn_fwd_f32(const void* q, int64_t q_row_stride, int64_t q_batch_stride, const void* k,
                                     int64_t k_row_stride, int64_t k_batch_stride, const void* v, int64_t v_row_stride,
                                     int64_t v_batch_stride, void* o, int64_t o_row_stride, int64_t o_batch_stride,
                                     int64_t batches, int32_t heads, int64_t Lq, int64_t Lkv, int32_t dh, float softmax_scale,
                                     ug_stream_t stream) {
    if (batches == 0 || Lq == 0) return UG_OK;
    UG_REQUIRE(q && k && v && o && batches > 0 && heads > 0 && Lq > 0 && Lkv > 0, UG_ERR_BAD_SHAPE, "ug_flash_attn_fwd_f32: bad arguments");
    UG_REQUIRE(dh == 128 || dh == 64, UG_ERR_UNSUPPORTED, "ug_flash_attn_fwd_f32: head dim %d not in {64, 128}", dh);
    UG_REQUIRE(Lq < (1 << 30) && Lkv < (1 << 30), UG_ERR_UNSUPPORTED, "ug_flash_attn_fwd_f32: sequence too long");
    UG_REQUIRE(q_row_stride % 4 == 0 && k_row_stride % 4 == 0 && v_row_stride % 4 == 0 && o_row_stride % 4 == 0 &&
               q_batch_stride % 4 == 0 && k_batch_stride % 4 == 0 && v_batch_stride % 4 == 0 && o_batch_stride % 4 == 0 &&
               ug_aligned(q, 16) && ug_aligned(k, 16) && ug_aligned(v, 16) && ug_aligned(o, 16),
               UG_ERR_BAD_ALIGN, "ug_flash_attn_fwd_f32: strides must be multiples of 4 elements and bases 16-byte aligned");
    const int nQ = (int)((Lq + 63) / 64);
    const int64_t nwg = (int64_t)nQ * heads * batches;
    UG_REQUIRE(nwg < (1ll << 31), UG_ERR_UNSUPPORTED, "ug_flash_attn_fwd_f32: grid too large");
    if (dh == 128)
        hipLaunchKernelGGL(flash_attn_f32_kernel<128>, dim3((unsigned)nwg), dim3(64), 0, (hipStream_t)stream, (const float*)q, q_row_stride, q_batch_stride,
                           (const float*)k, k_row_stride, k_batch_stride, (const float*)v, v_row_stride, v_batch_stride, (float*)o, o_row_stride,
                           o_batch_stride, (int)heads, (int)Lq, (int)Lkv, nQ, softmax_scale);
    else
        hipLaunchKernelGGL(flash_attn_f32_kernel<64>, dim3((unsigned)nwg), dim3(64), 0, (hipStream_t)stream, (const float*)q, q_row_stride, q_batch_stride,
                           (const float*)k, k_row_stride, k_batch_stride, (const float*)v, v_row_stride, v_batch_stride, (float*)o, o_row_stride,
                           o_batch_stride, (int)heads, (int)Lq, (int)Lkv, nQ, softmax_scale);
    UG_CHECK_LAUNCH("ug_flash_attn_fwd_f32");
    return UG_OK;
}
