// HBM-bound row kernels of the UniGen forward: AdaLayerNorm-Zero modulate, q/k RMSNorm + RoPE, per-sample
// (small-M) linears, sinusoidal timestep features, Euler step, add. All bf16 in/out, fp32 math, 16-byte accesses.
// Rounding points mirror the reference's separate bf16 torch ops so results are comparable element-for-element.
// Every kernel is a template over the element type (ElemT, ug_common.h): T = bf16_t is the product path, T = float the
// fp32 VERIFICATION path (`ug_*_f32`: same code, fp32 storage, no intermediate rounding).
#include "ug_common.h"
#include <algorithm>

namespace {

__device__ __forceinline__ void unpack8(const u32x4 v, float* f) {
    f[0] = bflo(v.x); f[1] = bfhi(v.x); f[2] = bflo(v.y); f[3] = bfhi(v.y);
    f[4] = bflo(v.z); f[5] = bfhi(v.z); f[6] = bflo(v.w); f[7] = bfhi(v.w);
}
__device__ __forceinline__ u32x4 pack8(const float* f) {
    u32x4 v;
    v.x = pack2bf(f[0], f[1]); v.y = pack2bf(f[2], f[3]); v.z = pack2bf(f[4], f[5]); v.w = pack2bf(f[6], f[7]);
    return v;
}

// ---------------------------------------------------------------------------------------------------------
// AdaLayerNorm modulate: out = LN(x) * (1 + scale[b]) + shift[b].  One wave per row, row kept in registers.
// reference: diffusers AdaLayerNormZero.forward / FluxTransformerBlock norm2 + modulate; src/UniGenUtils.py:354-373
// ---------------------------------------------------------------------------------------------------------
constexpr int LN_MAXCH = 8;  // chunks (of 8 elements) per lane -> D <= 4096

template <typename T>
__global__ __launch_bounds__(256) void adaln_modulate_kernel(
    const T* __restrict__ x, int64_t ldx, int64_t x_rpb, int64_t x_bstride,
    const T* __restrict__ shift, const T* __restrict__ scale, int64_t mod_ld, int64_t rows_per_sample,
    T* __restrict__ out, int64_t ldo, int64_t rows, int D, float eps) {
    using E = ElemT<T>;
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int nchunk = D >> 3;
    const T* xr = x + ug_rowmap(row, x_rpb, x_bstride) * ldx;
    float v[LN_MAXCH][8];
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < LN_MAXCH; ++t) {
        const int c = lane + t * 64;
        if (c < nchunk) {
            E::load8(xr + c * 8, v[t]);
#pragma unroll
            for (int e = 0; e < 8; ++e) s += v[t][e];
        }
    }
    const float mean = wave_sum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int t = 0; t < LN_MAXCH; ++t) {
        const int c = lane + t * 64;
        if (c < nchunk) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float d = v[t][e] - mean; q += d * d; }
        }
    }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)D + eps);
    const int64_t b = row / rows_per_sample;
    const T* sh = shift + b * mod_ld;
    const T* sc = scale + b * mod_ld;
    T* orow = out + row * ldo;
#pragma unroll
    for (int t = 0; t < LN_MAXCH; ++t) {
        const int c = lane + t * 64;
        if (c < nchunk) {
            float fs[8], fc[8], o[8];
            E::load8(sh + c * 8, fs);
            E::load8(sc + c * 8, fc);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float n = E::rnd((v[t][e] - mean) * rstd);   // LayerNorm output (bf16 tensor in the reference)
                const float s1 = E::rnd(1.0f + fc[e]);             // (1 + scale)
                o[e] = E::rnd(n * s1) + fs[e];                     // * then +, each a bf16 op in the reference
            }
            E::store8(orow + c * 8, o);
        }
    }
}

// Fast path of the same op for the model widths (bf16, D = NCH * 512 exactly: 3072 FLUX, 1536 SD3.5, 4096): round 3.
// The generic kernel above puts every `if (c < nchunk)` load in its own basic block, and hipcc waits vmcnt(0) behind each one: six
// dependent HBM round trips per row (~4 TB/s, and only by occupancy). Here a wave issues its whole row (NCH x 16 B per lane) and the
// sample's shift / scale chunks back to back with no branch in between, keeps the row PACKED (NCH x 4 registers instead of NCH x 8 floats),
// and reduces with DPP row operations + 4 v_readlane instead of 12 dependent ds_bpermute (each an LDS crossbar round trip with its own
// lgkmcnt(0)). Same formula and rounding points; only the association of the two fp32 row sums differs from the generic kernel.
#define UG_DPP_ADD(V, CTRL) ((V) + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, (V)), (CTRL), 0xf, 0xf, true)))
__device__ __forceinline__ float wave_sum_dpp(float v) {
    v = UG_DPP_ADD(v, 0xB1);      // quad_perm [1,0,3,2]: lane ^ 1
    v = UG_DPP_ADD(v, 0x4E);      // quad_perm [2,3,0,1]: lane ^ 2
    v = UG_DPP_ADD(v, 0x141);     // row_half_mirror: the other quad of the 8-lane half row
    v = UG_DPP_ADD(v, 0x140);     // row_mirror: the other half of the 16-lane row -> every lane holds its row's sum
    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 0));
    const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 16));
    const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 32));
    const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 48));
    return (r0 + r1) + (r2 + r3);
}

template <int NCH>
__global__ __launch_bounds__(256) void adaln_modulate_fast_kernel(
    const bf16_t* __restrict__ x, int64_t ldx, unsigned x_rpb, unsigned x_bstride,
    const bf16_t* __restrict__ shift, const bf16_t* __restrict__ scale, int64_t mod_ld, unsigned rows_per_sample,
    bf16_t* __restrict__ out, int64_t ldo, unsigned rows, float eps) {
    constexpr int D = NCH * 512;
    const int lane = threadIdx.x & 63;
    const unsigned row = blockIdx.x * 4u + (unsigned)__builtin_amdgcn_readfirstlane(threadIdx.x >> 6);     // scalar: the row maps below are SALU
    if (row >= rows) return;
    unsigned prow = row;
    if (x_rpb) { const unsigned b = row / x_rpb; prow = b * x_bstride + (row - b * x_rpb); }
    const bf16_t* xr = x + (int64_t)prow * ldx + lane * 8;
    u32x4 xv[NCH];
#pragma unroll
    for (int t = 0; t < NCH; ++t) xv[t] = __builtin_nontemporal_load((const u32x4*)(xr + t * 512));      // read once: do not displace the GEMM operands in L2
    const unsigned b = row / rows_per_sample;
    const bf16_t* sh = shift + (int64_t)b * mod_ld + lane * 8;
    const bf16_t* sc = scale + (int64_t)b * mod_ld + lane * 8;
    u32x4 shv[NCH], scv[NCH];
#pragma unroll
    for (int t = 0; t < NCH; ++t) { shv[t] = *(const u32x4*)(sh + t * 512); scv[t] = *(const u32x4*)(sc + t * 512); }
    __builtin_amdgcn_sched_barrier(0);                     // every load is in flight before the first use
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < NCH; ++t) {
        float f[8]; unpack8(xv[t], f);
#pragma unroll
        for (int e = 0; e < 8; ++e) s += f[e];
    }
    const float mean = wave_sum_dpp(s) / (float)D;
    // the row stays packed between the passes: an opaque use stops hipcc from keeping all NCH x 8 unpacked floats alive (133 registers, 3 waves
    // per SIMD) - the unpack is two VALU ops per pair and this kernel waits on memory
#pragma unroll
    for (int t = 0; t < NCH; ++t) asm volatile("" : "+v"(xv[t]));
    float q = 0.f;
#pragma unroll
    for (int t = 0; t < NCH; ++t) {
        float f[8]; unpack8(xv[t], f);
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float d = f[e] - mean; q += d * d; }
    }
    const float rstd = 1.0f / sqrtf(wave_sum_dpp(q) / (float)D + eps);
#pragma unroll
    for (int t = 0; t < NCH; ++t) asm volatile("" : "+v"(xv[t]), "+v"(shv[t]), "+v"(scv[t]));
    bf16_t* orow = out + (int64_t)row * ldo + lane * 8;
#pragma unroll
    for (int t = 0; t < NCH; ++t) {
        float f[8], fs[8], fc[8], o[8];
        unpack8(xv[t], f); unpack8(shv[t], fs); unpack8(scv[t], fc);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float n = rbf((f[e] - mean) * rstd);     // LayerNorm output (bf16 tensor in the reference)
            const float s1 = rbf(1.0f + fc[e]);            // (1 + scale)
            o[e] = rbf(n * s1) + fs[e];                    // * then +, each a bf16 op in the reference (the + is rounded by the pack)
        }
        *(u32x4*)(orow + t * 512) = pack8(o);
    }
}

// ---------------------------------------------------------------------------------------------------------
// q/k RMSNorm + RoPE, in place on the fused projection buffer. DH/8 lanes per head vector.
// reference: diffusers RMSNorm (Attention.norm_q/k, norm_added_q/k) + apply_rotary_emb; src/UniGenUtils.py:561-599
// ---------------------------------------------------------------------------------------------------------
template <typename T, int DH>
__global__ __launch_bounds__(256) void qk_rmsnorm_rope_kernel(
    T* __restrict__ buf, int64_t ld, int64_t total_rows, int64_t rows_per_batch, int64_t batch_stride_rows,
    int64_t pos_offset, int64_t q_off, int64_t k_off, int heads, const T* __restrict__ wq_a, const T* __restrict__ wk_a, const T* __restrict__ wq_b,
    const T* __restrict__ wk_b, int64_t split, const float* __restrict__ cos_tab,
    const float* __restrict__ sin_tab, float eps) {
    using E = ElemT<T>;
    constexpr int LPV = DH / 8;                 // lanes per head vector
    constexpr int VPW = 64 / LPV;               // vectors per wave
    const int lane = threadIdx.x & 63;
    const int sub = lane % LPV, vin = lane / LPV;
    const int nwhich = q_off >= 0 ? 2 : 1;
    const int64_t nvec = total_rows * nwhich * heads;
    const int64_t wave_id = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t vec = wave_id * VPW + vin;
    const bool active = vec < nvec;
    const unsigned vv = active ? (unsigned)vec : 0u;      // 32-bit index math (host checks the ranges): 64-bit divisions are slow
    const unsigned t1 = vv / (unsigned)heads;
    const int h = (int)(vv - t1 * (unsigned)heads);
    const int which = q_off >= 0 ? (int)(t1 & 1u) : 1;   // 0 = q, 1 = k
    const unsigned row = q_off >= 0 ? (t1 >> 1) : t1;
    const unsigned bidx_u = row / (unsigned)rows_per_batch;
    const int64_t bidx = bidx_u;
    const int64_t rr = row - bidx_u * (unsigned)rows_per_batch;
    const int64_t pos = pos_offset + rr;
    T* p = buf + (bidx * batch_stride_rows + rr) * ld + (which == 0 ? q_off : k_off) + (int64_t)h * DH + sub * 8;
    float x[8];
    E::load8(p, x);
    const T* w = (pos < split) ? (which == 0 ? wq_a : wk_a) : (which == 0 ? wq_b : wk_b);
    if (w) {
        float ss = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) ss += x[e] * x[e];
#pragma unroll
        for (int o = LPV / 2; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
        const float rs = rsqrtf(ss / (float)DH + eps);
        float wf[8];
        E::load8(w + sub * 8, wf);
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = E::rnd(E::rnd(x[e] * rs) * wf[e]);
    }
    if (cos_tab) {
        const float* cp = cos_tab + pos * DH + sub * 8;
        const float* sp = sin_tab + pos * DH + sub * 8;
        const f32x4 c0 = *(const f32x4*)cp, c1 = *(const f32x4*)(cp + 4);
        const f32x4 s0 = *(const f32x4*)sp, s1 = *(const f32x4*)(sp + 4);
        const float c[8] = {c0[0], c0[1], c0[2], c0[3], c1[0], c1[1], c1[2], c1[3]};
        const float s[8] = {s0[0], s0[1], s0[2], s0[3], s1[0], s1[1], s1[2], s1[3]};
        float o[8];
#pragma unroll
        for (int e = 0; e < 8; e += 2) {
            o[e] = x[e] * c[e] + (-x[e + 1]) * s[e];
            o[e + 1] = x[e + 1] * c[e + 1] + x[e] * s[e + 1];
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = o[e];
    }
    if (active) E::store8(p, x);
}

// ---------------------------------------------------------------------------------------------------------
// Per-sample linears (M <= 16): out = R + bf16(act(x) W^T + b). x is staged once per block in LDS (with SiLU
// applied), each wave streams whole weight rows from HBM with 16-byte loads. HBM-bound on W.
// reference: AdaLayerNormZero.linear(silu(emb)), TimestepEmbedding, PixArtAlphaTextProjection (diffusers 0.32.2)
// ---------------------------------------------------------------------------------------------------------
constexpr int SL_COLS_PER_WAVE = 8;

template <int MT>
__global__ __launch_bounds__(256) void small_linear_kernel(
    const bf16_t* __restrict__ x, int64_t ldx, const bf16_t* __restrict__ W, int64_t ldw,
    const bf16_t* __restrict__ bias, const bf16_t* __restrict__ R, int64_t ldr, bf16_t* __restrict__ out,
    int64_t ldo, int M, int64_t N, int K, int act_in) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    bf16_t* xs = (bf16_t*)smem;  // [MT][K]
    const int nchunk = K >> 3;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t nbase = ((int64_t)blockIdx.x * 4 + wave) * SL_COLS_PER_WAVE;
    // The weight stream is the whole cost (HBM-bound, M <= 16 rows of x against N x K weights): every wave owns 8 weight rows and has
    // all 8 of their 16-byte chunks in flight per step; the first step's loads are issued before x is staged so that the staging
    // (SiLU: exp2 + rcp, no IEEE division) runs under their latency. (4 columns per wave, two rows per pass and a division-based
    // SiLU recomputed by every block streamed at 1.9 TB/s.)
    const bf16_t* wr[SL_COLS_PER_WAVE];
#pragma unroll
    for (int j = 0; j < SL_COLS_PER_WAVE; ++j) {
        int64_t n = nbase + j; if (n > N - 1) n = N - 1;                // rows past N re-read the last row; their sums are dropped
        wr[j] = W + n * ldw;
    }
    u32x4 wq[SL_COLS_PER_WAVE];
    if (lane < nchunk) {
#pragma unroll
        for (int j = 0; j < SL_COLS_PER_WAVE; ++j) wq[j] = __builtin_nontemporal_load((const u32x4*)(wr[j] + lane * 8));
    }
    for (int i = threadIdx.x; i < MT * nchunk; i += 256) {
        const int m = i / nchunk, c = i - m * nchunk;
        float f[8];
        if (m < M) {
            unpack8(*(const u32x4*)(x + (int64_t)m * ldx + c * 8), f);
            if (act_in == 1) {
#pragma unroll
                for (int e = 0; e < 8; ++e) f[e] = f[e] * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * f[e]));   // SiLU, rounded to bf16 below
            }
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] = 0.f;
        }
        *(u32x4*)(xs + (int64_t)m * K + c * 8) = pack8(f);
    }
    __syncthreads();
    if (nbase >= N) return;
    float acc[SL_COLS_PER_WAVE][MT];
#pragma unroll
    for (int j = 0; j < SL_COLS_PER_WAVE; ++j)
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[j][m] = 0.f;
    for (int c = lane; c < nchunk; c += 64) {            // one step ahead (a deeper register ring measured slower: hipcc drains it with vmcnt(0))
        u32x4 wn[SL_COLS_PER_WAVE];
        const bool more = c + 64 < nchunk;
        if (more) {
#pragma unroll
            for (int j = 0; j < SL_COLS_PER_WAVE; ++j) wn[j] = __builtin_nontemporal_load((const u32x4*)(wr[j] + (c + 64) * 8));
        }
        float xf[MT][8];
#pragma unroll
        for (int m = 0; m < MT; ++m) unpack8(*(const u32x4*)(xs + (int64_t)m * K + c * 8), xf[m]);
#pragma unroll
        for (int j = 0; j < SL_COLS_PER_WAVE; ++j) {
            float w[8];
            unpack8(wq[j], w);
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[j][m] += xf[m][e] * w[e];
        }
        if (more) {
#pragma unroll
            for (int j = 0; j < SL_COLS_PER_WAVE; ++j) wq[j] = wn[j];
        }
    }
#pragma unroll
    for (int j = 0; j < SL_COLS_PER_WAVE; ++j)
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[j][m] = wave_sum(acc[j][m]);
    // lane l writes out[l / 8 (+ 8)][nbase + l % 8]
    const int jo = lane & 7;
    const int64_t n = nbase + jo;
    for (int mo = lane >> 3; mo < M && n < N; mo += 8) {
        float v = 0.f;
#pragma unroll
        for (int j = 0; j < SL_COLS_PER_WAVE; ++j)
#pragma unroll
            for (int m = 0; m < MT; ++m) if (m == mo && j == jo) v = acc[j][m];
        if (bias) v += bf2f(bias[n]);
        v = rbf(v);
        if (R) v += bf2f(R[(int64_t)mo * ldr + n]);
        out[(int64_t)mo * ldo + n] = f2bf(v);
    }
}

// fp32 verification twin of small_linear_kernel: one wave per output column, fp32 x / W / bias / R / out, no rounding.
__global__ __launch_bounds__(256) void small_linear_f32_kernel(
    const float* __restrict__ x, int64_t ldx, const float* __restrict__ W, int64_t ldw, const float* __restrict__ bias,
    const float* __restrict__ R, int64_t ldr, float* __restrict__ out, int64_t ldo, int M, int64_t N, int K, int act_in) {
    const int lane = threadIdx.x & 63;
    const int64_t n = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    const float* w = W + n * ldw;
    for (int m = 0; m < M; ++m) {
        float acc = 0.f;
        for (int k = lane; k < K; k += 64) {
            float xv = x[(int64_t)m * ldx + k];
            if (act_in == 1) xv = xv / (1.0f + expf(-xv));
            acc += xv * w[k];
        }
        acc = wave_sum(acc);
        if (lane == 0) {
            float v = acc + (bias ? bias[n] : 0.f);
            if (R) v += R[(int64_t)m * ldr + n];
            out[(int64_t)m * ldo + n] = v;
        }
    }
}

template <typename T>
__global__ void timestep_embed_kernel(const float* __restrict__ t, T* __restrict__ out, int64_t ldo, int B, int dim) {
    const int half = dim >> 1;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * half) return;
    const int b = i / half, k = i - b * half;
    const float f = expf(-9.210340371976184f * (float)k / (float)half);  // exp(-ln(10000) k / half)
    const float a = t[b] * f;
    ElemT<T>::st(out + (int64_t)b * ldo + k, cosf(a));          // flip_sin_to_cos=True -> [cos | sin]
    ElemT<T>::st(out + (int64_t)b * ldo + half + k, sinf(a));
}

// FlowMatchEulerDiscreteScheduler.step as torch evaluates it (diffusers 0.32.2): `sample.float() + (sigma_next - sigma) * model_output`, where the
// step is a 0-dim fp32 tensor (an element difference of the scheduler's fp32 sigmas, on the model's device) and model_output a bf16 tensor - torch's type
// promotion gives the PRODUCT the dimensioned operand's dtype and casts BOTH operands to it: the step is rounded to bf16, the product is rounded to bf16, then
// the fp32 add, then the cast back. (With FLUX-schnell's four steps dt = -0.25 and everything is exact; SD3's 28 shifted steps round.)
template <typename T>
__global__ void euler_step_kernel(T* __restrict__ x, const T* __restrict__ v, float dt, int64_t nchunk) {
    for (int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; c < nchunk; c += (int64_t)gridDim.x * blockDim.x) {
        float a[8], b[8];
        const float dtr = ElemT<T>::rnd(dt);
        ElemT<T>::load8(x + c * 8, a);
        ElemT<T>::load8(v + c * 8, b);
#pragma unroll
        for (int e = 0; e < 8; ++e) a[e] = a[e] + ElemT<T>::rnd(dtr * b[e]);
        ElemT<T>::store8(x + c * 8, a);
    }
}

template <typename T>
__global__ void add_kernel(const T* __restrict__ a, int64_t lda, const T* __restrict__ b, int64_t ldb,
                           T* __restrict__ out, int64_t ldo, int64_t rows, int chunks_per_row) {
    const int64_t total = rows * chunks_per_row;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / chunks_per_row; const int c = (int)(i - r * chunks_per_row);
        float fa[8], fb[8];
        ElemT<T>::load8(a + r * lda + c * 8, fa);
        ElemT<T>::load8(b + r * ldb + c * 8, fb);
#pragma unroll
        for (int e = 0; e < 8; ++e) fa[e] += fb[e];
        ElemT<T>::store8(out + r * ldo + c * 8, fa);
    }
}

template <typename T>
__global__ void cfg_combine_kernel(const T* __restrict__ u, const T* __restrict__ t, float gs, T* __restrict__ out, int64_t nchunk) {
    using E = ElemT<T>;
    for (int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; c < nchunk; c += (int64_t)gridDim.x * blockDim.x) {
        float a[8], b[8];
        E::load8(u + c * 8, a);
        E::load8(t + c * 8, b);
#pragma unroll
        for (int e = 0; e < 8; ++e) a[e] = a[e] + E::rnd(gs * E::rnd(b[e] - a[e]));
        E::store8(out + c * 8, a);
    }
}

template <typename T>
__global__ void add_rowbcast_f32_kernel(T* __restrict__ x, int64_t ldx, const float* __restrict__ t, int64_t ldt, int64_t rows,
                                        int64_t rpb, int chunks_per_row) {
    const int64_t total = rows * chunks_per_row;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / chunks_per_row; const int c = (int)(i - r * chunks_per_row);
        float f[8];
        ElemT<T>::load8(x + r * ldx + c * 8, f);
        const float* tp = t + (r % rpb) * ldt + c * 8;
        const f32x4 a = *(const f32x4*)tp, b = *(const f32x4*)(tp + 4);
        f[0] += a[0]; f[1] += a[1]; f[2] += a[2]; f[3] += a[3]; f[4] += b[0]; f[5] += b[1]; f[6] += b[2]; f[7] += b[3];
        ElemT<T>::store8(x + r * ldx + c * 8, f);
    }
}

template <typename T>
__global__ void gather_rows_kernel(const T* __restrict__ src, int64_t ld_src, const int32_t* __restrict__ idx, T* __restrict__ out,
                                   int64_t ld_out, int64_t n, int chunks_per_row) {
    const int64_t total = n * chunks_per_row;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / chunks_per_row; const int c = (int)(i - r * chunks_per_row);
        const int s = idx[r];
        if (s >= 0) {
            float f[8];
            ElemT<T>::load8(src + (int64_t)s * ld_src + c * 8, f);       // exact: a bf16 -> f32 -> bf16 round trip is the identity
            ElemT<T>::store8(out + r * ld_out + c * 8, f);
        } else {
            ElemT<T>::zero8(out + r * ld_out + c * 8);
        }
    }
}

// FluxPipeline._pack_latents / _unpack_latents (src/UniGenPipeline.py:641,796): [B, C, H, W] <-> [B, (H/2)(W/2), 4C] with
// packed column c*4 + dy*2 + dx = pixel (2i + dy, 2j + dx) of channel c. One thread per packed element; pure data movement.
template <typename T, bool PACK>
__global__ void pack_latents_kernel(const T* __restrict__ src, T* __restrict__ dst, int B, int C, int H, int W) {
    const int64_t total = (int64_t)B * C * H * W;
    const int h2 = H >> 1, w2 = W >> 1;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        // i enumerates the PACKED tensor [b][tok][c][dy][dx]
        int64_t r = i;
        const int dx = (int)(r & 1); r >>= 1;
        const int dy = (int)(r & 1); r >>= 1;
        const int c = (int)(r % C); r /= C;
        const int tok = (int)(r % ((int64_t)h2 * w2));
        const int b = (int)(r / ((int64_t)h2 * w2));
        const int ti = tok / w2, tj = tok - ti * w2;
        const int64_t img = (((int64_t)b * C + c) * H + (2 * ti + dy)) * W + (2 * tj + dx);
        if (PACK) dst[i] = src[img]; else dst[img] = src[i];
    }
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------------------
// host side: one implementation per op, instantiated for bf16 (product) and fp32 (verification)
// ---------------------------------------------------------------------------------------------------------------------
namespace {

template <typename T>
int cfg_combine_impl(const void* uncond, const void* text, float guidance_scale, void* out, int64_t n, ug_stream_t stream) {
    if (n == 0) return UG_OK;
    UG_REQUIRE(uncond && text && out && n > 0 && n % 8 == 0 && ug_aligned(uncond, 16) && ug_aligned(text, 16) && ug_aligned(out, 16),
               UG_ERR_BAD_ALIGN, "ug_cfg_combine: n must be a multiple of 8 and pointers 16-byte aligned");
    const int64_t nchunk = n / 8;
    const unsigned grid = (unsigned)std::min<int64_t>((nchunk + 255) / 256, 2048);
    hipLaunchKernelGGL(cfg_combine_kernel<T>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const T*)uncond, (const T*)text,
                       guidance_scale, (T*)out, nchunk);
    UG_CHECK_LAUNCH("ug_cfg_combine");
    return UG_OK;
}

template <typename T>
int add_rowbcast_impl(void* x, int64_t ldx, const float* table, int64_t ldt, int64_t rows, int64_t rows_per_batch, int64_t D, ug_stream_t stream) {
    if (rows == 0 || D == 0) return UG_OK;
    UG_REQUIRE(x && table && rows > 0 && rows_per_batch > 0 && D > 0, UG_ERR_BAD_SHAPE, "ug_add_rowbcast_f32: bad arguments");
    UG_REQUIRE(D % 8 == 0 && ldx % 8 == 0 && ldt % 4 == 0 && ug_aligned(x, 16) && ug_aligned(table, 16), UG_ERR_BAD_ALIGN,
               "ug_add_rowbcast_f32: 16-byte alignment required");
    const int64_t total = rows * (D / 8);
    const unsigned grid = (unsigned)std::min<int64_t>((total + 255) / 256, 2048);
    hipLaunchKernelGGL(add_rowbcast_f32_kernel<T>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (T*)x, ldx, table, ldt, rows, rows_per_batch, (int)(D / 8));
    UG_CHECK_LAUNCH("ug_add_rowbcast_f32");
    return UG_OK;
}

template <typename T>
int gather_rows_impl(const void* src, int64_t ld_src, const int32_t* idx, void* out, int64_t ld_out, int64_t n, int64_t W, ug_stream_t stream) {
    if (n == 0 || W == 0) return UG_OK;
    UG_REQUIRE(src && idx && out && n > 0 && W > 0, UG_ERR_BAD_SHAPE, "ug_gather_rows: bad arguments");
    UG_REQUIRE(W % 8 == 0 && ld_src % 8 == 0 && ld_out % 8 == 0 && ug_aligned(src, 16) && ug_aligned(out, 16), UG_ERR_BAD_ALIGN,
               "ug_gather_rows: 16-byte alignment required");
    const int64_t total = n * (W / 8);
    const unsigned grid = (unsigned)std::min<int64_t>((total + 255) / 256, 4096);
    hipLaunchKernelGGL(gather_rows_kernel<T>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const T*)src, ld_src, idx, (T*)out, ld_out, n, (int)(W / 8));
    UG_CHECK_LAUNCH("ug_gather_rows");
    return UG_OK;
}

template <typename T>
int adaln_modulate_impl(const void* x, int64_t ldx, int64_t x_rpb, int64_t x_bstride, const void* shift, const void* scale, int64_t mod_ld,
                        int64_t rows_per_sample, void* out, int64_t ldo, int64_t rows, int64_t D, float eps, ug_stream_t stream) {
    if (rows == 0) return UG_OK;
    UG_REQUIRE(x && shift && scale && out && rows > 0 && rows_per_sample > 0, UG_ERR_BAD_SHAPE, "ug_adaln_modulate: bad arguments");
    UG_REQUIRE(D % 8 == 0 && D > 0 && D <= LN_MAXCH * 64 * 8, UG_ERR_UNSUPPORTED, "ug_adaln_modulate: D=%lld must be a multiple of 8 and <= 4096", (long long)D);
    UG_REQUIRE(ldx % 8 == 0 && ldo % 8 == 0 && mod_ld % 8 == 0 && ug_aligned(x, 16) && ug_aligned(out, 16) &&
               ug_aligned(shift, 16) && ug_aligned(scale, 16), UG_ERR_BAD_ALIGN, "ug_adaln_modulate: 16-byte alignment required");
    const unsigned grid = (unsigned)((rows + 3) / 4);
    if constexpr (!ElemT<T>::kF32) {
        // model widths: the branch-free kernel (UG_ADALN_FAST=0 keeps the generic one, for A/B and as the reference of the bit-equality test)
        const bool fits32 = rows < (1ll << 31) && x_rpb < (1ll << 31) && x_bstride < (1ll << 31) && rows_per_sample < (1ll << 31) && x_rpb >= 0;
        if (fits32 && (D == 1536 || D == 3072 || D == 4096) && ug_env_int("UG_ADALN_FAST", 1)) {
#define UG_ADALN_GO(NCH) hipLaunchKernelGGL(adaln_modulate_fast_kernel<NCH>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, ldx, (unsigned)x_rpb, \
            (unsigned)x_bstride, (const bf16_t*)shift, (const bf16_t*)scale, mod_ld, (unsigned)rows_per_sample, (bf16_t*)out, ldo, (unsigned)rows, eps)
            if (D == 1536) UG_ADALN_GO(3); else if (D == 3072) UG_ADALN_GO(6); else UG_ADALN_GO(8);
#undef UG_ADALN_GO
            UG_CHECK_LAUNCH("ug_adaln_modulate");
            return UG_OK;
        }
    }
    hipLaunchKernelGGL(adaln_modulate_kernel<T>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const T*)x, ldx, x_rpb,
                       x_bstride, (const T*)shift, (const T*)scale, mod_ld, rows_per_sample, (T*)out, ldo, rows, (int)D, eps);
    UG_CHECK_LAUNCH("ug_adaln_modulate");
    return UG_OK;
}

template <typename T>
int qk_rmsnorm_rope_impl(void* buf, int64_t ld, int64_t batches, int64_t rows_per_batch, int64_t batch_stride_rows, int64_t pos_offset,
                         int64_t q_off, int64_t k_off, int32_t heads, int32_t dh, const void* wq_a, const void* wk_a, const void* wq_b,
                         const void* wk_b, int64_t split, const float* cos_tab, const float* sin_tab, float eps, ug_stream_t stream) {
    const int64_t total_rows = batches * rows_per_batch;
    if (total_rows == 0) return UG_OK;
    UG_REQUIRE(buf && heads > 0 && k_off >= 0 && batch_stride_rows >= rows_per_batch && pos_offset >= 0, UG_ERR_BAD_SHAPE,
               "ug_qk_rmsnorm_rope: bad arguments");
    UG_REQUIRE(dh == 64 || dh == 128, UG_ERR_UNSUPPORTED, "ug_qk_rmsnorm_rope: head dim %d not in {64,128}", dh);
    UG_REQUIRE(ld % 8 == 0 && k_off % 8 == 0 && (q_off < 0 || q_off % 8 == 0) && ug_aligned(buf, 16), UG_ERR_BAD_ALIGN,
               "ug_qk_rmsnorm_rope: 16-byte alignment required");
    UG_REQUIRE((cos_tab == nullptr) == (sin_tab == nullptr), UG_ERR_BAD_SHAPE, "ug_qk_rmsnorm_rope: cos/sin must both be given");
    UG_REQUIRE(!cos_tab || (ug_aligned(cos_tab, 16) && ug_aligned(sin_tab, 16)), UG_ERR_BAD_ALIGN, "ug_qk_rmsnorm_rope: tables misaligned");
    const bool haveq = q_off >= 0;
    if (haveq) UG_REQUIRE((wq_a == nullptr) == (wk_a == nullptr) && (wq_b == nullptr) == (wk_b == nullptr), UG_ERR_BAD_SHAPE,
                          "ug_qk_rmsnorm_rope: q/k norm weights must come in pairs");
    const int64_t nvec = total_rows * (haveq ? 2 : 1) * heads;
    UG_REQUIRE(nvec < (1ll << 31) && rows_per_batch < (1ll << 31), UG_ERR_UNSUPPORTED, "ug_qk_rmsnorm_rope: too many head vectors");
    const int vpw = 64 / (dh / 8);
    const int64_t waves = (nvec + vpw - 1) / vpw;
    const unsigned grid = (unsigned)((waves + 3) / 4);
    hipStream_t s = (hipStream_t)stream;
    if (dh == 128)
        hipLaunchKernelGGL((qk_rmsnorm_rope_kernel<T, 128>), dim3(grid), dim3(256), 0, s, (T*)buf, ld, total_rows, rows_per_batch,
                           batch_stride_rows, pos_offset, q_off, k_off, heads, (const T*)wq_a, (const T*)wk_a, (const T*)wq_b,
                           (const T*)wk_b, split, cos_tab, sin_tab, eps);
    else
        hipLaunchKernelGGL((qk_rmsnorm_rope_kernel<T, 64>), dim3(grid), dim3(256), 0, s, (T*)buf, ld, total_rows, rows_per_batch,
                           batch_stride_rows, pos_offset, q_off, k_off, heads, (const T*)wq_a, (const T*)wk_a, (const T*)wq_b,
                           (const T*)wk_b, split, cos_tab, sin_tab, eps);
    UG_CHECK_LAUNCH("ug_qk_rmsnorm_rope");
    return UG_OK;
}

template <typename T>
int timestep_embed_impl(const float* t, void* out, int64_t ldo, int64_t B, int32_t dim, ug_stream_t stream) {
    if (B == 0) return UG_OK;
    UG_REQUIRE(t && out && B > 0 && dim > 0 && dim % 2 == 0 && ldo >= dim, UG_ERR_BAD_SHAPE, "ug_timestep_embed: bad arguments");
    const int total = (int)B * (dim / 2);
    hipLaunchKernelGGL(timestep_embed_kernel<T>, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, t, (T*)out, ldo, (int)B, dim);
    UG_CHECK_LAUNCH("ug_timestep_embed");
    return UG_OK;
}

template <typename T>
int euler_step_impl(void* x, const void* v, float dt, int64_t n, ug_stream_t stream) {
    if (n == 0) return UG_OK;
    UG_REQUIRE(x && v && n > 0 && n % 8 == 0 && ug_aligned(x, 16) && ug_aligned(v, 16), UG_ERR_BAD_ALIGN,
               "ug_euler_step: n must be a multiple of 8 and pointers 16-byte aligned");
    const int64_t nchunk = n / 8;
    const unsigned grid = (unsigned)std::min<int64_t>((nchunk + 255) / 256, 2048);
    hipLaunchKernelGGL(euler_step_kernel<T>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (T*)x, (const T*)v, dt, nchunk);
    UG_CHECK_LAUNCH("ug_euler_step");
    return UG_OK;
}

template <typename T>
int add_impl(const void* a, int64_t lda, const void* b, int64_t ldb, void* out, int64_t ldo, int64_t rows, int64_t D, ug_stream_t stream) {
    if (rows == 0 || D == 0) return UG_OK;
    UG_REQUIRE(a && b && out && rows > 0 && D > 0, UG_ERR_BAD_SHAPE, "ug_add: bad arguments");
    UG_REQUIRE(D % 8 == 0 && lda % 8 == 0 && ldb % 8 == 0 && ldo % 8 == 0 && ug_aligned(a, 16) && ug_aligned(b, 16) && ug_aligned(out, 16),
               UG_ERR_BAD_ALIGN, "ug_add: 16-byte alignment required");
    const int64_t total = rows * (D / 8);
    const unsigned grid = (unsigned)std::min<int64_t>((total + 255) / 256, 2048);
    hipLaunchKernelGGL(add_kernel<T>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const T*)a, lda, (const T*)b, ldb, (T*)out, ldo, rows, (int)(D / 8));
    UG_CHECK_LAUNCH("ug_add");
    return UG_OK;
}

template <typename T, bool PACK>
int pack_latents_impl(const void* src, void* dst, int64_t B, int64_t C, int64_t H, int64_t W, ug_stream_t stream) {
    if (B == 0) return UG_OK;
    UG_REQUIRE(src && dst && B > 0 && C > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0, UG_ERR_BAD_SHAPE,
               "ug_pack_latents: need [B, C, H, W] with even H and W");
    UG_REQUIRE(B * C * H * W < (1ll << 40) && H < (1 << 20) && W < (1 << 20) && C < (1 << 20) && B < (1 << 20), UG_ERR_UNSUPPORTED, "ug_pack_latents: too large");
    const int64_t total = B * C * H * W;
    const unsigned grid = (unsigned)std::min<int64_t>((total + 255) / 256, 4096);
    hipLaunchKernelGGL((pack_latents_kernel<T, PACK>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (const T*)src, (T*)dst, (int)B, (int)C, (int)H, (int)W);
    UG_CHECK_LAUNCH("ug_pack_latents");
    return UG_OK;
}

}  // namespace

extern "C" int ug_cfg_combine(const void* u, const void* t, float gs, void* out, int64_t n, ug_stream_t s) { return cfg_combine_impl<bf16_t>(u, t, gs, out, n, s); }
extern "C" int ug_cfg_combine_f32(const void* u, const void* t, float gs, void* out, int64_t n, ug_stream_t s) { return cfg_combine_impl<float>(u, t, gs, out, n, s); }

extern "C" int ug_add_rowbcast_f32(void* x, int64_t ldx, const float* table, int64_t ldt, int64_t rows, int64_t rpb, int64_t D, ug_stream_t s) {
    return add_rowbcast_impl<bf16_t>(x, ldx, table, ldt, rows, rpb, D, s);
}
extern "C" int ug_add_rowbcast_f32_f32(void* x, int64_t ldx, const float* table, int64_t ldt, int64_t rows, int64_t rpb, int64_t D, ug_stream_t s) {
    return add_rowbcast_impl<float>(x, ldx, table, ldt, rows, rpb, D, s);
}

extern "C" int ug_gather_rows(const void* src, int64_t ld_src, const int32_t* idx, void* out, int64_t ld_out, int64_t n, int64_t W, ug_stream_t s) {
    return gather_rows_impl<bf16_t>(src, ld_src, idx, out, ld_out, n, W, s);
}
extern "C" int ug_gather_rows_f32(const void* src, int64_t ld_src, const int32_t* idx, void* out, int64_t ld_out, int64_t n, int64_t W, ug_stream_t s) {
    return gather_rows_impl<float>(src, ld_src, idx, out, ld_out, n, W, s);
}

extern "C" int ug_adaln_modulate(const void* x, int64_t ldx, int64_t x_rpb, int64_t x_bstride, const void* shift,
                                 const void* scale, int64_t mod_ld, int64_t rows_per_sample, void* out, int64_t ldo,
                                 int64_t rows, int64_t D, float eps, ug_stream_t stream) {
    return adaln_modulate_impl<bf16_t>(x, ldx, x_rpb, x_bstride, shift, scale, mod_ld, rows_per_sample, out, ldo, rows, D, eps, stream);
}
extern "C" int ug_adaln_modulate_f32(const void* x, int64_t ldx, int64_t x_rpb, int64_t x_bstride, const void* shift,
                                     const void* scale, int64_t mod_ld, int64_t rows_per_sample, void* out, int64_t ldo,
                                     int64_t rows, int64_t D, float eps, ug_stream_t stream) {
    return adaln_modulate_impl<float>(x, ldx, x_rpb, x_bstride, shift, scale, mod_ld, rows_per_sample, out, ldo, rows, D, eps, stream);
}

extern "C" int ug_qk_rmsnorm_rope(void* buf, int64_t ld, int64_t batches, int64_t rows_per_batch, int64_t batch_stride_rows,
                                  int64_t pos_offset, int64_t q_off, int64_t k_off, int32_t heads, int32_t dh, const void* wq_a, const void* wk_a,
                                  const void* wq_b, const void* wk_b, int64_t split, const float* cos_tab,
                                  const float* sin_tab, float eps, ug_stream_t stream) {
    return qk_rmsnorm_rope_impl<bf16_t>(buf, ld, batches, rows_per_batch, batch_stride_rows, pos_offset, q_off, k_off, heads, dh, wq_a, wk_a, wq_b, wk_b,
                                        split, cos_tab, sin_tab, eps, stream);
}
extern "C" int ug_qk_rmsnorm_rope_f32(void* buf, int64_t ld, int64_t batches, int64_t rows_per_batch, int64_t batch_stride_rows,
                                      int64_t pos_offset, int64_t q_off, int64_t k_off, int32_t heads, int32_t dh, const void* wq_a, const void* wk_a,
                                      const void* wq_b, const void* wk_b, int64_t split, const float* cos_tab,
                                      const float* sin_tab, float eps, ug_stream_t stream) {
    return qk_rmsnorm_rope_impl<float>(buf, ld, batches, rows_per_batch, batch_stride_rows, pos_offset, q_off, k_off, heads, dh, wq_a, wk_a, wq_b, wk_b,
                                       split, cos_tab, sin_tab, eps, stream);
}

extern "C" int ug_small_linear_bf16(const void* x, int64_t ldx, const void* W, int64_t ldw, const void* bias,
                                    const void* R, int64_t ldr, void* out, int64_t ldo, int64_t M, int64_t N, int64_t K,
                                    int32_t act_in, ug_stream_t stream) {
    if (M == 0 || N == 0) return UG_OK;
    UG_REQUIRE(x && W && out && M > 0 && N > 0 && K > 0, UG_ERR_BAD_SHAPE, "ug_small_linear_bf16: bad arguments");
    UG_REQUIRE(M <= 16, UG_ERR_UNSUPPORTED, "ug_small_linear_bf16: M=%lld > 16 (use ug_gemm_bf16)", (long long)M);
    UG_REQUIRE(K % 8 == 0 && ldx % 8 == 0 && ldw % 8 == 0 && ug_aligned(x, 16) && ug_aligned(W, 16), UG_ERR_BAD_ALIGN,
               "ug_small_linear_bf16: K, ldx, ldw must be multiples of 8 and bases 16-byte aligned");
    UG_REQUIRE(act_in == 0 || act_in == 1, UG_ERR_UNSUPPORTED, "ug_small_linear_bf16: act_in %d", act_in);
    const int MT = M <= 4 ? 4 : (M <= 8 ? 8 : 16);
    const size_t lds = (size_t)MT * K * 2;
    UG_REQUIRE(lds <= 128 * 1024, UG_ERR_UNSUPPORTED, "ug_small_linear_bf16: M*K too large for LDS staging");
    const int64_t cols_per_block = 4 * SL_COLS_PER_WAVE;
    const unsigned grid = (unsigned)((N + cols_per_block - 1) / cols_per_block);
    hipStream_t s = (hipStream_t)stream;
#define UG_SL_LAUNCH(MTV)                                                                                             \
    do {                                                                                                              \
        static bool set_ = false;                                                                                     \
        if (!set_) { (void)hipFuncSetAttribute((const void*)small_linear_kernel<MTV>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024); set_ = true; } \
        hipLaunchKernelGGL(small_linear_kernel<MTV>, dim3(grid), dim3(256), lds, s, (const bf16_t*)x, ldx, (const bf16_t*)W, ldw,  \
                           (const bf16_t*)bias, (const bf16_t*)R, ldr, (bf16_t*)out, ldo, (int)M, N, (int)K, act_in); \
    } while (0)
    if (MT == 4) UG_SL_LAUNCH(4); else if (MT == 8) UG_SL_LAUNCH(8); else UG_SL_LAUNCH(16);
#undef UG_SL_LAUNCH
    UG_CHECK_LAUNCH("ug_small_linear_bf16");
    return UG_OK;
}

extern "C" int ug_small_linear_f32(const void* x, int64_t ldx, const void* W, int64_t ldw, const void* bias,
                                   const void* R, int64_t ldr, void* out, int64_t ldo, int64_t M, int64_t N, int64_t K,
                                   int32_t act_in, ug_stream_t stream) {
    if (M == 0 || N == 0) return UG_OK;
    UG_REQUIRE(x && W && out && M > 0 && N > 0 && K > 0 && M <= 64, UG_ERR_BAD_SHAPE, "ug_small_linear_f32: bad arguments");
    UG_REQUIRE(act_in == 0 || act_in == 1, UG_ERR_UNSUPPORTED, "ug_small_linear_f32: act_in %d", act_in);
    hipLaunchKernelGGL(small_linear_f32_kernel, dim3((unsigned)((N + 3) / 4)), dim3(256), 0, (hipStream_t)stream, (const float*)x, ldx,
                       (const float*)W, ldw, (const float*)bias, (const float*)R, ldr, (float*)out, ldo, (int)M, N, (int)K, act_in);
    UG_CHECK_LAUNCH("ug_small_linear_f32");
    return UG_OK;
}

extern "C" int ug_timestep_embed(const float* t, void* out, int64_t ldo, int64_t B, int32_t dim, ug_stream_t s) { return timestep_embed_impl<bf16_t>(t, out, ldo, B, dim, s); }
extern "C" int ug_timestep_embed_f32(const float* t, void* out, int64_t ldo, int64_t B, int32_t dim, ug_stream_t s) { return timestep_embed_impl<float>(t, out, ldo, B, dim, s); }

extern "C" int ug_euler_step(void* x, const void* v, float dt, int64_t n, ug_stream_t s) { return euler_step_impl<bf16_t>(x, v, dt, n, s); }
extern "C" int ug_euler_step_f32(void* x, const void* v, float dt, int64_t n, ug_stream_t s) { return euler_step_impl<float>(x, v, dt, n, s); }

extern "C" int ug_add_bf16(const void* a, int64_t lda, const void* b, int64_t ldb, void* out, int64_t ldo, int64_t rows, int64_t D, ug_stream_t s) {
    return add_impl<bf16_t>(a, lda, b, ldb, out, ldo, rows, D, s);
}
extern "C" int ug_add_f32(const void* a, int64_t lda, const void* b, int64_t ldb, void* out, int64_t ldo, int64_t rows, int64_t D, ug_stream_t s) {
    return add_impl<float>(a, lda, b, ldb, out, ldo, rows, D, s);
}

extern "C" int ug_pack_latents(const void* latents, void* packed, int64_t B, int64_t C, int64_t H, int64_t W, ug_stream_t s) {
    return pack_latents_impl<bf16_t, true>(latents, packed, B, C, H, W, s);
}
extern "C" int ug_unpack_latents(const void* packed, void* latents, int64_t B, int64_t C, int64_t H, int64_t W, ug_stream_t s) {
    return pack_latents_impl<bf16_t, false>(packed, latents, B, C, H, W, s);
}
extern "C" int ug_pack_latents_f32(const void* latents, void* packed, int64_t B, int64_t C, int64_t H, int64_t W, ug_stream_t s) {
    return pack_latents_impl<float, true>(latents, packed, B, C, H, W, s);
}
extern "C" int ug_unpack_latents_f32(const void* packed, void* latents, int64_t B, int64_t C, int64_t H, int64_t W, ug_stream_t s) {
    return pack_latents_impl<float, false>(packed, latents, B, C, H, W, s);
}
