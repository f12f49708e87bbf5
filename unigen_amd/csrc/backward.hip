// Backward-pass kernels of the control-module training step (reference train.py:622-662 calls
// accelerator.backward(loss) on the same forward). The matrix work of the backward runs through ug_gemm_bf16 (dX = dY W, dW = dY^T X,
// and the attention backward as five grouped GEMMs per sample, see unigen_amd/autograd.py); this file holds what sits between those GEMMs:
//   ug_transpose            batched 2-D transpose with zero padding of the new row length (operand layouts of dgrad / wgrad)
//   ug_colsum               per-group column sums of a or a (.) b, fp32 accumulation (bias / gate / shift / scale / RMSNorm-weight gradients)
//   ug_gelu_tanh(_bwd)      GELU(tanh) and its derivative (FeedForward net.0, proj_mlp)
//   ug_adaln_modulate_bwd   LayerNorm(x) (1 + scale) + shift  ->  dx, and dy (.) xhat for the scale gradient
//   ug_qk_rmsnorm_rope_bwd  RMSNorm(q|k) . weight, apply_rotary_emb  ->  d(q|k), and du (.) xhat for the weight gradient
//   ug_row_lse, ug_attn_prob, ug_attn_dscore, ug_rowdot   softmax statistics and the elementwise steps of the attention backward
// Every kernel is a template over the element type (bf16 product / fp32 verification twin); arithmetic in fp32, outputs rounded once.
#include "ug_common.h"
#include <algorithm>

namespace {

// gelu_tanh(x) = x / (1 + exp(-2u)), u = sqrt(2/pi) (x + 0.044715 x^3); same formulation as the GEMM epilogue (gemm_epilogue.h)
__device__ __forceinline__ float gelu_f(float x) {
    constexpr float C0 = -2.3022081986f, C1 = C0 * 0.044715f;
    const float u = x * fmaf(x * x, C1, C0);
    return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(u));
}
__device__ __forceinline__ float gelu_grad_f(float x) {
    // y = x s, s = sigmoid(2u): dy/dx = s + x s (1 - s) 2 du/dx, 2 du/dx = 2 sqrt(2/pi) (1 + 3 * 0.044715 x^2)
    constexpr float K2 = 1.5957691216f;           // 2 sqrt(2/pi)
    const float u2 = K2 * x * fmaf(0.044715f * x, x, 1.0f);
    const float s = 1.0f / (1.0f + __expf(-u2));
    return s + x * s * (1.0f - s) * K2 * fmaf(3.0f * 0.044715f * x, x, 1.0f);
}

template <typename T>
__global__ __launch_bounds__(256) void transpose_kernel(const T* __restrict__ src, int64_t ld_src, int64_t src_bstride, T* __restrict__ dst,
                                                        int64_t ld_dst, int64_t dst_bstride, int rows, int cols, int rows_pad) {
    __shared__ T tile[64][65];
    const T* s = src + (int64_t)blockIdx.z * src_bstride;
    T* d = dst + (int64_t)blockIdx.z * dst_bstride;
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int i = ty; i < 64; i += 4) {
        const int r = r0 + i, c = c0 + tx;
        T v; ElemT<T>::st(&v, 0.f);
        if (r < rows && c < cols) v = s[(int64_t)r * ld_src + c];
        tile[i][tx] = v;
    }
    __syncthreads();
    for (int i = ty; i < 64; i += 4) {
        const int c = c0 + i, r = r0 + tx;       // dst row = c, dst column = r
        if (c < cols && r < rows_pad) d[(int64_t)c * ld_dst + r] = tile[tx][i];
    }
}

// bf16 fast path of the transpose: 16-byte global loads and stores. A 64 x 64 tile goes through LDS transposed ([column][row], pitch 68 elements:
// 8-byte aligned rows, the 2-byte scatter writes of one instruction fall on 32 distinct banks twice); needs cols, rows_pad, both leading dimensions
// and batch strides to be multiples of 8 and 16-byte aligned bases. 2.5 -> ~4 TB/s (read + write) on the wgrad operands of a training step.
__global__ __launch_bounds__(256) void transpose8_kernel(const bf16_t* __restrict__ src, int64_t ld_src, int64_t src_bstride, bf16_t* __restrict__ dst,
                                                         int64_t ld_dst, int64_t dst_bstride, int rows, int cols, int rows_pad) {
    constexpr int LD = 68;
    __shared__ __attribute__((aligned(16))) bf16_t tile[64 * LD];
    const bf16_t* s = src + (int64_t)blockIdx.z * src_bstride;
    bf16_t* d = dst + (int64_t)blockIdx.z * dst_bstride;
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int id = threadIdx.x + 256 * u, r = id >> 3, ch = id & 7;
        u32x4 v = {0u, 0u, 0u, 0u};
        if (r0 + r < rows && c0 + 8 * ch < cols) v = *(const u32x4*)(s + (int64_t)(r0 + r) * ld_src + c0 + 8 * ch);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            tile[(8 * ch + 2 * j) * LD + r] = (bf16_t)(v[j] & 0xffffu);
            tile[(8 * ch + 2 * j + 1) * LD + r] = (bf16_t)(v[j] >> 16);
        }
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int id = threadIdx.x + 256 * u, c = id >> 3, rch = id & 7;       // dst row = source column c0 + c, dst columns r0 + 8 rch .. + 7
        if (c0 + c < cols && r0 + 8 * rch < rows_pad) {
            const u32x2 lo = *(const u32x2*)(tile + c * LD + 8 * rch), hi = *(const u32x2*)(tile + c * LD + 8 * rch + 4);
            u32x4 w; w.x = lo.x; w.y = lo.y; w.z = hi.x; w.w = hi.y;
            *(u32x4*)(d + (int64_t)(c0 + c) * ld_dst + r0 + 8 * rch) = w;
        }
    }
}

// out[g][c] = rnd(alpha * sum_{r in group g} a[r][c] * (b ? b[r][c] : 1)) in two deterministic stages: a block sums COLSUM_ROWS rows of 512 columns
// (8 per lane, 16-byte loads; its 4 waves take every 4th row, then meet in LDS) into an fp32 partial; the second stage adds the partials of a
// group in order.
constexpr int COLSUM_ROWS = 128;
template <typename T>
__global__ __launch_bounds__(256) void colsum_partial_kernel(const T* __restrict__ a, int64_t lda, const T* __restrict__ b, int64_t ldb, float* __restrict__ part,
                                                             int64_t rows_per_group, int cols, int nchunks) {
    __shared__ float red[4][512];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int c0 = blockIdx.x * 512 + lane * 8;
    const int64_t r0 = (int64_t)blockIdx.z * rows_per_group + (int64_t)blockIdx.y * COLSUM_ROWS;
    const int64_t r1 = min((int64_t)blockIdx.z * rows_per_group + rows_per_group, r0 + COLSUM_ROWS);
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (c0 < cols) {
        for (int64_t r = r0 + w; r < r1; r += 4) {
            float va[8], vb[8];
            ElemT<T>::load8(a + r * lda + c0, va);
            if (b) {
                ElemT<T>::load8(b + r * ldb + c0, vb);
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[i] += va[i] * vb[i];
            } else {
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[i] += va[i];
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) red[w][lane * 8 + i] = acc[i];
    __syncthreads();
    for (int c = threadIdx.x; c < 512; c += 256) {
        const int col = blockIdx.x * 512 + c;
        if (col < cols) part[((int64_t)blockIdx.z * nchunks + blockIdx.y) * cols + col] = red[0][c] + red[1][c] + red[2][c] + red[3][c];
    }
}
template <typename T>
__global__ void colsum_final_kernel(const float* __restrict__ part, T* __restrict__ out, int64_t ldo, int cols, int nchunks, int groups, float alpha) {
    const int64_t total = (int64_t)groups * cols;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int g = (int)(i / cols), c = (int)(i - (int64_t)g * cols);
        const float* pp = part + (int64_t)g * nchunks * cols + c;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;          // four loads in flight per lane; the order of the additions is still fixed
        int k = 0;
        for (; k + 4 <= nchunks; k += 4) {
            s0 += pp[(int64_t)k * cols]; s1 += pp[(int64_t)(k + 1) * cols]; s2 += pp[(int64_t)(k + 2) * cols]; s3 += pp[(int64_t)(k + 3) * cols];
        }
        for (; k < nchunks; ++k) s0 += pp[(int64_t)k * cols];
        ElemT<T>::st(out + (int64_t)g * ldo + c, alpha * ((s0 + s1) + (s2 + s3)));
    }
}

// y[r][:] = (x ? x[r][:] : 0) + rnd(gate[r / rows_per_sample][:] * a[r][:]): the gated residual `x + gate.unsqueeze(1) * a` of every block (forward of the
// training path; with x = NULL its backward d a = gate * d y). 8 elements per thread.
template <typename T>
__global__ void gate_residual_kernel(const T* __restrict__ x, int64_t ldx, const T* __restrict__ a, int64_t lda, const T* __restrict__ gate, int64_t gate_ld,
                                     int64_t rows_per_sample, T* __restrict__ y, int64_t ldy, int64_t rows, int D8) {
    const int64_t total = rows * D8;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / D8; const int c = (int)(i - r * D8) * 8;
        float av[8], gv[8], xv[8], o[8];
        ElemT<T>::load8(a + r * lda + c, av);
        ElemT<T>::load8(gate + (r / rows_per_sample) * gate_ld + c, gv);
        if (x) ElemT<T>::load8(x + r * ldx + c, xv);
#pragma unroll
        for (int k = 0; k < 8; ++k) { const float p = ElemT<T>::rnd(gv[k] * av[k]); o[k] = x ? xv[k] + p : p; }
        ElemT<T>::store8(y + r * ldy + c, o);
    }
}

// 8 elements per lane and access (16-byte loads / stores) when n and the bases allow (VEC); element-wise otherwise
template <typename T, bool VEC>
__global__ void gelu_kernel(const T* __restrict__ x, T* __restrict__ y, int64_t n) {
    if constexpr (VEC) {
        for (int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 8; i < n; i += (int64_t)gridDim.x * blockDim.x * 8) {
            float v[8];
            ElemT<T>::load8(x + i, v);
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = gelu_f(v[k]);
            ElemT<T>::store8(y + i, v);
        }
    } else {
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
            ElemT<T>::st(y + i, gelu_f(ElemT<T>::ld(x + i)));
    }
}
template <typename T, bool VEC>
__global__ void gelu_bwd_kernel(const T* __restrict__ x, const T* __restrict__ dy, T* __restrict__ dx, int64_t n) {
    if constexpr (VEC) {
        for (int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 8; i < n; i += (int64_t)gridDim.x * blockDim.x * 8) {
            float v[8], g[8];
            ElemT<T>::load8(x + i, v); ElemT<T>::load8(dy + i, g);
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = g[k] * gelu_grad_f(v[k]);
            ElemT<T>::store8(dx + i, v);
        }
    } else {
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
            ElemT<T>::st(dx + i, ElemT<T>::ld(dy + i) * gelu_grad_f(ElemT<T>::ld(x + i)));
    }
}

// y = xhat (1 + s) + sh, xhat = (x - mu) rstd:   g = dy (1 + s);  dx = rstd (g - mean(g) - xhat mean(g xhat));  d shift[sample] = sum_rows dy,
// d scale[sample] = sum_rows dy xhat. Block (p, sample) takes the p-th chunk of the sample's rows; its W = ceil(D / 512) <= 8 waves share every row,
// each wave one 512-column segment (8 elements per lane, one 16-byte access per operand), the row statistics meet in LDS (two barriers per row,
// slots alternating by row parity). x and dy are read once; every column belongs to one lane of one wave, so the two column sums are accumulated in
// registers over the block's rows and stored straight into the block's fp32 partial [2][D] - the caller adds the partials of a sample.
// (First version: dy xhat written as a tensor, both sums left to ug_colsum - twice the traffic, four more launches. Second: one wave per row with
// 48 elements per lane - 252 registers, latency-bound at 1.8 TB/s.)
template <typename T>
__global__ __launch_bounds__(512) void adaln_bwd_kernel(const T* __restrict__ x, int64_t ldx, const T* __restrict__ dy, int64_t lddy,
                                                        const T* __restrict__ scale, int64_t mod_ld, int64_t rows_per_sample, T* __restrict__ dx,
                                                        int64_t lddx, float* __restrict__ part /* [samples][gridDim.x][2][D] */, int D, float eps,
                                                        int64_t samples) {
    __shared__ float slot[2][4][8];                                  // [row parity][s1, s2, a, b][wave]
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, W = blockDim.x >> 6;
    for (int64_t sample = blockIdx.y; sample < samples; sample += gridDim.y) {        // gridDim.y = min(samples, 65535)
    const int64_t chunk = (rows_per_sample + gridDim.x - 1) / gridDim.x;
    const int64_t r_lo = (int64_t)blockIdx.x * chunk, r_hi = r_lo + chunk < rows_per_sample ? r_lo + chunk : rows_per_sample;
    const int c = 8 * (lane + 64 * wv);
    const bool in = c < D;
    float sv[8], pd[8], px[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { pd[k] = px[k] = 0.f; sv[k] = 0.f; }
    if (in) {
        ElemT<T>::load8(scale + sample * mod_ld + c, sv);
#pragma unroll
        for (int k = 0; k < 8; ++k) sv[k] += 1.0f;
    }
    int par = 0;
    for (int64_t rr = r_lo; rr < r_hi; ++rr, par ^= 1) {
        const int64_t row = sample * rows_per_sample + rr;
        float xv[8], dv[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) xv[k] = dv[k] = 0.f;
        if (in) { ElemT<T>::load8(x + row * ldx + c, xv); ElemT<T>::load8(dy + row * lddy + c, dv); }
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) { s1 += xv[k]; s2 += xv[k] * xv[k]; }
        s1 = wave_sum(s1); s2 = wave_sum(s2);
        if (lane == 0) { slot[par][0][wv] = s1; slot[par][1][wv] = s2; }
        __syncthreads();
        s1 = 0.f; s2 = 0.f;
        for (int w8 = 0; w8 < W; ++w8) { s1 += slot[par][0][w8]; s2 += slot[par][1][w8]; }
        const float mu = s1 / D;
        const float rstd = rsqrtf(fmaxf(s2 / D - mu * mu, 0.f) + eps);
        float a = 0.f, bsum = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            xv[k] = in ? (xv[k] - mu) * rstd : 0.f;                  // xhat from here on
            const float g = dv[k] * sv[k];
            a += g; bsum += g * xv[k];
        }
        a = wave_sum(a); bsum = wave_sum(bsum);
        if (lane == 0) { slot[par][2][wv] = a; slot[par][3][wv] = bsum; }
        __syncthreads();
        a = 0.f; bsum = 0.f;
        for (int w8 = 0; w8 < W; ++w8) { a += slot[par][2][w8]; bsum += slot[par][3][w8]; }
        a /= D; bsum /= D;
        if (in) {
            float o[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                o[k] = rstd * (dv[k] * sv[k] - a - xv[k] * bsum);
                pd[k] += dv[k]; px[k] += dv[k] * xv[k];
            }
            ElemT<T>::store8(dx + row * lddx + c, o);
        }
    }
    if (in) {
        float* out = part + (sample * gridDim.x + blockIdx.x) * 2 * (int64_t)D;
#pragma unroll
        for (int k = 0; k < 8; ++k) { out[c + k] = pd[k]; out[D + c + k] = px[k]; }
    }
    __syncthreads();                                                  // the slots are reused by the block's next sample
    }
}

// One wave per (row, head) vector of DH elements (lane l: pairs l, l + 64 < DH / 2), a grid-stride loop over the vectors: forward was u = x rs,
// un = u w, y = rope(un).   dun = rope^T(dy);  du = dun w;  dx = rs (du - u mean(du u));  d weight = sum over rows and heads of dun u.
// Every element is loaded once (a pair per 4- / 8-byte load); the weight gradient is accumulated in registers over the wave's vectors, the four
// waves of a block meet in LDS in a fixed order and the block writes ONE fp32 partial row [DH] - the caller adds the <= 2048 partial rows
// (the first version wrote a [vectors][DH] product tensor for ug_colsum: 2/5 of its traffic and a second pair of launches).
constexpr int QKB_MAXP = 2;        // pairs per lane: head widths up to 256
template <typename T>
__global__ __launch_bounds__(256) void qk_bwd_kernel(const T* __restrict__ x, int64_t ldx, const T* __restrict__ dy, int64_t lddy, T* __restrict__ dx,
                                                     int64_t lddx, float* __restrict__ dw_part /* [gridDim.x][DH] */, const T* __restrict__ w,
                                                     const float* __restrict__ cos_tab, const float* __restrict__ sin_tab, int64_t rows_per_batch,
                                                     int64_t pos_offset, int64_t nvec, int heads, int DH, float eps) {
    __shared__ float red[4][2 * 64 * QKB_MAXP];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    float dwa[QKB_MAXP][2], wq[QKB_MAXP][2];
#pragma unroll
    for (int i = 0; i < QKB_MAXP; ++i) {
        const int p = lane + 64 * i;
        dwa[i][0] = dwa[i][1] = 0.f; wq[i][0] = wq[i][1] = 0.f;
        if (w && p < DH / 2) { wq[i][0] = ElemT<T>::ld(w + 2 * p); wq[i][1] = ElemT<T>::ld(w + 2 * p + 1); }
    }
    for (int64_t vec = (int64_t)blockIdx.x * 4 + wv; vec < nvec; vec += (int64_t)gridDim.x * 4) {
        const int64_t row = vec / heads; const int h = (int)(vec - row * heads);
        const int64_t pos = pos_offset + row % rows_per_batch;
        const T* xr = x + row * ldx + (int64_t)h * DH; const T* gr = dy + row * lddy + (int64_t)h * DH;
        float xa[QKB_MAXP][2], g[QKB_MAXP][2];
        float ss = 0.f;
#pragma unroll
        for (int i = 0; i < QKB_MAXP; ++i) {
            const int p = lane + 64 * i;
            xa[i][0] = xa[i][1] = g[i][0] = g[i][1] = 0.f;
            if (p < DH / 2) {
                ElemT<T>::load2(xr + 2 * p, xa[i]);
                ElemT<T>::load2(gr + 2 * p, g[i]);
                if (cos_tab) {
                    const float2 cc = *(const float2*)(cos_tab + pos * DH + 2 * p), sn = *(const float2*)(sin_tab + pos * DH + 2 * p);
                    // y0 = a c0 - b s0, y1 = b c1 + a s1  ->  da = g0 c0 + g1 s1, db = g1 c1 - g0 s0
                    const float t0 = g[i][0] * cc.x + g[i][1] * sn.y, t1 = g[i][1] * cc.y - g[i][0] * sn.x;
                    g[i][0] = t0; g[i][1] = t1;
                }
                ss += xa[i][0] * xa[i][0] + xa[i][1] * xa[i][1];
            }
        }
        float rs = 1.0f, dot = 0.f;
        if (w) {
            ss = wave_sum(ss);
            rs = rsqrtf(ss / DH + eps);
#pragma unroll
            for (int i = 0; i < QKB_MAXP; ++i) dot += g[i][0] * wq[i][0] * (xa[i][0] * rs) + g[i][1] * wq[i][1] * (xa[i][1] * rs);
            dot = wave_sum(dot) / DH;
        }
#pragma unroll
        for (int i = 0; i < QKB_MAXP; ++i) {
            const int p = lane + 64 * i;
            if (p < DH / 2) {
                float o[2] = {g[i][0], g[i][1]};
                if (w) {
                    const float u0 = xa[i][0] * rs, u1 = xa[i][1] * rs;
                    dwa[i][0] += g[i][0] * u0; dwa[i][1] += g[i][1] * u1;
                    o[0] = rs * (g[i][0] * wq[i][0] - u0 * dot); o[1] = rs * (g[i][1] * wq[i][1] - u1 * dot);
                }
                ElemT<T>::store2(dx + row * lddx + (int64_t)h * DH + 2 * p, o);
            }
        }
    }
    if (w) {
#pragma unroll
        for (int i = 0; i < QKB_MAXP; ++i) { red[wv][2 * (lane + 64 * i)] = dwa[i][0]; red[wv][2 * (lane + 64 * i) + 1] = dwa[i][1]; }
        __syncthreads();
        for (int e = threadIdx.x; e < DH; e += 256) dw_part[(int64_t)blockIdx.x * DH + e] = ((red[0][e] + red[1][e]) + red[2][e]) + red[3][e];
    }
}

// The same backward with 16-byte accesses: a (row, head) vector takes LPV = DH / 8 lanes (8 consecutive elements = 4 rotation pairs per lane), a wave
// handles 64 / LPV vectors per iteration, the two reductions run over the LPV lanes of a vector (xor shuffles). Head widths 64 / 128 / 256.
template <typename T, int LPV>
__global__ __launch_bounds__(256) void qk_bwd8_kernel(const T* __restrict__ x, int64_t ldx, const T* __restrict__ dy, int64_t lddy, T* __restrict__ dx,
                                                      int64_t lddx, float* __restrict__ dw_part /* [gridDim.x][DH] */, const T* __restrict__ w,
                                                      const float* __restrict__ cos_tab, const float* __restrict__ sin_tab, int64_t rows_per_batch,
                                                      int64_t pos_offset, int64_t nvec, int heads, float eps) {
    constexpr int VPW = 64 / LPV, DH = 8 * LPV;
    __shared__ float red[4][64][8];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int sub = lane % LPV, slot = lane / LPV;
    float wq[8], dwa[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { dwa[k] = 0.f; wq[k] = w ? ElemT<T>::ld(w + 8 * sub + k) : 0.f; }
    auto group_sum = [](float v) {
#pragma unroll
        for (int m = 1; m < LPV; m <<= 1) v += __shfl_xor(v, m, 64);
        return v;
    };
    for (int64_t vec0 = ((int64_t)blockIdx.x * 4 + wv) * VPW; vec0 < nvec; vec0 += (int64_t)gridDim.x * 4 * VPW) {
        const int64_t vec = vec0 + slot;
        const bool valid = vec < nvec;
        const int64_t vc = valid ? vec : nvec - 1;
        const int64_t row = vc / heads; const int h = (int)(vc - row * heads);
        const int64_t pos = pos_offset + row % rows_per_batch;
        float xa[8], g[8];
        ElemT<T>::load8(x + row * ldx + (int64_t)h * DH + 8 * sub, xa);
        ElemT<T>::load8(dy + row * lddy + (int64_t)h * DH + 8 * sub, g);
        if (cos_tab) {
            const f32x4 c0 = *(const f32x4*)(cos_tab + pos * DH + 8 * sub), c1 = *(const f32x4*)(cos_tab + pos * DH + 8 * sub + 4);
            const f32x4 s0 = *(const f32x4*)(sin_tab + pos * DH + 8 * sub), s1 = *(const f32x4*)(sin_tab + pos * DH + 8 * sub + 4);
            const float cc[8] = {c0[0], c0[1], c0[2], c0[3], c1[0], c1[1], c1[2], c1[3]}, sn[8] = {s0[0], s0[1], s0[2], s0[3], s1[0], s1[1], s1[2], s1[3]};
#pragma unroll
            for (int q = 0; q < 8; q += 2) {       // y0 = a c0 - b s0, y1 = b c1 + a s1  ->  da = g0 c0 + g1 s1, db = g1 c1 - g0 s0
                const float t0 = g[q] * cc[q] + g[q + 1] * sn[q + 1], t1 = g[q + 1] * cc[q + 1] - g[q] * sn[q];
                g[q] = t0; g[q + 1] = t1;
            }
        }
        float rs = 1.0f, dot = 0.f;
        if (w) {
            float ss = 0.f;
#pragma unroll
            for (int k = 0; k < 8; ++k) ss += xa[k] * xa[k];
            rs = rsqrtf(group_sum(ss) / DH + eps);
#pragma unroll
            for (int k = 0; k < 8; ++k) dot += g[k] * wq[k] * (xa[k] * rs);
            dot = group_sum(dot) / DH;
        }
        float o[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            o[k] = g[k];
            if (w) {
                const float u = xa[k] * rs;
                if (valid) dwa[k] += g[k] * u;
                o[k] = rs * (g[k] * wq[k] - u * dot);
            }
        }
        if (valid) ElemT<T>::store8(dx + row * lddx + (int64_t)h * DH + 8 * sub, o);
    }
    if (w) {
#pragma unroll
        for (int k = 0; k < 8; ++k) red[wv][lane][k] = dwa[k];
        __syncthreads();
        for (int e = threadIdx.x; e < DH; e += 256) {
            float acc = 0.f;
            for (int w4 = 0; w4 < 4; ++w4)
                for (int sl = 0; sl < VPW; ++sl) acc += red[w4][sl * LPV + e / 8][e % 8];
            dw_part[(int64_t)blockIdx.x * DH + e] = acc;
        }
    }
}

// lse[r] = log(sum_c exp(scale S[r][c])), natural log; one block per row
__global__ __launch_bounds__(256) void row_lse_kernel(const float* __restrict__ S, int64_t ld, float* __restrict__ lse, int cols, float scale) {
    __shared__ float red[256];
    const float* s = S + (int64_t)blockIdx.x * ld;
    float mx = -INFINITY;
    for (int c = threadIdx.x; c < cols; c += 256) mx = fmaxf(mx, s[c] * scale);
    red[threadIdx.x] = mx; __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + o]); __syncthreads(); }
    mx = red[0]; __syncthreads();
    float sum = 0.f;
    for (int c = threadIdx.x; c < cols; c += 256) sum += expf(s[c] * scale - mx);
    red[threadIdx.x] = sum; __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x == 0) lse[blockIdx.x] = mx + logf(red[0]);
}
// P[r][c] = c < valid_cols ? exp(scale S[r][c] - lse[r]) : 0   (columns past valid_cols are zero-padded keys)
template <typename T>
__global__ void attn_prob_kernel(const float* __restrict__ S, int64_t ld_s, const float* __restrict__ lse, T* __restrict__ P, int64_t ld_p, int64_t rows, int cols,
                                 int valid_cols, float scale) {
    const int64_t total = rows * cols;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / cols; const int c = (int)(i - r * cols);
        ElemT<T>::st(P + r * ld_p + c, c < valid_cols ? expf(S[r * ld_s + c] * scale - lse[r]) : 0.f);
    }
}
// dS[r][c] = scale P[r][c] (dP[r][c] - delta[r])
template <typename T>
__global__ void attn_dscore_kernel(const T* __restrict__ P, int64_t ld_p, const float* __restrict__ dP, int64_t ld_dp, const float* __restrict__ delta,
                                   T* __restrict__ dS, int64_t ld_ds, int64_t rows, int cols, float scale) {
    const int64_t total = rows * cols;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / cols; const int c = (int)(i - r * cols);
        ElemT<T>::st(dS + r * ld_ds + c, scale * ElemT<T>::ld(P + r * ld_p + c) * (dP[r * ld_dp + c] - delta[r]));
    }
}
// out[g][r] = sum_c a[r][g * cols + c] b[r][g * cols + c]   (delta = rowsum(dO . O) per head): one wave per (row, group)
template <typename T>
__global__ __launch_bounds__(256) void rowdot_kernel(const T* __restrict__ a, int64_t lda, const T* __restrict__ b, int64_t ldb, float* __restrict__ out,
                                                     int64_t rows, int groups, int cols) {
    const int lane = threadIdx.x & 63;
    const int64_t id = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (id >= rows * groups) return;
    const int64_t r = id / groups; const int g = (int)(id - r * groups);
    float acc = 0.f;
    for (int c = lane; c < cols; c += 64) acc += ElemT<T>::ld(a + r * lda + (int64_t)g * cols + c) * ElemT<T>::ld(b + r * ldb + (int64_t)g * cols + c);
    acc = wave_sum(acc);
    if (lane == 0) out[(int64_t)g * rows + r] = acc;
}

unsigned grid1d(int64_t n, int per) { return (unsigned)std::min<int64_t>((n + per - 1) / per, 65535 * 16); }

template <typename T>
int transpose_impl(const void* src, int64_t ld_src, int64_t src_bstride, void* dst, int64_t ld_dst, int64_t dst_bstride, int64_t batch, int64_t rows,
                   int64_t cols, int64_t rows_pad, ug_stream_t stream) {
    if (batch == 0 || rows == 0 || cols == 0) return UG_OK;
    UG_REQUIRE(src && dst && rows > 0 && cols > 0 && batch > 0 && batch < 65536 && rows_pad >= rows && ld_src >= cols && ld_dst >= rows_pad, UG_ERR_BAD_SHAPE,
               "ug_transpose: bad arguments");
    dim3 grid((unsigned)((cols + 63) / 64), (unsigned)((rows_pad + 63) / 64), (unsigned)batch);
    UG_REQUIRE(grid.y < 65536, UG_ERR_UNSUPPORTED, "ug_transpose: too many rows");
    if constexpr (!ElemT<T>::kF32) {
        if (cols % 8 == 0 && rows_pad % 8 == 0 && ld_src % 8 == 0 && ld_dst % 8 == 0 && src_bstride % 8 == 0 && dst_bstride % 8 == 0 && ug_aligned(src, 16) &&
            ug_aligned(dst, 16)) {
            hipLaunchKernelGGL(transpose8_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)src, ld_src, src_bstride, (bf16_t*)dst, ld_dst, dst_bstride,
                               (int)rows, (int)cols, (int)rows_pad);
            UG_CHECK_LAUNCH("ug_transpose");
            return UG_OK;
        }
    }
    hipLaunchKernelGGL(transpose_kernel<T>, grid, dim3(256), 0, (hipStream_t)stream, (const T*)src, ld_src, src_bstride, (T*)dst, ld_dst, dst_bstride,
                       (int)rows, (int)cols, (int)rows_pad);
    UG_CHECK_LAUNCH("ug_transpose");
    return UG_OK;
}
template <typename T>
int colsum_impl(const void* a, int64_t lda, const void* b, int64_t ldb, void* out, int64_t ldo, int64_t rows, int64_t cols, int64_t rows_per_group,
                float alpha, void* workspace, int64_t workspace_bytes, ug_stream_t stream) {
    if (rows == 0 || cols == 0) return UG_OK;
    UG_REQUIRE(a && out && rows > 0 && cols > 0 && rows_per_group > 0 && rows % rows_per_group == 0 && rows / rows_per_group < 65536 && lda >= cols &&
               ldo >= cols && (!b || ldb >= cols), UG_ERR_BAD_SHAPE, "ug_colsum: bad arguments");
    constexpr int EB = ElemT<T>::kF32 ? 4 : 2;
    UG_REQUIRE(cols % 8 == 0 && lda % 8 == 0 && (!b || ldb % 8 == 0) && ug_aligned(a, 8 * EB > 16 ? 16 : 8 * EB) && (!b || ug_aligned(b, 8 * EB > 16 ? 16 : 8 * EB)),
               UG_ERR_BAD_ALIGN, "ug_colsum: cols and leading dimensions must be multiples of 8, bases 16-byte aligned");
    const int64_t groups = rows / rows_per_group;
    const int nchunks = (int)((rows_per_group + COLSUM_ROWS - 1) / COLSUM_ROWS);
    UG_REQUIRE(workspace && workspace_bytes >= groups * nchunks * cols * (int64_t)sizeof(float) && nchunks < 65536, UG_ERR_BAD_SHAPE,
               "ug_colsum: workspace of ug_colsum_workspace_bytes() needed");
    dim3 grid((unsigned)((cols + 511) / 512), (unsigned)nchunks, (unsigned)groups);
    hipLaunchKernelGGL(colsum_partial_kernel<T>, grid, dim3(256), 0, (hipStream_t)stream, (const T*)a, lda, (const T*)b, ldb, (float*)workspace, rows_per_group,
                       (int)cols, nchunks);
    hipLaunchKernelGGL(colsum_final_kernel<T>, dim3(grid1d(groups * cols, 256)), dim3(256), 0, (hipStream_t)stream, (const float*)workspace, (T*)out, ldo, (int)cols,
                       nchunks, (int)groups, alpha);
    UG_CHECK_LAUNCH("ug_colsum");
    return UG_OK;
}

// ---------------------------------------------------------------------------------------------------------
// Backward of the top-1 gate (deepspeed TopKGate, src/UniGenUtils.py:99): gates = softmax(F.linear((x + c).float(), wg.float())).
// Given d gates [S, E] (fp32, from l_aux and the combine weights):  d logits = gates * (d gates - sum_e d gates * gates);
//   d(x + c)[s] = sum_e d logits[s, e] * wg[e]        (one wave per token; the same tensor is d x and d c)
//   d wg[e]     = sum_s d logits[s, e] * bf16(x + c)[s]  (column blocks x token slices -> fp32 partials, added in a fixed order by the caller)
// Round 2 ran this through F.linear / autograd (vendor BLAS on a product path); these two kernels read x and c once each.
// ---------------------------------------------------------------------------------------------------------
constexpr int GBW_MAXE = 16;
__device__ __forceinline__ void gate_dlogits(const float* __restrict__ g, const float* __restrict__ dg, int E, float* dl) {
    float dot = 0.f;
#pragma unroll
    for (int e = 0; e < GBW_MAXE; ++e) if (e < E) dot += dg[e] * g[e];
#pragma unroll
    for (int e = 0; e < GBW_MAXE; ++e) dl[e] = e < E ? g[e] * (dg[e] - dot) : 0.f;
}
template <typename T>
__global__ __launch_bounds__(256) void moe_gate_bwd_dx_kernel(const float* __restrict__ gates, const float* __restrict__ dgates, const T* __restrict__ wg,
                                                              int64_t S, int D, int E, T* __restrict__ dx, int64_t lddx) {
    using EL = ElemT<T>;
    const int lane = threadIdx.x & 63;
    const int64_t s = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (s >= S) return;
    float dl[GBW_MAXE];
    gate_dlogits(gates + s * E, dgates + s * E, E, dl);
    for (int ch = lane; ch < (D >> 3); ch += 64) {
        float o[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int e = 0; e < GBW_MAXE; ++e) {
            if (e < E) {
                float w[8];
                EL::load8(wg + (int64_t)e * D + ch * 8, w);
#pragma unroll
                for (int i = 0; i < 8; ++i) o[i] += dl[e] * w[i];
            }
        }
        EL::store8(dx + s * lddx + ch * 8, o);
    }
}
constexpr int GBW_TOK = 128;          // tokens per slice of the d wg partial sums
template <typename T>
__global__ __launch_bounds__(256) void moe_gate_bwd_dw_kernel(const float* __restrict__ gates, const float* __restrict__ dgates, const T* __restrict__ x,
                                                              const T* __restrict__ c, int64_t ld, int64_t S, int D, int E, float* __restrict__ part) {
    // block = 4 waves; blockIdx.x = column block of 512 (8 per lane), blockIdx.y = token slice; each wave takes every 4th token of the slice,
    // the four waves' sums meet in LDS. part[slice][e][D].
    using EL = ElemT<T>;
    __shared__ float red[4][64 * 8];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = blockIdx.x * 512 + lane * 8;
    const int64_t s0 = (int64_t)blockIdx.y * GBW_TOK;
    float acc[GBW_MAXE][8];
#pragma unroll
    for (int e = 0; e < GBW_MAXE; ++e)
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[e][i] = 0.f;
    if (col < D) {
        for (int t = wave; t < GBW_TOK; t += 4) {
            const int64_t s = s0 + t;
            if (s >= S) break;
            float dl[GBW_MAXE], a[8], b[8];
            gate_dlogits(gates + s * E, dgates + s * E, E, dl);
            EL::load8(x + s * ld + col, a);
            EL::load8(c + s * ld + col, b);
#pragma unroll
            for (int i = 0; i < 8; ++i) a[i] = EL::rnd(a[i] + b[i]);
#pragma unroll
            for (int e = 0; e < GBW_MAXE; ++e)
                if (e < E) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) acc[e][i] += dl[e] * a[i];
                }
        }
    }
#pragma unroll
    for (int e = 0; e < GBW_MAXE; ++e) {
        if (e < E) {                                   // E is uniform: every thread reaches the barriers
#pragma unroll
            for (int i = 0; i < 8; ++i) red[wave][lane * 8 + i] = acc[e][i];
            __syncthreads();
            if (wave == 0 && col < D) {
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    part[((int64_t)blockIdx.y * E + e) * D + col + i] = (red[0][lane * 8 + i] + red[1][lane * 8 + i]) + (red[2][lane * 8 + i] + red[3][lane * 8 + i]);
            }
            __syncthreads();
        }
    }
}

template <typename T>
int gelu_impl(const void* x, const void* dy, void* out, int64_t n, ug_stream_t stream) {
    if (n == 0) return UG_OK;
    UG_REQUIRE(x && out && n > 0, UG_ERR_BAD_SHAPE, "ug_gelu_tanh: bad arguments");
    const bool vec = n % 8 == 0 && ug_aligned(x, 16) && ug_aligned(out, 16) && (!dy || ug_aligned(dy, 16));
    const dim3 grid(grid1d(n, 256 * 8 * (vec ? 2 : 1)));
    if (dy) {
        if (vec) hipLaunchKernelGGL((gelu_bwd_kernel<T, true>), grid, dim3(256), 0, (hipStream_t)stream, (const T*)x, (const T*)dy, (T*)out, n);
        else hipLaunchKernelGGL((gelu_bwd_kernel<T, false>), grid, dim3(256), 0, (hipStream_t)stream, (const T*)x, (const T*)dy, (T*)out, n);
    } else {
        if (vec) hipLaunchKernelGGL((gelu_kernel<T, true>), grid, dim3(256), 0, (hipStream_t)stream, (const T*)x, (T*)out, n);
        else hipLaunchKernelGGL((gelu_kernel<T, false>), grid, dim3(256), 0, (hipStream_t)stream, (const T*)x, (T*)out, n);
    }
    UG_CHECK_LAUNCH("ug_gelu_tanh");
    return UG_OK;
}
template <typename T>
int moe_gate_bwd_impl(const float* gates, const float* dgates, const void* x, const void* c, int64_t ld, const void* wg, int64_t S, int64_t D, int32_t E,
                      void* dx, int64_t lddx, float* dwg_partials, ug_stream_t stream) {
    if (S == 0) return UG_OK;
    UG_REQUIRE(gates && dgates && x && c && wg && dx && dwg_partials && S > 0 && D > 0 && E >= 1 && E <= GBW_MAXE && ld >= D && lddx >= D, UG_ERR_BAD_SHAPE,
               "ug_moe_gate_bwd: bad arguments (E <= %d)", GBW_MAXE);
    UG_REQUIRE(D % 8 == 0 && ld % 8 == 0 && lddx % 8 == 0 && ug_aligned(x, 16) && ug_aligned(c, 16) && ug_aligned(wg, 16) && ug_aligned(dx, 16), UG_ERR_BAD_ALIGN,
               "ug_moe_gate_bwd: 16-byte alignment required");
    hipLaunchKernelGGL(moe_gate_bwd_dx_kernel<T>, dim3((unsigned)((S + 3) / 4)), dim3(256), 0, (hipStream_t)stream, gates, dgates, (const T*)wg, S, (int)D, (int)E,
                       (T*)dx, lddx);
    hipLaunchKernelGGL(moe_gate_bwd_dw_kernel<T>, dim3((unsigned)((D + 511) / 512), (unsigned)((S + GBW_TOK - 1) / GBW_TOK)), dim3(256), 0, (hipStream_t)stream, gates,
                       dgates, (const T*)x, (const T*)c, ld, S, (int)D, (int)E, dwg_partials);
    UG_CHECK_LAUNCH("ug_moe_gate_bwd");
    return UG_OK;
}
template <typename T>
int gate_residual_impl(const void* x, int64_t ldx, const void* a, int64_t lda, const void* gate, int64_t gate_ld, int64_t rows_per_sample, void* y, int64_t ldy,
                       int64_t rows, int64_t D, ug_stream_t stream) {
    if (rows == 0) return UG_OK;
    UG_REQUIRE(a && gate && y && rows > 0 && D > 0 && rows_per_sample > 0 && lda >= D && ldy >= D && gate_ld >= D && (!x || ldx >= D), UG_ERR_BAD_SHAPE,
               "ug_gate_residual: bad arguments");
    UG_REQUIRE(D % 8 == 0 && lda % 8 == 0 && ldy % 8 == 0 && gate_ld % 8 == 0 && (!x || ldx % 8 == 0) && ug_aligned(a, 16) && ug_aligned(gate, 16) &&
               ug_aligned(y, 16) && (!x || ug_aligned(x, 16)), UG_ERR_BAD_ALIGN, "ug_gate_residual: 16-byte alignment / multiples of 8 required");
    hipLaunchKernelGGL(gate_residual_kernel<T>, dim3(grid1d(rows * (D / 8), 256)), dim3(256), 0, (hipStream_t)stream, (const T*)x, ldx, (const T*)a, lda,
                       (const T*)gate, gate_ld, rows_per_sample, (T*)y, ldy, rows, (int)(D / 8));
    UG_CHECK_LAUNCH("ug_gate_residual");
    return UG_OK;
}
}  // namespace
extern "C" int64_t ug_adaln_modulate_bwd_partials(int64_t rows, int64_t rows_per_sample) {
    if (rows <= 0 || rows_per_sample <= 0) return 0;
    const int64_t samples = rows / rows_per_sample, by_rows = (rows_per_sample + 3) / 4;
    int64_t p = 1024 / (samples > 0 ? samples : 1);
    if (p > by_rows) p = by_rows;
    return p < 1 ? 1 : p;
}
namespace {
template <typename T>
int adaln_bwd_impl(const void* x, int64_t ldx, const void* dy, int64_t lddy, const void* scale, int64_t mod_ld, int64_t rows_per_sample, void* dx,
                   int64_t lddx, void* part, int64_t rows, int64_t D, float eps, ug_stream_t stream) {
    if (rows == 0) return UG_OK;
    UG_REQUIRE(x && dy && scale && dx && part && rows > 0 && D > 0 && rows_per_sample > 0 && rows % rows_per_sample == 0 && ldx >= D && lddy >= D && lddx >= D &&
               mod_ld >= D, UG_ERR_BAD_SHAPE, "ug_adaln_modulate_bwd: bad arguments");
    constexpr int EA = 16;
    UG_REQUIRE(D % 8 == 0 && D <= 4096 && ldx % 8 == 0 && lddy % 8 == 0 && lddx % 8 == 0 && mod_ld % 8 == 0 && ug_aligned(x, EA) && ug_aligned(dy, EA) &&
               ug_aligned(dx, EA) && ug_aligned(scale, EA) && ug_aligned(part, 4), UG_ERR_BAD_ALIGN,
               "ug_adaln_modulate_bwd: D <= 4096, D and leading dimensions multiples of 8, 16-byte aligned bases");
    const int64_t samples = rows / rows_per_sample;
    const dim3 grid((unsigned)ug_adaln_modulate_bwd_partials(rows, rows_per_sample), (unsigned)(samples < 65535 ? samples : 65535));
    const int waves = (int)((D + 511) / 512);                      // <= 8: D <= 4096
    hipLaunchKernelGGL(adaln_bwd_kernel<T>, grid, dim3(64 * waves), 0, (hipStream_t)stream, (const T*)x, ldx, (const T*)dy, lddy, (const T*)scale, mod_ld,
                       rows_per_sample, (T*)dx, lddx, (float*)part, (int)D, eps, samples);
    UG_CHECK_LAUNCH("ug_adaln_modulate_bwd");
    return UG_OK;
}
}  // namespace
extern "C" int64_t ug_qk_rmsnorm_rope_bwd_partials(int64_t rows, int32_t heads) {
    const int64_t blocks = (rows * (int64_t)heads + 3) / 4;
    return blocks < 1 ? 1 : (blocks > 2048 ? 2048 : blocks);
}
namespace {
template <typename T>
int qk_bwd_impl(const void* x, int64_t ldx, const void* dy, int64_t lddy, void* dx, int64_t lddx, void* dwx, const void* w, const float* cos_tab,
                const float* sin_tab, int64_t rows, int64_t rows_per_batch, int64_t pos_offset, int32_t heads, int32_t dh, float eps, ug_stream_t stream) {
    if (rows == 0) return UG_OK;
    UG_REQUIRE(x && dy && dx && rows > 0 && heads > 0 && dh > 0 && dh % 2 == 0 && rows_per_batch > 0 && (!w || dwx) && (cos_tab == nullptr) == (sin_tab == nullptr),
               UG_ERR_BAD_SHAPE, "ug_qk_rmsnorm_rope_bwd: bad arguments");
    UG_REQUIRE(dh <= 128 * QKB_MAXP && ldx % 2 == 0 && lddy % 2 == 0 && lddx % 2 == 0 && ug_aligned(x, 2 * sizeof(T)) && ug_aligned(dy, 2 * sizeof(T)) &&
               ug_aligned(dx, 2 * sizeof(T)), UG_ERR_UNSUPPORTED, "ug_qk_rmsnorm_rope_bwd: head width <= 256, even leading dimensions, pair-aligned bases");
    const int64_t nvec = rows * heads;
    const dim3 grid((unsigned)ug_qk_rmsnorm_rope_bwd_partials(rows, heads));
    const bool vec8 = (dh == 64 || dh == 128 || dh == 256) && ldx % 8 == 0 && lddy % 8 == 0 && lddx % 8 == 0 && ug_aligned(x, 16) && ug_aligned(dy, 16) &&
                      ug_aligned(dx, 16) && (!w || ug_aligned(w, 2)) && (!cos_tab || (ug_aligned(cos_tab, 16) && ug_aligned(sin_tab, 16)));
#define UG_QKB8(LPVV)                                                                                                                            \
    hipLaunchKernelGGL((qk_bwd8_kernel<T, LPVV>), grid, dim3(256), 0, (hipStream_t)stream, (const T*)x, ldx, (const T*)dy, lddy, (T*)dx, lddx, (float*)dwx, \
                       (const T*)w, cos_tab, sin_tab, rows_per_batch, pos_offset, nvec, (int)heads, eps)
    if (vec8 && dh == 128) UG_QKB8(16);
    else if (vec8 && dh == 64) UG_QKB8(8);
    else if (vec8 && dh == 256) UG_QKB8(32);
    else
        hipLaunchKernelGGL(qk_bwd_kernel<T>, grid, dim3(256), 0, (hipStream_t)stream, (const T*)x, ldx, (const T*)dy, lddy, (T*)dx, lddx, (float*)dwx, (const T*)w,
                           cos_tab, sin_tab, rows_per_batch, pos_offset, nvec, (int)heads, (int)dh, eps);
#undef UG_QKB8
    UG_CHECK_LAUNCH("ug_qk_rmsnorm_rope_bwd");
    return UG_OK;
}
template <typename T>
int attn_prob_impl(const float* S, int64_t ld_s, const float* lse, void* P, int64_t ld_p, int64_t rows, int64_t cols, int64_t valid_cols, float scale,
                   ug_stream_t stream) {
    if (rows == 0) return UG_OK;
    UG_REQUIRE(S && lse && P && rows > 0 && cols > 0 && valid_cols > 0 && valid_cols <= cols && ld_s >= cols && ld_p >= cols, UG_ERR_BAD_SHAPE,
               "ug_attn_prob: bad arguments");
    hipLaunchKernelGGL(attn_prob_kernel<T>, dim3(grid1d(rows * cols, 256 * 8)), dim3(256), 0, (hipStream_t)stream, S, ld_s, lse, (T*)P, ld_p, rows, (int)cols,
                       (int)valid_cols, scale);
    UG_CHECK_LAUNCH("ug_attn_prob");
    return UG_OK;
}
template <typename T>
int attn_dscore_impl(const void* P, int64_t ld_p, const float* dP, int64_t ld_dp, const float* delta, void* dS, int64_t ld_ds, int64_t rows, int64_t cols,
                     float scale, ug_stream_t stream) {
    if (rows == 0) return UG_OK;
    UG_REQUIRE(P && dP && delta && dS && rows > 0 && cols > 0 && ld_p >= cols && ld_dp >= cols && ld_ds >= cols, UG_ERR_BAD_SHAPE, "ug_attn_dscore: bad arguments");
    hipLaunchKernelGGL(attn_dscore_kernel<T>, dim3(grid1d(rows * cols, 256 * 8)), dim3(256), 0, (hipStream_t)stream, (const T*)P, ld_p, dP, ld_dp, delta, (T*)dS,
                       ld_ds, rows, (int)cols, scale);
    UG_CHECK_LAUNCH("ug_attn_dscore");
    return UG_OK;
}
template <typename T>
int rowdot_impl(const void* a, int64_t lda, const void* b, int64_t ldb, float* out, int64_t rows, int64_t groups, int64_t cols, ug_stream_t stream) {
    if (rows == 0) return UG_OK;
    UG_REQUIRE(a && b && out && rows > 0 && groups > 0 && cols > 0 && lda >= groups * cols && ldb >= groups * cols, UG_ERR_BAD_SHAPE, "ug_rowdot: bad arguments");
    hipLaunchKernelGGL(rowdot_kernel<T>, dim3((unsigned)((rows * groups + 3) / 4)), dim3(256), 0, (hipStream_t)stream, (const T*)a, lda, (const T*)b, ldb, out,
                       rows, (int)groups, (int)cols);
    UG_CHECK_LAUNCH("ug_rowdot");
    return UG_OK;
}

}  // namespace

#define UG_TWINS(NAME, IMPL, PARAMS, ARGS)                                              \
    extern "C" int NAME PARAMS { return IMPL<bf16_t> ARGS; }                            \
    extern "C" int NAME##_f32 PARAMS { return IMPL<float> ARGS; }

UG_TWINS(ug_transpose, transpose_impl,
         (const void* src, int64_t ld_src, int64_t src_bstride, void* dst, int64_t ld_dst, int64_t dst_bstride, int64_t batch, int64_t rows, int64_t cols,
          int64_t rows_pad, ug_stream_t stream),
         (src, ld_src, src_bstride, dst, ld_dst, dst_bstride, batch, rows, cols, rows_pad, stream))
UG_TWINS(ug_colsum, colsum_impl,
         (const void* a, int64_t lda, const void* b, int64_t ldb, void* out, int64_t ldo, int64_t rows, int64_t cols, int64_t rows_per_group, float alpha,
          void* workspace, int64_t workspace_bytes, ug_stream_t stream),
         (a, lda, b, ldb, out, ldo, rows, cols, rows_per_group, alpha, workspace, workspace_bytes, stream))
extern "C" int64_t ug_colsum_workspace_bytes(int64_t rows, int64_t cols, int64_t rows_per_group) {
    if (rows <= 0 || cols <= 0 || rows_per_group <= 0) return 0;
    return (rows / rows_per_group) * ((rows_per_group + COLSUM_ROWS - 1) / COLSUM_ROWS) * cols * (int64_t)sizeof(float);
}
UG_TWINS(ug_gate_residual, gate_residual_impl,
         (const void* x, int64_t ldx, const void* a, int64_t lda, const void* gate, int64_t gate_ld, int64_t rows_per_sample, void* y, int64_t ldy, int64_t rows,
          int64_t D, ug_stream_t stream),
         (x, ldx, a, lda, gate, gate_ld, rows_per_sample, y, ldy, rows, D, stream))
UG_TWINS(ug_moe_gate_bwd, moe_gate_bwd_impl,
         (const float* gates, const float* dgates, const void* x, const void* c, int64_t ld, const void* wg, int64_t S, int64_t D, int32_t E, void* dx,
          int64_t lddx, float* dwg_partials, ug_stream_t stream),
         (gates, dgates, x, c, ld, wg, S, D, E, dx, lddx, dwg_partials, stream))
extern "C" int64_t ug_moe_gate_bwd_slices(int64_t S) { return S <= 0 ? 0 : (S + GBW_TOK - 1) / GBW_TOK; }
UG_TWINS(ug_gelu_tanh, gelu_impl, (const void* x, void* y, int64_t n, ug_stream_t stream), (x, nullptr, y, n, stream))
UG_TWINS(ug_gelu_tanh_bwd, gelu_impl, (const void* x, const void* dy, void* dx, int64_t n, ug_stream_t stream), (x, dy, dx, n, stream))
UG_TWINS(ug_adaln_modulate_bwd, adaln_bwd_impl,
         (const void* x, int64_t ldx, const void* dy, int64_t lddy, const void* scale, int64_t mod_ld, int64_t rows_per_sample, void* dx, int64_t lddx,
          void* partials, int64_t rows, int64_t D, float eps, ug_stream_t stream),
         (x, ldx, dy, lddy, scale, mod_ld, rows_per_sample, dx, lddx, partials, rows, D, eps, stream))
UG_TWINS(ug_qk_rmsnorm_rope_bwd, qk_bwd_impl,
         (const void* x, int64_t ldx, const void* dy, int64_t lddy, void* dx, int64_t lddx, void* dwx, const void* w, const float* cos_tab,
          const float* sin_tab, int64_t rows, int64_t rows_per_batch, int64_t pos_offset, int32_t heads, int32_t dh, float eps, ug_stream_t stream),
         (x, ldx, dy, lddy, dx, lddx, dwx, w, cos_tab, sin_tab, rows, rows_per_batch, pos_offset, heads, dh, eps, stream))
UG_TWINS(ug_attn_prob, attn_prob_impl,
         (const float* S, int64_t ld_s, const float* lse, void* P, int64_t ld_p, int64_t rows, int64_t cols, int64_t valid_cols, float scale,
          ug_stream_t stream),
         (S, ld_s, lse, P, ld_p, rows, cols, valid_cols, scale, stream))
UG_TWINS(ug_attn_dscore, attn_dscore_impl,
         (const void* P, int64_t ld_p, const float* dP, int64_t ld_dp, const float* delta, void* dS, int64_t ld_ds, int64_t rows, int64_t cols, float scale,
          ug_stream_t stream),
         (P, ld_p, dP, ld_dp, delta, dS, ld_ds, rows, cols, scale, stream))
UG_TWINS(ug_rowdot, rowdot_impl,
         (const void* a, int64_t lda, const void* b, int64_t ldb, float* out, int64_t rows, int64_t groups, int64_t cols, ug_stream_t stream),
         (a, lda, b, ldb, out, rows, groups, cols, stream))

extern "C" int ug_row_lse(const float* S, int64_t ld, float* lse, int64_t rows, int64_t cols, float scale, ug_stream_t stream) {
    if (rows == 0) return UG_OK;
    UG_REQUIRE(S && lse && rows > 0 && cols > 0 && ld >= cols && rows < (1ll << 31), UG_ERR_BAD_SHAPE, "ug_row_lse: bad arguments");
    hipLaunchKernelGGL(row_lse_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, S, ld, lse, (int)cols, scale);
    UG_CHECK_LAUNCH("ug_row_lse");
    return UG_OK;
}
