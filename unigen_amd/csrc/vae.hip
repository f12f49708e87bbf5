// AutoencoderKL kernels (the row either side of the hot path): vae.encode of the condition image and vae.decode of
// the denoised latents (reference: src/UniGenPipeline.py:635-636, 797-798 -> diffusers 0.32.2 AutoencoderKL / Encoder / Decoder /
// ResnetBlock2D / Upsample2D / Downsample2D / Attention). Activations are NHWC ([B, H, W, C] = pixel rows x channels), which makes every
// convolution an implicit GEMM over MFMA and every 1x1 convolution / attention projection a plain ug_gemm_bf16.
//
//   ug_conv2d_nhwc      3x3 / strided / padded convolution as an IMPLICIT GEMM: M = B Ho Wo output pixels, N = Cout, K = KH KW Cin. The A
//                       operand is gathered by the LDS-DMA itself: each lane's source address is its pixel's row of the tap being staged
//                       (or a zero page for padding), so no im2col buffer exists; nearest-2x upsampling (Upsample2D) is folded into the
//                       gather (source pixel = (y >> 1, x >> 1)), the asymmetric (0, 1, 0, 1) padding of Downsample2D is just pad_t = pad_l = 0.
//                       128 x 128 x 64 tiles, 4 waves, v_mfma_f32_16x16x32_bf16, double-buffered LDS; bias and residual add fused. Round 3: with
//                       Cout and B Ho Wo multiples of 256 and Cin / 64 a power of two the SAME convolution runs on the 256^2 8-phase GEMM
//                       kernel (gemm.hip, CONV: per-tap A gather; bit-identical); a 256 x 128 three-stage form of this kernel is kept as a switch.
//   ug_groupnorm_nhwc   GroupNorm(eps) [+ SiLU] in two deterministic passes (per-chunk partial moments, fixed-order combine in fp64); 16-byte
//                       kernels for the AutoencoderKL widths (C = 128 / 256 / 512) at the HBM roofline.
//   ug_softmax_rows     row softmax of fp32 scores -> probabilities (the mid-block attention has ONE head of dim C = 512: it runs as
//                       scores = q k^T (ug_gemm, fp32 out), this kernel, then P v (ug_gemm): a 512-wide head does not fit the flash kernel's
//                       registers); rows of 1024 n <= 16384 scores are read once and kept in registers.
//   ug_nchw_to_nhwc / ug_nhwc_to_nchw   boundary layout changes (+ channel zero-padding to the conv's K granularity, + the latent scale / shift).
//   ug_vae_sample       DiagonalGaussianDistribution.sample() + (z - shift) * scale.
// Every kernel is a template over the element type: bf16 = product, fp32 = verification twin (the fp32 convolution is a direct loop).
#include "ug_common.h"
#include "gemm_epilogue.h"
#include <algorithm>

namespace {

constexpr int CBN = 128, CBK = 64;

struct ConvP {
    const bf16_t* x; const bf16_t* w; const bf16_t* bias; const bf16_t* R; bf16_t* out; const bf16_t* zero;
    int B, H, W, Cin, Ho, Wo, Cout, KH, KW, stride, pad_t, pad_l, up;
};

// Tile (64 NWM) x 128 x 64, 2 NWM waves (NWM x 2, each 64 x 64), a ring of NST K-tile stages in LDS, ONE barrier per K-tile: a stage is
// waited for with a counted vmcnt (the NST - 2 younger stages stay in flight), the barrier that follows both publishes it and retires
// every read of the stage re-filled right after. <2, 2>: 128 x 128, 64 KiB, two workgroups per CU (rounds 1-2; ragged and small shapes);
// <4, 3>: 256 x 128, 144 KiB, one workgroup per CU, two K-tiles ahead - 0.75x the L2 -> LDS bytes per FLOP of the 128^2 tile (UG_CONV_BIG=1).
// MEASURED (round 3, profiles/r03q_vae_bench.log): bit-identical and NOT faster - decode 14.05-14.09 ms with it against 13.89 ms without: the
// convolutions left on this kernel (Cout = 128: K = 9 x 128 = 18 K-tiles) are short tiles whose first-DMA latency and C stores only hide
// under ANOTHER workgroup of the same CU, which the 64 KiB form has and the 144 KiB form does not. Off by default.
template <int NWM, int NST>
__global__ __launch_bounds__(128 * NWM, 2) void conv2d_nhwc_kernel(const ConvP p) {
    constexpr int NW = 2 * NWM, CBM = 64 * NWM, APER = 4, WPER = CBN / (NW * 8), PER = APER + WPER;
    constexpr int ATILE = CBM * CBK * 2, WTILE = CBN * CBK * 2, STAGE = ATILE + WTILE;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int M = p.B * p.Ho * p.Wo, N = p.Cout;
    const int nM = (M + CBM - 1) / CBM, nN = (N + CBN - 1) / CBN;
    const TileCoord tc = tile_of_block(blockIdx.x, nM, nN);
    const int m0 = tc.tm * CBM, n0 = tc.tn * CBN;
    const int ldw = p.KH * p.KW * p.Cin;
    const int Hv = p.H << p.up, Wv = p.W << p.up;

    // staging rows of this lane: output pixel (b, oy, ox) of A rows wave*32 + i*8 + (lane >> 3); W rows wave*(8 WPER) + i*8 + (lane >> 3)
    int pb[APER], py[APER], px[APER];
    const bf16_t* wsrc[WPER];
    int chunk[APER], wchunk[WPER];
#pragma unroll
    for (int i = 0; i < APER; ++i) {
        const int row = wave * 32 + i * 8 + (lane >> 3);
        chunk[i] = ((lane & 7) ^ (row & 7)) * 8;
        int m = m0 + row; if (m > M - 1) m = M - 1;
        const int b = m / (p.Ho * p.Wo), r = m - b * (p.Ho * p.Wo);
        pb[i] = b; py[i] = r / p.Wo; px[i] = r - py[i] * p.Wo;
    }
#pragma unroll
    for (int i = 0; i < WPER; ++i) {
        const int row = wave * (8 * WPER) + i * 8 + (lane >> 3);
        wchunk[i] = ((lane & 7) ^ (row & 7)) * 8;
        int n = n0 + row; if (n > N - 1) n = N - 1;
        wsrc[i] = p.w + (int64_t)n * ldw + wchunk[i];
    }
    const int kt_per_tap = p.Cin / CBK;
    const int nk = p.KH * p.KW * kt_per_tap;
    const bf16_t* asrc[APER];
    bool aok[APER];
    auto tap_sources = [&](int tap) __attribute__((always_inline)) {        // the pixel each staging row reads for this tap (virtual, i.e. upsampled, coordinates -> stored ones)
        const int ky = tap / p.KW, kx = tap - ky * p.KW;
#pragma unroll
        for (int i = 0; i < APER; ++i) {
            const int yv = py[i] * p.stride + ky - p.pad_t, xv = px[i] * p.stride + kx - p.pad_l;
            aok[i] = yv >= 0 && yv < Hv && xv >= 0 && xv < Wv;
            const int sy = aok[i] ? yv >> p.up : 0, sx = aok[i] ? xv >> p.up : 0;
            asrc[i] = p.x + (((int64_t)pb[i] * p.H + sy) * p.W + sx) * p.Cin + chunk[i];
        }
    };
    auto stage = [&](int buf, int kt) __attribute__((always_inline)) {
        const int tap = kt / kt_per_tap, ci = (kt - tap * kt_per_tap) * CBK;
        if (ci == 0) tap_sources(tap);                       // wave-uniform
        unsigned char* Abuf = smem + buf * STAGE;
        unsigned char* Wbuf = Abuf + ATILE;
#pragma unroll
        for (int i = 0; i < APER; ++i)
            glds16(aok[i] ? asrc[i] + ci : p.zero + chunk[i], Abuf + (wave * 32 + i * 8) * 128);     // padding lanes read the 128-byte zero page
#pragma unroll
        for (int i = 0; i < WPER; ++i) glds16(wsrc[i] + (int64_t)kt * CBK, Wbuf + (wave * (8 * WPER) + i * 8) * 128);
    };
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int frow = lane & 15, fch = lane >> 4, fsw = lane & 7;
#pragma unroll
    for (int st = 0; st < NST - 1; ++st)
        if (st < nk) stage(st, st);
    int cur = 0;
    for (int kt = 0; kt < nk; ++kt) {
        // stage kt landed (this wave's share): at most NST - 2 younger stages stay in flight
        if (NST >= 3 && kt + 1 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NST - 2) * PER) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0);
        if (kt + NST - 1 < nk) stage(cur == 0 ? NST - 1 : cur - 1, kt + NST - 1);     // the stage read in the previous iteration
        const unsigned char* Abuf = smem + cur * STAGE;
        const unsigned char* Wbuf = Abuf + ATILE;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            bf16x8 af[4], wf[4];
            const int choff = (((s * 4 + fch) ^ fsw) << 4);
#pragma unroll
            for (int i = 0; i < 4; ++i) af[i] = *(const bf16x8*)(Abuf + (wr * 64 + i * 16 + frow) * 128 + choff);
#pragma unroll
            for (int j = 0; j < 4; ++j) wf[j] = *(const bf16x8*)(Wbuf + (wc * 64 + j * 16 + frow) * 128 + choff);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], af[i], acc[i][j], 0, 0, 0);
        }
        cur = cur + 1 == NST ? 0 : cur + 1;
    }
    // epilogue: lane holds D[n = 4 consecutive][m]: out = R + bf16(acc + bias)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + wr * 64 + i * 16 + (lane & 15);
        if (m >= M) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + wc * 64 + j * 16 + (lane >> 4) * 4;
            if (n >= N) continue;
            float bv[4];
            load_bias4(p.bias, n, bv);
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = rbf(acc[i][j][r] + bv[r]);
            if (p.R) {
                const u32x2 r2 = *(const u32x2*)(p.R + (int64_t)m * N + n);
                v[0] += bflo(r2.x); v[1] += bfhi(r2.x); v[2] += bflo(r2.y); v[3] += bfhi(r2.y);
            }
            u32x2 o; o.x = pack2bf(v[0], v[1]); o.y = pack2bf(v[2], v[3]);
            *(u32x2*)(p.out + (int64_t)m * N + n) = o;
        }
    }
}

// fp32 verification twin: one thread per output element, taps x channels in the same (tap, ci) order
struct ConvPF { const float* x; const float* w; const float* bias; const float* R; float* out; int B, H, W, Cin, Ho, Wo, Cout, KH, KW, stride, pad_t, pad_l, up; };
__global__ void conv2d_nhwc_f32_kernel(const ConvPF p) {
    const int64_t total = (int64_t)p.B * p.Ho * p.Wo * p.Cout;
    const int Hv = p.H << p.up, Wv = p.W << p.up;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int n = (int)(idx % p.Cout);
        const int64_t m = idx / p.Cout;
        const int b = (int)(m / (p.Ho * p.Wo)), r = (int)(m - (int64_t)b * p.Ho * p.Wo);
        const int oy = r / p.Wo, ox = r - oy * p.Wo;
        float acc = 0.f;
        for (int ky = 0; ky < p.KH; ++ky)
            for (int kx = 0; kx < p.KW; ++kx) {
                const int yv = oy * p.stride + ky - p.pad_t, xv = ox * p.stride + kx - p.pad_l;
                if (yv < 0 || yv >= Hv || xv < 0 || xv >= Wv) continue;
                const float* xp = p.x + (((int64_t)b * p.H + (yv >> p.up)) * p.W + (xv >> p.up)) * p.Cin;
                const float* wp = p.w + ((int64_t)n * p.KH * p.KW + ky * p.KW + kx) * p.Cin;
                for (int c = 0; c < p.Cin; ++c) acc = fmaf(xp[c], wp[c], acc);
            }
        float v = acc + (p.bias ? p.bias[n] : 0.f);
        if (p.R) v += p.R[m * p.Cout + n];
        p.out[m * p.Cout + n] = v;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// GroupNorm (+ SiLU), NHWC. Pass 1: block (b, chunk of GN_ROWS pixels) -> per-group (sum, sum of squares) partials, fp32 per thread, fp64 at
// the block level; pass 2 (inside the apply kernel's prologue): fixed-order sum of a sample's partials in fp64 -> mean, rstd.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int GN_ROWS = 64;

// pass 1: block (b, chunk of GN_ROWS pixels): thread t sums channel columns t, t + 256, ... over the chunk's rows (consecutive threads read
// consecutive channels), then the first thread of every group adds its group's cg columns: part[b][chunk][g] = (sum, sum of squares)
template <typename T>
__global__ __launch_bounds__(256) void gn_partial_kernel(const T* __restrict__ x, int64_t HW, int C, int G, int nchunks, double* __restrict__ part) {
    __shared__ double red[256][2];
    const int b = blockIdx.y, ch = blockIdx.x;
    const int cg = C / G;
    const int64_t r0 = (int64_t)ch * GN_ROWS, r1 = (r0 + GN_ROWS < HW) ? r0 + GN_ROWS : HW;
    const T* xb = x + ((int64_t)b * HW) * C;
    for (int c0 = 0; c0 < C; c0 += 256) {
        const int c = c0 + threadIdx.x;
        float s = 0.f, q = 0.f;
        if (c < C)
            for (int64_t r = r0; r < r1; ++r) { const float v = ElemT<T>::ld(xb + r * C + c); s += v; q += v * v; }
        red[threadIdx.x][0] = s; red[threadIdx.x][1] = q;
        __syncthreads();
        if (c < C && threadIdx.x % cg == 0) {            // cg divides 256 (host check): a group never straddles two passes
            double ss = 0.0, qq = 0.0;
            for (int k = 0; k < cg; ++k) { ss += red[threadIdx.x + k][0]; qq += red[threadIdx.x + k][1]; }
            double* o = part + (((int64_t)b * nchunks + ch) * G + c / cg) * 2;
            o[0] = ss; o[1] = qq;
        }
        __syncthreads();
    }
}

// pass 2: block (b, g): fixed-order sum of the chunk partials in fp64 -> stats[b][g] = (mean, rstd)
__global__ __launch_bounds__(256) void gn_finalize_kernel(const double* __restrict__ part, int64_t HW, int cg, int G, int nchunks, float eps, float* __restrict__ stats) {
    __shared__ double red[256][2];
    const int b = blockIdx.y, g = blockIdx.x;
    double s = 0.0, q = 0.0;
    for (int ch = threadIdx.x; ch < nchunks; ch += 256) { const double* p = part + (((int64_t)b * nchunks + ch) * G + g) * 2; s += p[0]; q += p[1]; }
    red[threadIdx.x][0] = s; red[threadIdx.x][1] = q;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) { red[threadIdx.x][0] += red[threadIdx.x + o][0]; red[threadIdx.x][1] += red[threadIdx.x + o][1]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const double n = (double)HW * cg;
        const double mean = red[0][0] / n;
        double var = red[0][1] / n - mean * mean; if (var < 0.0) var = 0.0;
        stats[((int64_t)b * G + g) * 2] = (float)mean;
        stats[((int64_t)b * G + g) * 2 + 1] = (float)(1.0 / sqrt(var + (double)eps));
    }
}

template <typename T>
__global__ __launch_bounds__(256) void gn_apply_kernel(const T* __restrict__ x, const T* __restrict__ gamma, const T* __restrict__ beta, const float* __restrict__ stats,
                                                       int64_t HW, int C, int G, int silu, T* __restrict__ out) {
    using E = ElemT<T>;
    const int b = blockIdx.y;
    const int cg = C / G;
    const float* st = stats + (int64_t)b * G * 2;
    const int cchunks = C >> 3;
    const int64_t r0 = (int64_t)blockIdx.x * GN_ROWS, r1 = (r0 + GN_ROWS < HW) ? r0 + GN_ROWS : HW;
    const int64_t total = (r1 - r0) * cchunks;
    for (int64_t i = threadIdx.x; i < total; i += 256) {
        const int64_t r = r0 + i / cchunks; const int c8 = (int)(i % cchunks) * 8;
        float v[8], ga[8], be[8];
        E::load8(x + ((int64_t)b * HW + r) * C + c8, v);
        E::load8(gamma + c8, ga); E::load8(beta + c8, be);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int g = (c8 + e) / cg;
            float y = E::rnd((v[e] - st[2 * g]) * st[2 * g + 1] * ga[e] + be[e]);            // F.group_norm output (one rounding)
            if (silu) y = E::rnd(y / (1.0f + expf(-y)));                                      // F.silu on that tensor
            v[e] = y;
        }
        E::store8(out + ((int64_t)b * HW + r) * C + c8, v);
    }
}

// bf16 product kernels for C = 8 * CCH, CCH in {16, 32, 64} (the AutoencoderKL widths 128 / 256 / 512): thread -> one 16-byte channel chunk
// (constant per thread: gamma / beta / statistics are loaded once) x every (256 / CCH)-th pixel row of a GN_FAST_ROWS-row slab, four loads in
// flight. Same partial layout and fp64 fixed-order combine as the generic kernels. SiLU as y * rcp(1 + 2^(-y log2 e)) (v_exp_f32 / v_rcp_f32,
// 1 ulp each, against expf + IEEE division: the bf16 result differs in the last place on a few elements per million).
// Round 3: the generic pair ran at ~40 % of the HBM roofline (2-byte loads of one channel column per thread in the
// statistics pass, expf + a division + two 64-bit div / mod per 8 elements in the apply pass).
constexpr int GN_FAST_ROWS = 256;

template <int CCH>
__global__ __launch_bounds__(256) void gn_partial_fast_kernel(const bf16_t* __restrict__ x, int64_t HW, int G, int nchunks, double* __restrict__ part) {
    constexpr int C = CCH * 8, RPP = 256 / CCH;
    __shared__ float red[256][16];
    const int t = threadIdx.x, cc = t % CCH, ro = t / CCH, b = blockIdx.y;
    const int64_t r0 = (int64_t)blockIdx.x * GN_FAST_ROWS, r1 = (r0 + GN_FAST_ROWS < HW) ? r0 + GN_FAST_ROWS : HW;
    const bf16_t* xp = x + ((int64_t)b * HW) * C + cc * 8;
    float sm[8], sq[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { sm[e] = 0.f; sq[e] = 0.f; }
    auto add = [&](const u32x4& v) __attribute__((always_inline)) {
        const float f[8] = {bflo(v.x), bfhi(v.x), bflo(v.y), bfhi(v.y), bflo(v.z), bfhi(v.z), bflo(v.w), bfhi(v.w)};
#pragma unroll
        for (int e = 0; e < 8; ++e) { sm[e] += f[e]; sq[e] += f[e] * f[e]; }
    };
    int64_t r = r0 + ro;
    for (; r + 3 * RPP < r1; r += 4 * RPP) {
        u32x4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = *(const u32x4*)(xp + (r + u * RPP) * C);
#pragma unroll
        for (int u = 0; u < 4; ++u) add(v[u]);
    }
    for (; r < r1; r += RPP) add(*(const u32x4*)(xp + r * C));
#pragma unroll
    for (int e = 0; e < 8; ++e) { red[t][e] = sm[e]; red[t][8 + e] = sq[e]; }
    __syncthreads();
    if (t < G) {                                     // fixed order: row offsets outer, the group's channels inner
        const int cg = C / G;
        double ss = 0.0, qq = 0.0;
        for (int o = 0; o < RPP; ++o)
            for (int k = 0; k < cg; ++k) {
                const int c = t * cg + k;
                ss += (double)red[o * CCH + (c >> 3)][c & 7]; qq += (double)red[o * CCH + (c >> 3)][8 + (c & 7)];
            }
        double* o2 = part + (((int64_t)b * nchunks + blockIdx.x) * G + t) * 2;
        o2[0] = ss; o2[1] = qq;
    }
}

template <int CCH>
__global__ __launch_bounds__(256) void gn_apply_fast_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ gamma, const bf16_t* __restrict__ beta,
                                                            const float* __restrict__ stats, int64_t HW, int G, int silu, bf16_t* __restrict__ out) {
    constexpr int C = CCH * 8, RPP = 256 / CCH;
    const int t = threadIdx.x, cc = t % CCH, ro = t / CCH, b = blockIdx.y;
    const int cg = C / G;
    const int64_t r0 = (int64_t)blockIdx.x * GN_FAST_ROWS, r1 = (r0 + GN_FAST_ROWS < HW) ? r0 + GN_FAST_ROWS : HW;
    float ga[8], be[8], mu[8], rs[8];
    ElemT<bf16_t>::load8(gamma + cc * 8, ga); ElemT<bf16_t>::load8(beta + cc * 8, be);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int g = (cc * 8 + e) / cg;
        mu[e] = stats[((int64_t)b * G + g) * 2]; rs[e] = stats[((int64_t)b * G + g) * 2 + 1];
    }
    const bf16_t* xp = x + ((int64_t)b * HW) * C + cc * 8;
    bf16_t* op = out + ((int64_t)b * HW) * C + cc * 8;
    auto one = [&](const u32x4& v, bf16_t* dst) __attribute__((always_inline)) {
        float f[8] = {bflo(v.x), bfhi(v.x), bflo(v.y), bfhi(v.y), bflo(v.z), bfhi(v.z), bflo(v.w), bfhi(v.w)};
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float y = rbf((f[e] - mu[e]) * rs[e] * ga[e] + be[e]);                                     // F.group_norm output (one rounding)
            if (silu) y = y * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(y * -1.44269504088896341f));   // F.silu, rounded by the store
            f[e] = y;
        }
        ElemT<bf16_t>::store8(dst, f);
    };
    int64_t r = r0 + ro;
    for (; r + 3 * RPP < r1; r += 4 * RPP) {
        u32x4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = *(const u32x4*)(xp + (r + u * RPP) * C);
#pragma unroll
        for (int u = 0; u < 4; ++u) one(v[u], op + (r + u * RPP) * C);
    }
    for (; r < r1; r += RPP) one(*(const u32x4*)(xp + r * C), op + r * C);
}

// P[r][:] = softmax(scale * S[r][:]) : S fp32 [rows][ld_s], P element type T [rows][ld_p]; one block per row
template <typename T>
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* __restrict__ S, int64_t ld_s, T* __restrict__ P, int64_t ld_p, int cols, float scale) {
    __shared__ float red[256];
    const float* s = S + (int64_t)blockIdx.x * ld_s;
    T* p = P + (int64_t)blockIdx.x * ld_p;
    float mx = -INFINITY;
    for (int c = threadIdx.x; c < cols; c += 256) mx = fmaxf(mx, s[c] * scale);
    red[threadIdx.x] = mx; __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + o]); __syncthreads(); }
    mx = red[0]; __syncthreads();
    float sum = 0.f;
    for (int c = threadIdx.x; c < cols; c += 256) sum += expf(s[c] * scale - mx);
    red[threadIdx.x] = sum; __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
    const float inv = 1.0f / red[0];
    for (int c = threadIdx.x; c < cols; c += 256) ElemT<T>::st(p + c, expf(s[c] * scale - mx) * inv);
}

// bf16 product form for cols == 1024 NV (NV <= 16: the VAE's 4096 / 16384 tokens): the row is read ONCE (16-byte loads, 4 NV values per thread in
// registers), exponentials are taken once (v_exp_f32 on log2(e)-scaled scores), probabilities leave as 8-byte stores. The generic kernel above
// read the row three times and took two expf per element: 1 GB of fp32 scores at 1024^2 in 0.66 ms (2.3 TB/s). UG_SOFTMAX_FAST=0 keeps it.
template <int NV>
__global__ __launch_bounds__(256) void softmax_rows_fast_kernel(const float* __restrict__ S, int64_t ld_s, bf16_t* __restrict__ P, int64_t ld_p, float scale) {
    __shared__ float red[2][4];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const f32x4* s = (const f32x4*)(S + (int64_t)blockIdx.x * ld_s) + t;
    bf16_t* p = P + (int64_t)blockIdx.x * ld_p + 4 * t;
    const float c = scale * 1.44269504088896341f;
    f32x4 v[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) v[j] = __builtin_nontemporal_load(s + 256 * j);
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < NV; ++j) mx = fmaxf(fmaxf(mx, fmaxf(v[j][0], v[j][1])), fmaxf(v[j][2], v[j][3]));
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) mx = fmaxf(mx, __shfl_xor(mx, m, 64));
    if (lane == 0) red[0][wave] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0][0], red[0][1]), fmaxf(red[0][2], red[0][3])) * c;        // scale > 0 (host check): the max commutes with it
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[j][e] = __builtin_amdgcn_exp2f(v[j][e] * c - mx); sum += v[j][e]; }
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) sum += __shfl_xor(sum, m, 64);
    if (lane == 0) red[1][wave] = sum;
    __syncthreads();
    const float inv = 1.0f / (((red[1][0] + red[1][1]) + red[1][2]) + red[1][3]);
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        u32x2 o; o.x = pack2bf(v[j][0] * inv, v[j][1] * inv); o.y = pack2bf(v[j][2] * inv, v[j][3] * inv);
        *(u32x2*)(p + 1024 * j) = o;
    }
}

// out[b][y][x][c] = c < C ? f(in[b][c][y][x]) : 0, f(v) = rnd(rnd(v / div) + add) when div != 0 (the latent un-scaling of the decode side)
template <typename T>
__global__ void nchw_to_nhwc_kernel(const T* __restrict__ in, T* __restrict__ out, int B, int C, int HW, int Cp, float div, float add) {
    const int64_t total = (int64_t)B * HW * Cp;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % Cp); const int64_t r = i / Cp; const int b = (int)(r / HW); const int64_t pix = r - (int64_t)b * HW;
        float v = 0.f;
        if (c < C) {
            v = ElemT<T>::ld(in + ((int64_t)b * C + c) * HW + pix);
            // torch divides a tensor by a Python scalar as a multiplication by the fp32 reciprocal (BinaryDivTrueKernel: "a * (1 / b)")
            if (div != 0.f) v = ElemT<T>::rnd(ElemT<T>::rnd(v * (1.0f / div)) + add);
        }
        ElemT<T>::st(out + i, v);
    }
}
template <typename T>
__global__ void nhwc_to_nchw_kernel(const T* __restrict__ in, T* __restrict__ out, int B, int C, int HW, int Cp) {
    const int64_t total = (int64_t)B * C * HW;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t pix = i % HW; const int64_t r = i / HW; const int c = (int)(r % C); const int b = (int)(r / C);
        ElemT<T>::st(out + i, ElemT<T>::ld(in + ((int64_t)b * HW + pix) * Cp + c));
    }
}

// z[b][l][pix] = ((mean + exp(0.5 clamp(logvar, -30, 20)) * noise) - shift) * scale, moments NHWC [B][HW][Cp] (mean = channels [0, L),
// logvar = [L, 2L)), noise / z NCHW [B][L][HW]; every step rounded as the reference's separate tensor ops do
template <typename T>
__global__ void vae_sample_kernel(const T* __restrict__ mom, int Cp, const T* __restrict__ noise, T* __restrict__ z, int B, int L, int HW, float shift, float scale) {
    using E = ElemT<T>;
    const int64_t total = (int64_t)B * L * HW;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t pix = i % HW; const int64_t r = i / HW; const int l = (int)(r % L); const int b = (int)(r / L);
        const T* m = mom + ((int64_t)b * HW + pix) * Cp;
        const float mean = E::ld(m + l);
        float lv = E::ld(m + L + l); lv = fminf(fmaxf(lv, -30.f), 20.f);
        const float sd = E::rnd(expf(E::rnd(0.5f * lv)));
        float v = E::rnd(mean + E::rnd(sd * E::ld(noise + i)));
        v = E::rnd(E::rnd(v - shift) * scale);
        E::st(z + i, v);
    }
}

template <typename T>
int groupnorm_impl(const void* x, const void* gamma, const void* beta, void* out, void* workspace, int64_t workspace_bytes, int64_t B, int64_t HW, int64_t C,
                   int32_t G, float eps, int32_t silu, ug_stream_t stream) {
    if (B == 0 || HW == 0) return UG_OK;
    UG_REQUIRE(x && gamma && beta && out && workspace && B > 0 && HW > 0 && C > 0 && G > 0 && C % G == 0, UG_ERR_BAD_SHAPE, "ug_groupnorm_nhwc: bad arguments");
    const int cg = (int)(C / G);
    UG_REQUIRE(C % 8 == 0 && cg <= 256 && 256 % cg == 0, UG_ERR_UNSUPPORTED, "ug_groupnorm_nhwc: C %% 8 == 0 and channels per group (%d) must divide 256", cg);
    UG_REQUIRE(ug_aligned(x, 16) && ug_aligned(out, 16) && ug_aligned(gamma, 16) && ug_aligned(beta, 16) && ug_aligned(workspace, 8), UG_ERR_BAD_ALIGN,
               "ug_groupnorm_nhwc: 16-byte alignment required");
    int64_t nchunks = (HW + GN_ROWS - 1) / GN_ROWS;
    const int64_t part_bytes = B * nchunks * G * 2 * (int64_t)sizeof(double);
    UG_REQUIRE(workspace_bytes >= part_bytes + B * G * 2 * (int64_t)sizeof(float), UG_ERR_BAD_SHAPE, "ug_groupnorm_nhwc: workspace too small (ug_groupnorm_workspace_bytes)");
    UG_REQUIRE(nchunks < (1 << 30) && B < 65536, UG_ERR_UNSUPPORTED, "ug_groupnorm_nhwc: too large");
    hipStream_t s = (hipStream_t)stream;
    double* part = (double*)workspace;
    float* stats = (float*)((char*)workspace + part_bytes);
    if constexpr (!ElemT<T>::kF32) {
        // the AutoencoderKL widths: 16-byte kernels over 256-row slabs (fewer, larger partials in the same workspace); UG_GN_FAST=0 keeps the generic pair
        if ((C == 128 || C == 256 || C == 512) && G <= 256 && (cg == 4 || cg == 8 || cg == 16) && ug_env_int("UG_GN_FAST", 1)) {
            nchunks = (HW + GN_FAST_ROWS - 1) / GN_FAST_ROWS;
            const dim3 grid((unsigned)nchunks, (unsigned)B);
#define UG_GN_FAST_LAUNCH(CCH)                                                                                                                                     \
    do {                                                                                                                                                           \
        hipLaunchKernelGGL(gn_partial_fast_kernel<CCH>, grid, dim3(256), 0, s, (const bf16_t*)x, HW, (int)G, (int)nchunks, part);                                  \
        UG_CHECK_LAUNCH("ug_groupnorm_nhwc(partial)");                                                                                                             \
        hipLaunchKernelGGL(gn_finalize_kernel, dim3((unsigned)G, (unsigned)B), dim3(256), 0, s, (const double*)part, HW, cg, (int)G, (int)nchunks, eps, stats);    \
        UG_CHECK_LAUNCH("ug_groupnorm_nhwc(finalize)");                                                                                                            \
        hipLaunchKernelGGL(gn_apply_fast_kernel<CCH>, grid, dim3(256), 0, s, (const bf16_t*)x, (const bf16_t*)gamma, (const bf16_t*)beta, (const float*)stats, HW, \
                           (int)G, (int)silu, (bf16_t*)out);                                                                                                       \
        UG_CHECK_LAUNCH("ug_groupnorm_nhwc(apply)");                                                                                                               \
    } while (0)
            if (C == 128) UG_GN_FAST_LAUNCH(16); else if (C == 256) UG_GN_FAST_LAUNCH(32); else UG_GN_FAST_LAUNCH(64);
#undef UG_GN_FAST_LAUNCH
            return UG_OK;
        }
    }
    hipLaunchKernelGGL(gn_partial_kernel<T>, dim3((unsigned)nchunks, (unsigned)B), dim3(256), 0, s, (const T*)x, HW, (int)C, (int)G, (int)nchunks, part);
    UG_CHECK_LAUNCH("ug_groupnorm_nhwc(partial)");
    hipLaunchKernelGGL(gn_finalize_kernel, dim3((unsigned)G, (unsigned)B), dim3(256), 0, s, (const double*)part, HW, cg, (int)G, (int)nchunks, eps, stats);
    UG_CHECK_LAUNCH("ug_groupnorm_nhwc(finalize)");
    hipLaunchKernelGGL(gn_apply_kernel<T>, dim3((unsigned)nchunks, (unsigned)B), dim3(256), 0, s, (const T*)x, (const T*)gamma, (const T*)beta, (const float*)stats, HW, (int)C,
                       (int)G, (int)silu, (T*)out);
    UG_CHECK_LAUNCH("ug_groupnorm_nhwc(apply)");
    return UG_OK;
}

template <typename T>
int softmax_impl(const float* S, int64_t ld_s, void* P, int64_t ld_p, int64_t rows, int64_t cols, float scale, ug_stream_t stream) {
    if (rows == 0) return UG_OK;
    UG_REQUIRE(S && P && rows > 0 && cols > 0 && ld_s >= cols && ld_p >= cols && rows < (1ll << 31) && cols < (1ll << 31), UG_ERR_BAD_SHAPE, "ug_softmax_rows: bad arguments");
    if constexpr (!ElemT<T>::kF32) {
        if (cols % 1024 == 0 && cols <= 16384 && scale > 0.f && ld_s % 4 == 0 && ld_p % 4 == 0 && ug_aligned(S, 16) && ug_aligned(P, 8) && ug_env_int("UG_SOFTMAX_FAST", 1)) {
            const dim3 grid((unsigned)rows);
            const hipStream_t st = (hipStream_t)stream;
            switch (cols / 1024) {
#define UG_SM_CASE(NV) case NV: hipLaunchKernelGGL(softmax_rows_fast_kernel<NV>, grid, dim3(256), 0, st, S, ld_s, (bf16_t*)P, ld_p, scale); break
                UG_SM_CASE(1); UG_SM_CASE(2); UG_SM_CASE(3); UG_SM_CASE(4); UG_SM_CASE(5); UG_SM_CASE(6); UG_SM_CASE(7); UG_SM_CASE(8);
                UG_SM_CASE(9); UG_SM_CASE(10); UG_SM_CASE(11); UG_SM_CASE(12); UG_SM_CASE(13); UG_SM_CASE(14); UG_SM_CASE(15); UG_SM_CASE(16);
#undef UG_SM_CASE
            }
            UG_CHECK_LAUNCH("ug_softmax_rows");
            return UG_OK;
        }
    }
    hipLaunchKernelGGL(softmax_rows_kernel<T>, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, S, ld_s, (T*)P, ld_p, (int)cols, scale);
    UG_CHECK_LAUNCH("ug_softmax_rows");
    return UG_OK;
}

template <typename T>
int to_nhwc_impl(const void* in, void* out, int64_t B, int64_t C, int64_t HW, int64_t Cp, float div, float add, ug_stream_t stream) {
    if (B == 0) return UG_OK;
    UG_REQUIRE(in && out && B > 0 && C > 0 && HW > 0 && Cp >= C && B * HW * Cp < (1ll << 40) && HW < (1ll << 31), UG_ERR_BAD_SHAPE, "ug_nchw_to_nhwc: bad arguments");
    const int64_t total = B * HW * Cp;
    hipLaunchKernelGGL(nchw_to_nhwc_kernel<T>, dim3((unsigned)std::min<int64_t>((total + 255) / 256, 8192)), dim3(256), 0, (hipStream_t)stream, (const T*)in, (T*)out, (int)B, (int)C,
                       (int)HW, (int)Cp, div, add);
    UG_CHECK_LAUNCH("ug_nchw_to_nhwc");
    return UG_OK;
}
template <typename T>
int to_nchw_impl(const void* in, void* out, int64_t B, int64_t C, int64_t HW, int64_t Cp, ug_stream_t stream) {
    if (B == 0) return UG_OK;
    UG_REQUIRE(in && out && B > 0 && C > 0 && HW > 0 && Cp >= C && B * HW * Cp < (1ll << 40) && HW < (1ll << 31), UG_ERR_BAD_SHAPE, "ug_nhwc_to_nchw: bad arguments");
    const int64_t total = B * C * HW;
    hipLaunchKernelGGL(nhwc_to_nchw_kernel<T>, dim3((unsigned)std::min<int64_t>((total + 255) / 256, 8192)), dim3(256), 0, (hipStream_t)stream, (const T*)in, (T*)out, (int)B, (int)C,
                       (int)HW, (int)Cp);
    UG_CHECK_LAUNCH("ug_nhwc_to_nchw");
    return UG_OK;
}
template <typename T>
int sample_impl(const void* mom, int64_t Cp, const void* noise, void* z, int64_t B, int64_t L, int64_t HW, float shift, float scale, ug_stream_t stream) {
    if (B == 0) return UG_OK;
    UG_REQUIRE(mom && noise && z && B > 0 && L > 0 && HW > 0 && Cp >= 2 * L && HW < (1ll << 31), UG_ERR_BAD_SHAPE, "ug_vae_sample: bad arguments");
    const int64_t total = B * L * HW;
    hipLaunchKernelGGL(vae_sample_kernel<T>, dim3((unsigned)std::min<int64_t>((total + 255) / 256, 8192)), dim3(256), 0, (hipStream_t)stream, (const T*)mom, (int)Cp, (const T*)noise,
                       (T*)z, (int)B, (int)L, (int)HW, shift, scale);
    UG_CHECK_LAUNCH("ug_vae_sample");
    return UG_OK;
}

int conv_check(const ug_conv_desc& d, bool bf16) {
    UG_REQUIRE(d.x && d.w && d.out && d.B > 0 && d.H > 0 && d.W > 0 && d.Cin > 0 && d.Cout > 0 && d.Ho > 0 && d.Wo > 0, UG_ERR_BAD_SHAPE, "ug_conv2d_nhwc: bad arguments");
    UG_REQUIRE(d.KH >= 1 && d.KH <= 7 && d.KW >= 1 && d.KW <= 7 && d.stride >= 1 && d.stride <= 4 && d.pad_t >= 0 && d.pad_l >= 0 && (d.up == 0 || d.up == 1), UG_ERR_UNSUPPORTED,
               "ug_conv2d_nhwc: kernel / stride / padding out of range");
    // the last tap of the last output pixel may reach at most one pixel row / column of bottom / right padding beyond what pad_t / pad_l imply
    const int64_t Hv = d.H << d.up, Wv = d.W << d.up;
    UG_REQUIRE((d.Ho - 1) * d.stride - d.pad_t <= Hv - 1 && (d.Wo - 1) * d.stride - d.pad_l <= Wv - 1, UG_ERR_BAD_SHAPE, "ug_conv2d_nhwc: output larger than the (upsampled) input allows");
    UG_REQUIRE(d.B * d.Ho * d.Wo < (1ll << 31) && d.B * d.H * d.W < (1ll << 31), UG_ERR_UNSUPPORTED, "ug_conv2d_nhwc: pixel counts must fit 31 bits");
    if (bf16) {
        UG_REQUIRE(d.Cin % CBK == 0, UG_ERR_UNSUPPORTED, "ug_conv2d_nhwc: Cin=%lld must be a multiple of %d (zero-pad the channels)", (long long)d.Cin, CBK);
        UG_REQUIRE(d.Cout % 4 == 0, UG_ERR_UNSUPPORTED, "ug_conv2d_nhwc: Cout=%lld must be a multiple of 4", (long long)d.Cout);
        UG_REQUIRE(d.zero_page && ug_aligned(d.zero_page, 16), UG_ERR_BAD_SHAPE, "ug_conv2d_nhwc: zero_page (>= 128 zero bytes, 16-byte aligned) required");
        UG_REQUIRE(ug_aligned(d.x, 16) && ug_aligned(d.w, 16) && ug_aligned(d.out, 8) && (!d.bias || ug_aligned(d.bias, 8)) && (!d.R || ug_aligned(d.R, 8)), UG_ERR_BAD_ALIGN,
                   "ug_conv2d_nhwc: alignment");
    }
    return UG_OK;
}

}  // namespace

extern "C" int64_t ug_groupnorm_workspace_bytes(int64_t B, int64_t HW, int32_t G) {
    const int64_t nchunks = (HW + GN_ROWS - 1) / GN_ROWS;
    return B * nchunks * G * 2 * (int64_t)sizeof(double) + B * G * 2 * (int64_t)sizeof(float) + 64;
}

extern "C" int ug_conv2d_nhwc(const ug_conv_desc* dp, ug_stream_t stream) {
    UG_REQUIRE(dp != nullptr, UG_ERR_BAD_SHAPE, "ug_conv2d_nhwc: null descriptor");
    const ug_conv_desc& d = *dp;
    const int rc = conv_check(d, true);
    if (rc != UG_OK) return rc;
    ConvP p;
    p.x = (const bf16_t*)d.x; p.w = (const bf16_t*)d.w; p.bias = (const bf16_t*)d.bias; p.R = (const bf16_t*)d.R; p.out = (bf16_t*)d.out; p.zero = (const bf16_t*)d.zero_page;
    p.B = (int)d.B; p.H = (int)d.H; p.W = (int)d.W; p.Cin = (int)d.Cin; p.Ho = (int)d.Ho; p.Wo = (int)d.Wo; p.Cout = (int)d.Cout;
    p.KH = d.KH; p.KW = d.KW; p.stride = d.stride; p.pad_t = d.pad_t; p.pad_l = d.pad_l; p.up = d.up;
    const int64_t M = d.B * d.Ho * d.Wo;
    {   // the 256^2 GEMM kernel with a per-tap A gather (gemm.hip, CONV): whole tiles, Cin / 64 a power of two >= 2, at least UG_CONV256_MIN_TILES tiles
        const int64_t ktp = d.Cin / 64, tiles = (M / 256) * (d.Cout / 256);
        if (M % 256 == 0 && d.Cout % 256 == 0 && ktp >= 2 && (ktp & (ktp - 1)) == 0 && d.zero_page_bytes >= 2 * (d.Cin + 64) && d.B < 256 && d.H < 2048 && d.W < 2048 &&
            d.Ho < 4096 && d.Wo < 4096 && ug_aligned(d.out, 16) && (!d.R || ug_aligned(d.R, 16)) && tiles >= ug_env_int("UG_CONV256_MIN_TILES", 192) && ug_env_int("UG_CONV256", 1)) {
            ug_gemm_desc g = {};
            g.A = d.x; g.lda = d.Cin; g.W = d.w; g.ldw = (int64_t)d.KH * d.KW * d.Cin; g.bias = d.bias; g.C = d.out; g.ldc = d.Cout; g.R = d.R; g.ldr = d.Cout;
            g.M = M; g.N = d.Cout; g.K = g.ldw; g.groups = 1; g.alpha = 1.0f; g.epilogue = d.R ? UG_EPI_RES_SCALE : UG_EPI_BIAS;
            UgConvGeom cv;
            cv.zero = (const bf16_t*)d.zero_page; cv.H = (int)d.H; cv.W = (int)d.W; cv.Cin = (int)d.Cin; cv.Ho = (int)d.Ho; cv.Wo = (int)d.Wo; cv.KW = d.KW;
            cv.stride = d.stride; cv.pad_t = d.pad_t; cv.pad_l = d.pad_l; cv.up = d.up; cv.ktp = (int)ktp;
            const int rc2 = ug_gemm_launch_conv256(g, cv, (hipStream_t)stream);
            if (rc2 != UG_ERR_UNSUPPORTED) return rc2;
        }
    }
    // 256 x 128 tiles, three stages (Cout = 128 layers; convolutions with too few 256^2 tiles): whole tiles in M, at least one tile per CU pair
    [[maybe_unused]] constexpr int LDS_BIG = 3 * (256 + CBN) * CBK * 2, LDS_SMALL = 2 * (128 + CBN) * CBK * 2;
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void*)conv2d_nhwc_kernel<2, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_SMALL);
#ifdef UG_PROBE_BUILD
        (void)hipFuncSetAttribute((const void*)conv2d_nhwc_kernel<4, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BIG);
#endif
        attr = true;
    }
    const int64_t nN = (d.Cout + CBN - 1) / CBN;
#ifdef UG_PROBE_BUILD      /* the 256 x 128 three-stage form: bit-identical and no faster (round 3) - probe library only */
    if (M % 256 == 0 && (M / 256) * nN >= ug_env_int("UG_CONV_BIG_MIN_TILES", 128) && ug_env_int("UG_CONV_BIG", 0)) {
        const int64_t grid = (M / 256) * nN;
        UG_REQUIRE(grid < (1ll << 31), UG_ERR_UNSUPPORTED, "ug_conv2d_nhwc: grid too large");
        hipLaunchKernelGGL((conv2d_nhwc_kernel<4, 3>), dim3((unsigned)grid), dim3(512), LDS_BIG, (hipStream_t)stream, p);
        UG_CHECK_LAUNCH("ug_conv2d_nhwc");
        return UG_OK;
    }
#endif
    const int64_t grid = ((M + 127) / 128) * nN;
    UG_REQUIRE(grid < (1ll << 31), UG_ERR_UNSUPPORTED, "ug_conv2d_nhwc: grid too large");
    hipLaunchKernelGGL((conv2d_nhwc_kernel<2, 2>), dim3((unsigned)grid), dim3(256), LDS_SMALL, (hipStream_t)stream, p);
    UG_CHECK_LAUNCH("ug_conv2d_nhwc");
    return UG_OK;
}

extern "C" int ug_conv2d_nhwc_f32(const ug_conv_desc* dp, ug_stream_t stream) {
    UG_REQUIRE(dp != nullptr, UG_ERR_BAD_SHAPE, "ug_conv2d_nhwc_f32: null descriptor");
    const ug_conv_desc& d = *dp;
    const int rc = conv_check(d, false);
    if (rc != UG_OK) return rc;
    ConvPF p;
    p.x = (const float*)d.x; p.w = (const float*)d.w; p.bias = (const float*)d.bias; p.R = (const float*)d.R; p.out = (float*)d.out;
    p.B = (int)d.B; p.H = (int)d.H; p.W = (int)d.W; p.Cin = (int)d.Cin; p.Ho = (int)d.Ho; p.Wo = (int)d.Wo; p.Cout = (int)d.Cout;
    p.KH = d.KH; p.KW = d.KW; p.stride = d.stride; p.pad_t = d.pad_t; p.pad_l = d.pad_l; p.up = d.up;
    const int64_t total = d.B * d.Ho * d.Wo * d.Cout;
    hipLaunchKernelGGL(conv2d_nhwc_f32_kernel, dim3((unsigned)std::min<int64_t>((total + 255) / 256, 16384)), dim3(256), 0, (hipStream_t)stream, p);
    UG_CHECK_LAUNCH("ug_conv2d_nhwc_f32");
    return UG_OK;
}

extern "C" int ug_groupnorm_nhwc(const void* x, const void* gamma, const void* beta, void* out, void* ws, int64_t wsb, int64_t B, int64_t HW, int64_t C, int32_t G, float eps,
                                 int32_t silu, ug_stream_t s) { return groupnorm_impl<bf16_t>(x, gamma, beta, out, ws, wsb, B, HW, C, G, eps, silu, s); }
extern "C" int ug_groupnorm_nhwc_f32(const void* x, const void* gamma, const void* beta, void* out, void* ws, int64_t wsb, int64_t B, int64_t HW, int64_t C, int32_t G, float eps,
                                     int32_t silu, ug_stream_t s) { return groupnorm_impl<float>(x, gamma, beta, out, ws, wsb, B, HW, C, G, eps, silu, s); }
extern "C" int ug_softmax_rows(const float* S, int64_t ld_s, void* P, int64_t ld_p, int64_t rows, int64_t cols, float scale, ug_stream_t s) {
    return softmax_impl<bf16_t>(S, ld_s, P, ld_p, rows, cols, scale, s);
}
extern "C" int ug_softmax_rows_f32(const float* S, int64_t ld_s, void* P, int64_t ld_p, int64_t rows, int64_t cols, float scale, ug_stream_t s) {
    return softmax_impl<float>(S, ld_s, P, ld_p, rows, cols, scale, s);
}
extern "C" int ug_nchw_to_nhwc(const void* in, void* out, int64_t B, int64_t C, int64_t HW, int64_t Cp, float div, float add, ug_stream_t s) {
    return to_nhwc_impl<bf16_t>(in, out, B, C, HW, Cp, div, add, s);
}
extern "C" int ug_nchw_to_nhwc_f32(const void* in, void* out, int64_t B, int64_t C, int64_t HW, int64_t Cp, float div, float add, ug_stream_t s) {
    return to_nhwc_impl<float>(in, out, B, C, HW, Cp, div, add, s);
}
extern "C" int ug_nhwc_to_nchw(const void* in, void* out, int64_t B, int64_t C, int64_t HW, int64_t Cp, ug_stream_t s) { return to_nchw_impl<bf16_t>(in, out, B, C, HW, Cp, s); }
extern "C" int ug_nhwc_to_nchw_f32(const void* in, void* out, int64_t B, int64_t C, int64_t HW, int64_t Cp, ug_stream_t s) { return to_nchw_impl<float>(in, out, B, C, HW, Cp, s); }
extern "C" int ug_vae_sample(const void* mom, int64_t Cp, const void* noise, void* z, int64_t B, int64_t L, int64_t HW, float shift, float scale, ug_stream_t s) {
    return sample_impl<bf16_t>(mom, Cp, noise, z, B, L, HW, shift, scale, s);
}
extern "C" int ug_vae_sample_f32(const void* mom, int64_t Cp, const void* noise, void* z, int64_t B, int64_t L, int64_t HW, float shift, float scale, ug_stream_t s) {
    return sample_impl<float>(mom, Cp, noise, z, B, L, HW, shift, scale, s);
}
