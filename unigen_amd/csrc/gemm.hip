// bf16 MFMA GEMM with fused epilogues for the MM-DiT projections (gfx950).
//
//   C[m][n] = epi( sum_k A[m][k] * W[n][k] + bias[n] )      A: [M][K] activations, W: [N][K] (nn.Linear layout)
//
// Replaces every nn.Linear on the UniGen hot path (reference: torch F.linear -> BLAS; call sites in diffusers
// FluxTransformerBlock / FluxSingleTransformerBlock invoked at src/UniGenTransformer.py:1129,1151,1097,1102,1104 and
// the expert linears :957-959) together with the elementwise ops the reference runs as separate kernels after it.
//
// Kernel: 128x128x64 tile, 256 threads = 4 waves (2x2), each wave a 64x64 sub-tile as 4x4 v_mfma_f32_16x16x32_bf16
// accumulators. Operands are staged HBM -> LDS with global_load_lds_dwordx4 (no VGPR round trip) into two LDS
// buffers; the LDS image is lane-linear, so the bank-conflict XOR swizzle is applied to the per-lane SOURCE address
// and to the ds_read address (cdna guide rule 21). The MFMA is issued with W as the "A" operand so each lane ends
// up with 4 consecutive n for one m -> 8-byte row-major stores. Tails in M/N are handled by clamping load rows and
// predicating stores; K must be a multiple of 64.
#include "ug_common.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = BM * BK * 2;        // 16 KiB per operand tile
constexpr int BUF_BYTES = 2 * TILE_BYTES;      // A + W
constexpr int LDS_BYTES = 2 * BUF_BYTES;       // double buffered: 64 KiB -> 2 workgroups / CU

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

__device__ __forceinline__ void glds16(const void* g, unsigned char* l) {
    __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)l, 16, 0, 0);
}

struct TileCoord { int tm, tn; };

// XCD-aware tile order: blocks b and b+8 share an XCD (round-robin dispatch), so give each XCD a contiguous chunk of
// the tile sequence, and walk that sequence in groups of 8 M-tiles x all N-tiles so co-resident tiles share A/W panels
// in the XCD's L2. Only affects speed.
__device__ __forceinline__ TileCoord tile_of_block(int bid, int nM, int nN) {
    const int nwg = nM * nN;
    const int q = nwg >> 3, r = nwg & 7;
    const int xcd = bid & 7, k = bid >> 3;
    const int id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;  // bijective for any nwg
    constexpr int GROUP_M = 8;
    const int per_group = GROUP_M * nN;
    const int gid = id / per_group;
    const int first_m = gid * GROUP_M;
    const int gsz = min(nM - first_m, GROUP_M);
    const int rem = id - gid * per_group;
    TileCoord t;
    t.tm = first_m + rem % gsz;
    t.tn = rem / gsz;
    return t;
}

__device__ __forceinline__ float gelu_tanh(float x) {
    // 0.5 x (1 + tanh( sqrt(2/pi) (x + 0.044715 x^3) ));  tanh(u) = 1 - 2 / (1 + exp(2u))
    const float u = 0.7978845608028654f * (x + 0.044715f * x * x * x);
    const float e = __expf(2.0f * u);
    const float t = 1.0f - 2.0f / (1.0f + e);
    return 0.5f * x * (1.0f + t);
}

template <int EPI>
__global__ __launch_bounds__(256, 2) void gemm128_kernel(const ug_gemm_desc p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;

    const int64_t M = p.M, N = p.N;
    const int nM = (int)((M + BM - 1) / BM), nN = (int)((N + BN - 1) / BN);
    const TileCoord tc = tile_of_block(blockIdx.x, nM, nN);
    const int64_t m0 = (int64_t)tc.tm * BM, n0 = (int64_t)tc.tn * BN;
    const int g = blockIdx.z;

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // fragment read offsets inside a tile (bytes), before the per-k-substep chunk XOR
    const int frow = lane & 15, fch = lane >> 4, fsw = lane & 7;

    // One pass = one (A, W, K) segment: the base product, then optionally the LoRA product T . B^T.
    for (int seg = 0; seg < 2; ++seg) {
        const bf16_t* Ab; const bf16_t* Wb; int64_t lda, ldw, Kseg, a_rpb, a_bs;
        if (seg == 0) {
            Ab = (const bf16_t*)p.A + (int64_t)g * p.a_gstride; lda = p.lda; a_rpb = p.a_rpb; a_bs = p.a_bstride;
            Wb = (const bf16_t*)p.W + (int64_t)g * p.w_gstride; ldw = p.ldw; Kseg = p.K;
        } else {
            if (p.lora_r <= 0) break;
            Ab = (const bf16_t*)p.lora_T; lda = p.ldt; a_rpb = 0; a_bs = 0;
            Wb = (const bf16_t*)p.lora_B; ldw = p.ldb; Kseg = p.lora_r;
        }
        // per-lane staging sources: 4 glds for A rows, 4 for W rows per K-tile
        const bf16_t* asrc[4]; const bf16_t* wsrc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = wave * 32 + i * 8 + (lane >> 3);
            const int c = (lane & 7) ^ (row & 7);              // source chunk that lands at linear position lane&7
            int64_t am = m0 + row; if (am > M - 1) am = M - 1;
            int64_t wn = n0 + row; if (wn > N - 1) wn = N - 1;
            asrc[i] = Ab + ug_rowmap(am, a_rpb, a_bs) * lda + c * 8;
            wsrc[i] = Wb + wn * ldw + c * 8;
        }
        const int nk = (int)(Kseg / BK);

        auto stage = [&](int buf, int kt) {
            unsigned char* Abuf = smem + buf * BUF_BYTES;
            unsigned char* Wbuf = Abuf + TILE_BYTES;
            const int64_t ko = (int64_t)kt * BK;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int off = (wave * 32 + i * 8) * 128;       // wave-uniform LDS row base; hardware adds lane*16
                glds16(asrc[i] + ko, Abuf + off);
                glds16(wsrc[i] + ko, Wbuf + off);
            }
        };

        stage(0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        for (int kt = 0; kt < nk; ++kt) {
            const int cur = kt & 1;
            if (kt + 1 < nk) stage(cur ^ 1, kt + 1);
            const unsigned char* Abuf = smem + cur * BUF_BYTES;
            const unsigned char* Wbuf = Abuf + TILE_BYTES;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                bf16x8 af[4], wf[4];
                const int choff = (((s * 4 + fch) ^ fsw) << 4);
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    af[i] = *(const bf16x8*)(Abuf + (wr * 64 + i * 16 + frow) * 128 + choff);
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    wf[j] = *(const bf16x8*)(Wbuf + (wc * 64 + j * 16 + frow) * 128 + choff);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], af[i], acc[i][j], 0, 0, 0);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
    }

    // ---- epilogue: lane holds D[row = n (4 consecutive)][col = m] ------------------------------------------
    const bf16_t* bias = p.bias ? (const bf16_t*)p.bias + (int64_t)g * p.bias_gstride : nullptr;
    const int64_t cg = (int64_t)g * p.c_gstride;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int64_t n = n0 + wc * 64 + j * 16 + (lane >> 4) * 4;
        if (n >= N) continue;
        float bv[4] = {0.f, 0.f, 0.f, 0.f};
        if (bias) {
            const u32x2 b2 = *(const u32x2*)(bias + n);
            bv[0] = bflo(b2.x); bv[1] = bfhi(b2.x); bv[2] = bflo(b2.y); bv[3] = bfhi(b2.y);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int64_t m = m0 + wr * 64 + i * 16 + (lane & 15);
            if (m >= M) continue;
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = acc[i][j][r] + bv[r];
            const int64_t crow = ug_rowmap(m, p.c_rpb, p.c_bstride);
            if constexpr (EPI == UG_EPI_F32) {
                float* C = (float*)p.C + cg + crow * p.ldc + n;
                *(f32x4*)C = (f32x4){v[0], v[1], v[2], v[3]};
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = rbf(v[r]);
                if constexpr (EPI == UG_EPI_BIAS_GELU) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = gelu_tanh(v[r]);
                } else if constexpr (EPI == UG_EPI_RES_GATE || EPI == UG_EPI_RES_SCALE) {
                    const bf16_t* R = (const bf16_t*)p.R + ug_rowmap(m, p.r_rpb, p.r_bstride) * p.ldr + n;
                    const u32x2 r2 = *(const u32x2*)R;
                    const float rv[4] = {bflo(r2.x), bfhi(r2.x), bflo(r2.y), bfhi(r2.y)};
                    if constexpr (EPI == UG_EPI_RES_GATE) {
                        const bf16_t* G = (const bf16_t*)p.gate + (m / p.rows_per_sample) * p.gate_ld + n;
                        const u32x2 g2 = *(const u32x2*)G;
                        const float gv[4] = {bflo(g2.x), bfhi(g2.x), bflo(g2.y), bfhi(g2.y)};
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = rv[r] + rbf(gv[r] * v[r]);
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = rv[r] + rbf(p.alpha * v[r]);
                    }
                }
                bf16_t* C = (bf16_t*)p.C + cg + crow * p.ldc + n;
                u32x2 o; o.x = pack2bf(v[0], v[1]); o.y = pack2bf(v[2], v[3]);
                *(u32x2*)C = o;
            }
        }
    }
}

template <int EPI>
int launch(const ug_gemm_desc& d, hipStream_t s) {
    const int nM = (int)((d.M + BM - 1) / BM), nN = (int)((d.N + BN - 1) / BN);
    dim3 grid((unsigned)(nM * nN), 1, (unsigned)(d.groups > 0 ? d.groups : 1));
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)gemm128_kernel<EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        attr_set = true;
    }
    hipLaunchKernelGGL(gemm128_kernel<EPI>, grid, dim3(256), LDS_BYTES, s, d);
    UG_CHECK_LAUNCH("ug_gemm_bf16");
    return UG_OK;
}

}  // namespace

extern "C" int ug_gemm_bf16(const ug_gemm_desc* dp, ug_stream_t stream) {
    UG_REQUIRE(dp != nullptr, UG_ERR_BAD_SHAPE, "ug_gemm_bf16: null descriptor");
    ug_gemm_desc d = *dp;
    if (d.groups <= 0) d.groups = 1;
    UG_REQUIRE(d.M >= 0 && d.N > 0 && d.K > 0, UG_ERR_BAD_SHAPE, "ug_gemm_bf16: bad M/N/K %lld/%lld/%lld",
               (long long)d.M, (long long)d.N, (long long)d.K);
    if (d.M == 0) return UG_OK;
    UG_REQUIRE(d.K % BK == 0, UG_ERR_UNSUPPORTED, "ug_gemm_bf16: K=%lld must be a multiple of %d", (long long)d.K, BK);
    UG_REQUIRE(d.N % 4 == 0, UG_ERR_UNSUPPORTED, "ug_gemm_bf16: N=%lld must be a multiple of 4", (long long)d.N);
    UG_REQUIRE(d.A && d.W && d.C, UG_ERR_BAD_SHAPE, "ug_gemm_bf16: null operand");
    UG_REQUIRE(d.lda >= d.K && d.ldw >= d.K && d.ldc >= d.N, UG_ERR_BAD_SHAPE, "ug_gemm_bf16: leading dims too small");
    UG_REQUIRE(d.lda % 8 == 0 && d.ldw % 8 == 0 && ug_aligned(d.A, 16) && ug_aligned(d.W, 16) &&
               d.a_gstride % 8 == 0 && d.w_gstride % 8 == 0,
               UG_ERR_BAD_ALIGN, "ug_gemm_bf16: A/W need 16-byte aligned rows (lda, ldw multiples of 8)");
    const bool f32 = d.epilogue == UG_EPI_F32;
    UG_REQUIRE(d.ldc % 4 == 0 && ug_aligned(d.C, f32 ? 16 : 8) && d.c_gstride % 4 == 0, UG_ERR_BAD_ALIGN,
               "ug_gemm_bf16: C needs ldc %% 4 == 0 and an aligned base");
    UG_REQUIRE(!d.bias || (ug_aligned(d.bias, 8) && d.bias_gstride % 4 == 0), UG_ERR_BAD_ALIGN, "ug_gemm_bf16: bias alignment");
    if (d.epilogue == UG_EPI_RES_GATE || d.epilogue == UG_EPI_RES_SCALE) {
        UG_REQUIRE(d.R && d.ldr % 4 == 0 && ug_aligned(d.R, 8), UG_ERR_BAD_ALIGN, "ug_gemm_bf16: residual missing/misaligned");
        UG_REQUIRE(d.groups == 1, UG_ERR_UNSUPPORTED, "ug_gemm_bf16: residual epilogues are not grouped");
    }
    if (d.epilogue == UG_EPI_RES_GATE)
        UG_REQUIRE(d.gate && d.rows_per_sample > 0 && d.gate_ld % 4 == 0 && ug_aligned(d.gate, 8), UG_ERR_BAD_SHAPE,
                   "ug_gemm_bf16: gate missing/misaligned");
    if (d.lora_r > 0) {
        UG_REQUIRE(d.lora_r % BK == 0, UG_ERR_UNSUPPORTED, "ug_gemm_bf16: lora_r=%d must be padded to a multiple of %d", d.lora_r, BK);
        UG_REQUIRE(d.lora_T && d.lora_B && d.ldt % 8 == 0 && d.ldb % 8 == 0 && ug_aligned(d.lora_T, 16) && ug_aligned(d.lora_B, 16),
                   UG_ERR_BAD_ALIGN, "ug_gemm_bf16: LoRA operands missing/misaligned");
        UG_REQUIRE(d.groups == 1, UG_ERR_UNSUPPORTED, "ug_gemm_bf16: LoRA epilogue is not grouped");
    }
    hipStream_t s = (hipStream_t)stream;
    switch (d.epilogue) {
        case UG_EPI_BIAS: return launch<UG_EPI_BIAS>(d, s);
        case UG_EPI_BIAS_GELU: return launch<UG_EPI_BIAS_GELU>(d, s);
        case UG_EPI_RES_GATE: return launch<UG_EPI_RES_GATE>(d, s);
        case UG_EPI_RES_SCALE: return launch<UG_EPI_RES_SCALE>(d, s);
        case UG_EPI_F32: return launch<UG_EPI_F32>(d, s);
        default: UG_FAIL(UG_ERR_UNSUPPORTED, "ug_gemm_bf16: unknown epilogue %d", d.epilogue);
    }
}
