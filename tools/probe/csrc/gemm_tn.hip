// C[I][J] = sum_r A[r][I] * B[r][J]  (bf16 in, fp32 accumulate, bf16 out): the weight gradient dW = dY^T X of a Linear layer (reference: autograd of
// F.linear under accelerator.backward, train.py:652) WITHOUT materialised transposes. Both operands arrive "contraction-major" (the contraction index
// r is the row of both row-major matrices), so both MFMA fragments are read from LDS with the transposing ds_read_b64_tr_b16 - the access the
// attention kernel uses for V^T - out of 64-row x 128-column tiles in the attention kernel's swizzled 256-byte-row image. 128 x 128 output tile,
// 4 waves (2 x 2, each 64 x 64 = four 32x32x16 accumulators), two tile pairs in LDS (64 KiB, 2 workgroups / CU), register staging one tile ahead.
#include "ug_common.h"
#include "../unigen_hip_probe.h"   // probe library only: not part of the product C ABI

namespace {

typedef __attribute__((address_space(3))) bf16x4* lds_b64_ptr;

__device__ __forceinline__ int tn_row_swz(int row) { return ((row & 3) << 2) | ((row >> 2) & 3); }       // as attention.hip, 256-byte rows
__device__ __forceinline__ int tn_img_off(int row, int ch) { return 256 * row + 16 * (ch ^ tn_row_swz(row)); }
__device__ __forceinline__ bf16x8 tn_tr_pair(const unsigned char* lo, const unsigned char* hi) {
    const bf16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_b64_ptr)lo);
    const bf16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_b64_ptr)hi);
    return (bf16x8){a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
}

constexpr int TN_TILE = 64 * 256;      // one operand tile: 64 rows x 128 bf16

__global__ __launch_bounds__(256, 2) void gemm_tn_kernel(const bf16_t* __restrict__ A, int64_t lda, const bf16_t* __restrict__ B, int64_t ldb,
                                                         bf16_t* __restrict__ C, int64_t ldc, int R, int I, int J, int nJ) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];      // [2][A tile | B tile]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wi = wave >> 1, wj = wave & 1;
    const int ti = blockIdx.x / nJ, tj = blockIdx.x - ti * nJ;
    const int i0 = ti * 128, j0 = tj * 128;
    // staging: thread -> 4 chunks (16 bytes) of each tile
    int st_row[4], st_ch[4], st_off[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int cid = tid + 256 * u;
        st_row[u] = cid >> 4; st_ch[u] = cid & 15;
        st_off[u] = tn_img_off(st_row[u], st_ch[u]);
    }
    u32x4 ra[4], rb[4];
    auto stage_load = [&](int r0) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int r = r0 + st_row[u];
            const int ci = i0 + st_ch[u] * 8, cj = j0 + st_ch[u] * 8;
            ra[u] = (u32x4){0u, 0u, 0u, 0u}; rb[u] = (u32x4){0u, 0u, 0u, 0u};
            if (r < R && ci < I) ra[u] = *(const u32x4*)(A + (int64_t)r * lda + ci);       // I, J multiples of 8: a chunk is in or out as a whole
            if (r < R && cj < J) rb[u] = *(const u32x4*)(B + (int64_t)r * ldb + cj);
        }
    };
    auto stage_write = [&](int buf) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            *(u32x4*)(smem + buf * 2 * TN_TILE + st_off[u]) = ra[u];
            *(u32x4*)(smem + buf * 2 * TN_TILE + TN_TILE + st_off[u]) = rb[u];
        }
    };
    // transposed-read offsets (attention.hip's V^T pattern): lane -> column 32 blk + (lane & 31), rows 16 ks + {4h + (i16 >> 2), + 8}
    const int h = lane >> 5, i16 = lane & 15, g16 = lane >> 4;
    const int t_row = 4 * h + (i16 >> 2), t_lowch = 2 * (g16 & 1) + ((i16 & 3) >> 1), t_b8 = 8 * (i16 & 1);
    int off_lo[4], off_hi[4];
#pragma unroll
    for (int blk = 0; blk < 4; ++blk) {
        const int ch = 4 * blk + t_lowch;
        off_lo[blk] = 256 * t_row + 16 * (ch ^ tn_row_swz(t_row)) + t_b8;
        off_hi[blk] = 256 * (t_row + 8) + 16 * (ch ^ tn_row_swz(t_row + 8)) + t_b8;
    }
    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;
    const int ntiles = (R + 63) / 64;
    stage_load(0);
    stage_write(0);
    __syncthreads();
    for (int t = 0; t < ntiles; ++t) {
        const int cur = t & 1;
        if (t + 1 < ntiles) stage_load((t + 1) * 64);
        const unsigned char* Ab = smem + cur * 2 * TN_TILE;
        const unsigned char* Bb = Ab + TN_TILE;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            bf16x8 af[2], bfr[2];
#pragma unroll
            for (int x = 0; x < 2; ++x) {
                af[x] = tn_tr_pair(Ab + ks * 16 * 256 + off_lo[wi * 2 + x], Ab + ks * 16 * 256 + off_hi[wi * 2 + x]);
                bfr[x] = tn_tr_pair(Bb + ks * 16 * 256 + off_lo[wj * 2 + x], Bb + ks * 16 * 256 + off_hi[wj * 2 + x]);
            }
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a], bfr[b], acc[a][b], 0, 0, 0);
        }
        if (t + 1 < ntiles) stage_write(cur ^ 1);
        __syncthreads();
    }
    // accumulator element i of lane (n = lane & 31, h): row m = (i & 3) + 8 (i >> 2) + 4 h of the 32 x 32 block
    const int n = lane & 31;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int col = j0 + wj * 64 + b * 32 + n;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int row = i0 + wi * 64 + a * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
                if (row < I && col < J) C[(int64_t)row * ldc + col] = f2bf(acc[a][b][i]);
            }
        }
}

}  // namespace

extern "C" int ug_gemm_tn_bf16(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, int64_t R, int64_t I, int64_t J,
                               ug_stream_t stream) {
    if (I == 0 || J == 0) return UG_OK;
    UG_REQUIRE(A && B && C && R > 0 && I > 0 && J > 0 && lda >= I && ldb >= J && ldc >= J, UG_ERR_BAD_SHAPE, "ug_gemm_tn_bf16: bad arguments");
    UG_REQUIRE(I % 8 == 0 && J % 8 == 0 && lda % 8 == 0 && ldb % 8 == 0 && ug_aligned(A, 16) && ug_aligned(B, 16), UG_ERR_BAD_ALIGN,
               "ug_gemm_tn_bf16: I, J and the leading dimensions must be multiples of 8, bases 16-byte aligned");
    UG_REQUIRE(R < (1ll << 31) && I < (1ll << 31) && J < (1ll << 31), UG_ERR_UNSUPPORTED, "ug_gemm_tn_bf16: sizes must fit 31 bits");
    const int64_t nI = (I + 127) / 128, nJ = (J + 127) / 128;
    UG_REQUIRE(nI * nJ < (1ll << 31), UG_ERR_UNSUPPORTED, "ug_gemm_tn_bf16: grid too large");
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute((const void*)gemm_tn_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * TN_TILE); attr = true; }
    hipLaunchKernelGGL(gemm_tn_kernel, dim3((unsigned)(nI * nJ)), dim3(256), 4 * TN_TILE, (hipStream_t)stream, (const bf16_t*)A, lda, (const bf16_t*)B, ldb,
                       (bf16_t*)C, ldc, (int)R, (int)I, (int)J, (int)nJ);
    UG_CHECK_LAUNCH("ug_gemm_tn_bf16");
    return UG_OK;
}
