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
#include "gemm_epilogue.h"
#include <stdlib.h>
#include <type_traits>

#ifndef UG_GEMM_SLAB_SC
#define UG_GEMM_SLAB_SC 1
#endif
#ifndef UG_GEMM128_W8
#define UG_GEMM128_W8 2      // round 6: 8-wave workgroups everywhere the 128^2 kernel runs: +3...+7 % on its launches, bit-identical (profiles/r06aa_gemm128_w8_ab.log)
#endif
#ifndef UG_GEMM_SPLITK_MIN_KT_DEFAULT
#define UG_GEMM_SPLITK_MIN_KT_DEFAULT 96
#endif
namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = BM * BK * 2;        // 16 KiB per operand tile
constexpr int BUF_BYTES = 2 * TILE_BYTES;      // A + W
constexpr int LDS_BYTES = 2 * BUF_BYTES;       // double buffered: 64 KiB -> 2 workgroups / CU

// NWV (round 6): waves per workgroup. 4 = rounds 1-5: a wave owns 64 x 64 of the tile and issues 8 LDS-DMAs per 32 MFMAs - alone on a CU (launches of at most
// one tile per CU: the batch-1 text / 512^2 projections) that issue sequence, not its latency, paces the K loop (0.94 us per K-tile = 28 % of the CU's MFMA
// rate; a four-stage prefetch changed nothing, profiles/r06q_gemm128_four_stage_ab.log). 8 = a wave owns 32 x 64 (16 MFMAs, 4 DMAs, 12 fragment reads per
// K-tile): two waves per SIMD from ONE workgroup cover each other's issue stalls. Same MFMA shape, same K order per output element: bit-identical.
template <int EPI, int NWV = 4>
__global__ __launch_bounds__(64 * NWV, 2) void gemm128_kernel(const ug_gemm_desc p) {
    constexpr int MI = 16 / NWV;                       // 16-row m-subtiles per wave: 4 (64 rows) or 2 (32 rows)
    constexpr int SR = 16 / NWV;                       // 8-row staging groups per wave and operand: 4 (32 rows) or 2 (16 rows)
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

    f32x4 acc[MI][4];
#pragma unroll
    for (int i = 0; i < MI; ++i)
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
        const bf16_t* asrc[SR]; const bf16_t* wsrc[SR];
#pragma unroll
        for (int i = 0; i < SR; ++i) {
            const int row = wave * (8 * SR) + i * 8 + (lane >> 3);
            const int c = (lane & 7) ^ (row & 7);              // source chunk that lands at linear position lane&7
            int64_t am = m0 + row; if (am > M - 1) am = M - 1;
            int64_t wn = n0 + row; if (wn > N - 1) wn = N - 1;
            asrc[i] = Ab + (int64_t)rowmap32((unsigned)am, (unsigned)a_rpb, (unsigned)a_bs) * lda + c * 8;
            wsrc[i] = Wb + wn * ldw + c * 8;
        }
        const int nk = (int)(Kseg / BK);

        auto stage = [&](int buf, int kt) {
            unsigned char* Abuf = smem + buf * BUF_BYTES;
            unsigned char* Wbuf = Abuf + TILE_BYTES;
            const int64_t ko = (int64_t)kt * BK;
#pragma unroll
            for (int i = 0; i < SR; ++i) {
                const int off = (wave * (8 * SR) + i * 8) * 128;       // wave-uniform LDS row base; hardware adds lane*16
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
                bf16x8 af[MI], wf[4];
                const int choff = (((s * 4 + fch) ^ fsw) << 4);
#pragma unroll
                for (int i = 0; i < MI; ++i)
                    af[i] = *(const bf16x8*)(Abuf + (wr * (16 * MI) + i * 16 + frow) * 128 + choff);
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    wf[j] = *(const bf16x8*)(Wbuf + (wc * 64 + j * 16 + frow) * 128 + choff);
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], af[i], acc[i][j], 0, 0, 0);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
    }

    // ---- epilogue: lane holds D[row = n (4 consecutive)][col = m]; rows outer so the row context is computed once ----
    const bf16_t* bias = p.bias ? (const bf16_t*)p.bias + (int64_t)g * p.bias_gstride : nullptr;
    float bv[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int64_t n = n0 + wc * 64 + j * 16 + (lane >> 4) * 4;
        load_bias4(n < N ? bias : nullptr, n, bv[j]);
    }
    const TileSplit ts = tile_split<EPI>(p, n0);
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int64_t m = m0 + wr * (16 * MI) + i * 16 + (lane & 15);
        if (m >= M) continue;
        RowCtx rc = row_ctx<EPI>(p, g, (unsigned)m);
        rc.coff += ts.cshift;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int64_t n = n0 + wc * 64 + j * 16 + (lane >> 4) * 4;
            if (n >= N) continue;
            if (EPI == UG_EPI_BIAS_GELU && !ts.gelu) epi_store<UG_EPI_BIAS>(p, rc, n, acc[i][j], bv[j]);
            else epi_store<EPI>(p, rc, n, acc[i][j], bv[j]);
        }
    }
}

// =====================================================================================================================
// 256x256x64 tile, 512 threads = 8 waves, one workgroup per CU (128 KiB LDS), for the large projections.
//
// Each K-tile is four 128x64 half-tiles (slots A0, B0, B1, A1; 16 KiB each, same swizzled image as above), two K-tile
// buffers. Wave (wr, wc) owns rows {i*128 + wr*64 .. +64 : i = 0,1} and columns {j*128 + wc*32 .. +32 : j = 0,1}, i.e. four
// 64x32 quadrants (i, j); a phase computes one quadrant (16 MFMAs) and needs only ONE new operand sub-tile:
//   phase 0: (0,0) reads A0,B0 | phase 1: (0,1) reads B1 | phase 2: (1,1) reads A1 | phase 3: (1,0) re-reads B0.
// Every phase is  L: {ds_read sub-tile, issue the LDS-DMA of one half-tile of the NEXT K-tile, counted vmcnt}  s_barrier
// M: {16 MFMAs}  s_barrier.  Waves 4-7 (wr = 1) run one barrier behind waves 0-3: a SIMD hosts wave w and w+4, so one of
// its two waves is always in an M segment while the other is in an L segment (cdna guide, "two waves per SIMD" item 9).
// Loads stay in flight across barriers: vmcnt(4) keeps the two youngest half-tiles outstanding; raw s_barrier, never
// __syncthreads. Hazards: a half-tile is waited for (by every wave, for its own DMA) at the end of the L segment BEFORE the
// M segment that precedes its first read, so the wait of the late group still precedes the early group's read by a
// barrier; a slot is re-staged >= 3 segments after its last ds_read.
// =====================================================================================================================
constexpr int HT_BYTES = 128 * 64 * 2;
constexpr int KT_BYTES = 4 * HT_BYTES;
constexpr int LDS256_BYTES = 2 * KT_BYTES;
constexpr int SLOT_A0 = 0, SLOT_B0 = HT_BYTES, SLOT_B1 = 2 * HT_BYTES, SLOT_A1 = 3 * HT_BYTES;
#ifndef UG_EPI_RES_PREFETCH
#define UG_EPI_RES_PREFETCH 8          /* row-groups of residual chunks in flight in the full-tile epilogue: all 8 (2 = rounds 1-2). Two-library A/B: 4: +-0.5 %, 8: +0.5...+1.1 % on the R + gate * v shapes; no spills (252 registers) */
#endif
#ifdef UG_DIAG_PHASES          /* tools/gemm_phase_stamps.py: per-segment stamps of one steady K-tile (implies UG_DIAG_STAMPS) */
#define UG_DIAG_STAMPS
#endif
#ifdef UG_DIAG_PHASES
constexpr int UG_STAMP_LDS = 4096;
#elif defined(UG_DIAG_STAMPS)
constexpr int UG_STAMP_LDS = 2048;
#else
constexpr int UG_STAMP_LDS = 0;
#endif

template <int EPI, bool LORA, int QKDH = 128, bool CONV = false>
__global__ __launch_bounds__(512, 2) void gemm256_kernel(const ug_gemm_desc p, const int tiles_per_group, const int total_tiles, const int wide16_gm,
                                                          const int full_tiles, const int nslices, float* __restrict__ slabs,
                                                          unsigned* __restrict__ tickets, const UgConvGeom cv) {
    // PERSISTENT: the grid is one workgroup per CU; each walks tiles blockIdx.x, +gridDim.x, ... (same XCD-aware order as a plain
    // launch would see round by round). The first K-tile of the NEXT tile is put in flight before the epilogue of the current one,
    // so workgroup relaunch, address set-up and the first DMA latency overlap the C stores instead of preceding the main loop.
    //
    // TAIL: tiles [0, full_tiles) fill whole rounds of the grid. When the remaining R tiles would occupy <= half the CUs they
    // are each cut into `nslices` K-slices that run concurrently in the last round: a slice stores its fp32 accumulators as a
    // slab, publishes it (agent-scope release + ticket), and the LAST arriver of a tile re-reads every slab in slice order
    // (agent-scope acquire; fixed order -> bitwise reproducible) and runs the epilogue. Nobody waits on anybody: no spin.
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int64_t M = p.M, N = p.N;
    const int nM = (int)((M + 255) / 256), nN = (int)((N + 255) / 256);
    const int st_off = wave * 16 * 128;
    // Which tile column a wave's quadrant (., j) covers. Rounds 1-3: j * 128 + wc * 32 (B half-tile j holds tile columns j * 128 ..). Round 4 (NPERM,
    // every epilogue except the q/k RMSNorm + RoPE one, whose head arithmetic is written for the old map): wc * 64 + j * 32, i.e. row rho of B half-tile h
    // is tile column (rho >> 5) * 64 + h * 32 + (rho & 31) - a pure permutation of which W rows the DMAs fetch into which LDS rows. A wave then owns 64
    // CONTIGUOUS columns = one whole 128-byte line of every output row, and the full-tile epilogue stores 8 rows x 128 B per instruction instead of
    // 16 rows x 64 B: half-line writes run at 14.4 B/clk per CU, whole lines at 48.5 (tools/probe/store_rate.hip, profiles/r04j_store_rate_probe.log).
    // Same MFMAs in the same K order on every output element: bit-identical results.
    constexpr bool NPERM = EPI != UG_EPI_QKV_ROPE;
    constexpr int CJ = NPERM ? 32 : 128, CW = NPERM ? 64 : 32;
    auto bcol = [](int h, int rho) __attribute__((always_inline)) { return NPERM ? (rho >> 5) * 64 + h * 32 + (rho & 31) : h * 128 + rho; };
#ifdef UG_DIAG_STAMPS
    constexpr int stamp_off = LDS256_BYTES + 16 + (EPI == UG_EPI_QKV_ROPE ? 256 * 8 * 4 : 0);
#endif
    const int wide16 = wide16_gm & 1;            // bits 8.. of the argument: GROUP_M of the tile walk (0 = 8)
    const bool rows_contig = (wide16_gm & 2) != 0;   // the C (and R) row maps never split a 256-row tile (rows per batch % 256 == 0): one scalar map per tile
    int a_off, b_off, ch0, ch1;          // fragment read offsets; set per tile (see the tile loop)
    // LORA: the K loop runs on through a second segment, T[m][0..r) . B[n][0..r) (the same K-segment the 128^2 kernel appends), with
    // the staging pointers of a half-tile pair swapped to the LoRA operands (pre-biased by -K) just before their first LoRA K-tile.
    const int nkA = (int)(p.K / BK);
    const int nk = nkA + (LORA ? p.lora_r / BK : 0);

    // BUF (round 5): the LDS-DMAs of the plain product kernel are issued in BUFFER form - buffer_load_dwordx4 ... offen lds with one resource per operand
    // (base = the tile's first row of the group's matrix, wave-uniform), a per-lane byte offset that is constant through the whole K loop and the K
    // offset in an SGPR - instead of the global form's 64-bit per-lane address, which costs a 64-bit VALU add per DMA (2 of the loop's VALU
    // instructions each: tools/loop_census.py) and 16 VGPRs of pointers instead of 8 of offsets. Offsets are tile-relative, so they fit 32 bits for
    // any matrix size (the launcher checks the one bound that remains, a row map's batch jump inside a tile). The LoRA segment and the convolution
    // gather switch base addresses inside the K loop (lora_src / conv_src) and keep the global form.
    constexpr bool BUF = !LORA && !CONV;
    struct TileSrc { const bf16_t* a[2][2]; const bf16_t* b[2][2]; int64_t m0, n0; int g; int nk; int rem; int slice; unsigned pk[2][2];
                     const bf16_t* abase; const bf16_t* wbase; int vo_a[2][2], vo_b[2][2]; };
    // CONV (AutoencoderKL 3x3 convolutions, vae.hip): A row m is output pixel (b, oy, ox), kept packed per staging row (b << 24 | oy << 12 | ox);
    // the K axis runs tap by tap (ktp K-tiles each), and a half-tile pair's A pointers are re-derived for the next tap right before that tap's
    // first K-tile is staged (the LoRA segment's switch, once per tap): source pixel of the tap (stride, one-sided padding and nearest-2x
    // upsampling folded in exactly as conv2d_nhwc_kernel does), or the zero page for padding; pre-biased by -tap * Cin so the running K offset applies.
    auto conv_ptr = [&](unsigned pk, int tap, int c) __attribute__((always_inline)) {
        const int ky = tap / cv.KW, kx = tap - ky * cv.KW;
        const int yv = (int)((pk >> 12) & 0xfffu) * cv.stride + ky - cv.pad_t, xv = (int)(pk & 0xfffu) * cv.stride + kx - cv.pad_l;
        const bool ok = (unsigned)yv < (unsigned)(cv.H << cv.up) && (unsigned)xv < (unsigned)(cv.W << cv.up);
        const int64_t pix = ((int64_t)(pk >> 24) * cv.H + (yv >> cv.up)) * cv.W + (xv >> cv.up);
        return (ok ? (const bf16_t*)p.A + pix * cv.Cin : cv.zero) + c * 8 - (int64_t)tap * cv.Cin;
    };
    // work item w -> tile and K range. Items >= full_tiles are K-slices of remainder tile `rem`; a tile's slices share blockIdx & 7
    // (= one XCD under round-robin placement; speed only).
    const int n_items = full_tiles + (((total_tiles - full_tiles) + 7) / 8) * 8 * nslices;
    auto tile_src = [&](int w, int lane) {
        TileSrc t;
        int tile = w, kb = 0;
        t.nk = nk; t.rem = -1; t.slice = 0;
        if (w >= full_tiles && nslices > 1) {
            const int j = w - full_tiles;
            const int xcd = j & 7, kx = j >> 3;
            t.rem = (kx / nslices) * 8 + xcd;
            t.slice = kx % nslices;
            tile = full_tiles + t.rem;
            if (tile >= total_tiles) { t.nk = 0; tile = total_tiles - 1; }       // padding item of the 8-aligned remainder: no work
            else {
                kb = (t.slice * nk) / nslices;
                t.nk = ((t.slice + 1) * nk) / nslices - kb;
            }
        }
        t.g = tile / tiles_per_group;
        const TileCoord tc = tile_of_block(tile - t.g * tiles_per_group, nM, nN, ((wide16_gm >> 8) & 0xff) ? ((wide16_gm >> 8) & 0xff) : 8, (wide16_gm >> 16) & 3);
        t.m0 = (int64_t)tc.tm * 256; t.n0 = (int64_t)tc.tn * 256;
        const bf16_t* Ab = (const bf16_t*)p.A + (int64_t)t.g * p.a_gstride;
        const bf16_t* Wb = (const bf16_t*)p.W + (int64_t)t.g * p.w_gstride;
        const unsigned pr0 = BUF ? rowmap32((unsigned)t.m0, (unsigned)p.a_rpb, (unsigned)p.a_bstride) : 0u;      // physical row of the tile's first A row
        if constexpr (BUF) {
            t.abase = Ab + (int64_t)pr0 * p.lda + (int64_t)kb * BK;
            t.wbase = Wb + t.n0 * p.ldw + (int64_t)kb * BK;
        }
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int row = wave * 16 + i * 8 + (lane >> 3);
                const int c = (lane & 7) ^ (row & 7);
                int64_t am = t.m0 + h * 128 + row; if (am > M - 1) am = M - 1;
                int64_t wn = t.n0 + bcol(h, row); if (wn > N - 1) wn = N - 1;
                if constexpr (BUF) {       // byte offsets from the tile's bases; row maps are monotonic, so both are >= 0
                    t.vo_a[h][i] = (int)(((int64_t)(rowmap32((unsigned)am, (unsigned)p.a_rpb, (unsigned)p.a_bstride) - pr0) * p.lda + c * 8) * 2);
                    t.vo_b[h][i] = (int)(((wn - t.n0) * p.ldw + c * 8) * 2);
                    continue;
                }
                if constexpr (CONV) {
                    const unsigned hw = (unsigned)(cv.Ho * cv.Wo), bb = (unsigned)am / hw, rr = (unsigned)am - bb * hw, oy = rr / (unsigned)cv.Wo;
                    t.pk[h][i] = bb << 24 | oy << 12 | (rr - oy * (unsigned)cv.Wo);
                    t.a[h][i] = conv_ptr(t.pk[h][i], 0, c);
                } else {
                    t.a[h][i] = Ab + (int64_t)rowmap32((unsigned)am, (unsigned)p.a_rpb, (unsigned)p.a_bstride) * p.lda + c * 8 + (int64_t)kb * BK;
                }
                t.b[h][i] = Wb + wn * p.ldw + c * 8 + (int64_t)kb * BK;
            }
        return t;
    };
    auto conv_src = [&](TileSrc& t, int h, int tap) __attribute__((always_inline)) {
        int lane_l = lane;
        asm volatile("" : "+v"(lane_l));
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = wave * 16 + i * 8 + (lane_l >> 3);
            t.a[h][i] = conv_ptr(t.pk[h][i], tap, (lane_l & 7) ^ (row & 7));
        }
    };
    auto lora_src = [&](TileSrc& t, int h) {
        int lane_l = lane;
        asm volatile("" : "+v"(lane_l));       // keeps these addresses from being hoisted (and held in registers) across the K loop
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = wave * 16 + i * 8 + (lane_l >> 3);
            const int c = (lane_l & 7) ^ (row & 7);
            int64_t am = t.m0 + h * 128 + row; if (am > M - 1) am = M - 1;
            int64_t wn = t.n0 + bcol(h, row); if (wn > N - 1) wn = N - 1;
            t.a[h][i] = (const bf16_t*)p.lora_T + am * p.ldt + c * 8 - p.K;
            t.b[h][i] = (const bf16_t*)p.lora_B + wn * p.ldb + c * 8 - p.K;
        }
    };
    // (Round 3's cross-tile stream - the last two K-tiles of a tile staging the next tile's first two, UG_GEMM_XTILE - measured neutral
    // (profiles/r03o_*: -0.7...+0.9 % per shape) and lived on in the probe build until round 5, when the buffer-form DMAs replaced the per-lane pointers it
    // swapped; removed from the source. Rebuilt on the round-5 loop in buffer form the same round - the next tile's A0 / B0 pieces of K-tiles 0 and 1
    // issued by the last two K-tiles of this one, sources selected by value, no spills, 242-250 VGPRs, bit-identical: -0.6...+0.2 % per shape and
    // -0.3 % images/s interleaved on one box; with the epilogue's bias / gate loads hoisted ahead of those DMAs as well, 16-40 B/lane of scratch and
    // -2 %: profiles/r05_gemm_loop_ab.log steps 8 and 9. Issuing the first K-tiles' DMAs earlier does not buy back the 3-5 us a tile waits for them
    // (profiles/r05x_gemm_tile_stamps.log); what they queue behind there was not identified.)
    auto stage = [&](unsigned char* slot, const bf16_t* const (&src)[2], int64_t ko) {
        glds16(src[0] + ko, slot + st_off);
        glds16(src[1] + ko, slot + st_off + 8 * 128);
    };
    using lds_ptr = __attribute__((address_space(3))) void*;
    auto stage_buf = [&](unsigned char* slot, const bf16_t* base, const int (&vo)[2], int64_t ko) __attribute__((always_inline)) {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 0xffffffff, 0x00020000);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)(slot + st_off), 16, vo[0], (int)ko * 2, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)(slot + st_off + 8 * 128), 16, vo[1], (int)ko * 2, 0, 0);
    };
    // half-tile h of operand A / B of this work item, K offset ko (elements), into `slot`
    auto stA = [&](unsigned char* slot, const TileSrc& t, int h, int64_t ko) __attribute__((always_inline)) {
        if constexpr (BUF) stage_buf(slot, t.abase, t.vo_a[h], ko); else stage(slot, t.a[h], ko);
    };
    auto stB = [&](unsigned char* slot, const TileSrc& t, int h, int64_t ko) __attribute__((always_inline)) {
        if constexpr (BUF) stage_buf(slot, t.wbase, t.vo_b[h], ko); else stage(slot, t.b[h], ko);
    };
    auto stage_first = [&](const TileSrc& t) {
        stA(smem + SLOT_A0, t, 0, 0); stB(smem + SLOT_B0, t, 0, 0);
        stB(smem + SLOT_B1, t, 1, 0); stA(smem + SLOT_A1, t, 1, 0);
        if (t.nk > 1) { stA(smem + KT_BYTES + SLOT_A0, t, 0, BK); stB(smem + KT_BYTES + SLOT_B0, t, 0, BK); }
        if (t.nk >= 3) { stB(smem + KT_BYTES + SLOT_B1, t, 1, BK); stA(smem + KT_BYTES + SLOT_A1, t, 1, BK); }   // the whole ring
    };
    bf16x8 areg[4][2], breg[2][2], breg0[2][2];    // breg0: B0 fragments, kept from phase 0 to phase 3
    auto read_A = [&](const unsigned char* slot) {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            areg[mt][0] = *(const bf16x8*)(slot + a_off + mt * 2048 + ch0);
            areg[mt][1] = *(const bf16x8*)(slot + a_off + mt * 2048 + ch1);
        }
    };
    auto read_B = [&](bf16x8 (&br)[2][2], const unsigned char* slot) {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            br[nt][0] = *(const bf16x8*)(slot + b_off + nt * 2048 + ch0);
            br[nt][1] = *(const bf16x8*)(slot + b_off + nt * 2048 + ch1);
        }
    };
/* s_setprio 1 around the MFMA segments (round 1) measured 0.3-1.7 % SLOWER than none on every cfg2 shape (round 2, two libraries on one box:
 * 1167 vs 1149, 1364 vs 1353, 1395 vs 1380, 1507 vs 1479 TFLOP/s), as in the attention kernel: -DUG_DIAG_PRIO restores it for A/B. */
#ifdef UG_DIAG_PRIO
#define UG_GEMM_PRIO(X) __builtin_amdgcn_s_setprio(X)
#else
#define UG_GEMM_PRIO(X) do { } while (0)
#endif
#ifdef UG_DIAG_NOMMA      /* diagnostic build (tools only, WRONG results): the whole data movement of the kernel without its MFMAs */
#define UG_MMA_QUADRANT(I, J, BR)                                                                                      \
    do {                                                                                                               \
        _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) {                                                             \
            _Pragma("unroll") for (int mt = 0; mt < 4; ++mt) asm volatile("" ::"v"(areg[mt][ks]));                    \
            _Pragma("unroll") for (int nt = 0; nt < 2; ++nt) asm volatile("" ::"v"(BR[nt][ks]));                      \
        }                                                                                                              \
    } while (0)
#else
#define UG_MMA_QUADRANT(I, J, BR)                                                                                      \
    do {                                                                                                               \
        UG_GEMM_PRIO(1);                                                                                               \
        /* (Measured and dropped, round 4: nt outside mt, so that consecutive MFMAs share their FIRST source operand as in the vendor's loop - +1.5...+2.5 % */ \
        /* on the one-wave-per-SIMD probe kernel, -1...-1.5 % per shape and -1.2 % in the forward here: profiles/r04y_mfma_order.log) */              \
        _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                               \
            _Pragma("unroll") for (int mt = 0; mt < 4; ++mt)                                                           \
                _Pragma("unroll") for (int nt = 0; nt < 2; ++nt)                                                       \
                    acc[I][J][mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(BR[nt][ks], areg[mt][ks], acc[I][J][mt][nt], 0, 0, 0); \
        UG_GEMM_PRIO(0);                                                                                               \
    } while (0)
#endif
#define UG_BARRIER() do { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); } while (0)
/* -DUG_DIAG_STAMPS (tools/gemm_stamps.py, a separate library build): s_memtime at five points of every tile of a workgroup, kept in 2 KiB of LDS
 * behind the kernel's own (no global store inside the loop: stores count on the VM counter the ring's waits are counted against) and flushed to
 * the split-K slab area after the last tile. Points: 0 tile top, 1 first K-tile's operands landed, 2 K loop done, 3 next tile's ring requested
 * (epilogue starts), 4 epilogue issued. */
#ifdef UG_DIAG_STAMPS
    unsigned long long* const stamp_lds = (unsigned long long*)(smem + stamp_off);
    int tile_seq = 0;
    if (threadIdx.x == 0) stamp_lds[250] = __builtin_amdgcn_s_memrealtime();      // 100 MHz, one counter for the chip: when this workgroup entered
#define UG_STAMP(I) do { if (threadIdx.x == 0 && tile_seq < 50) stamp_lds[tile_seq * 5 + (I)] = __builtin_amdgcn_s_memtime(); } while (0)
#ifdef UG_DIAG_PHASES
    // one steady K-tile (kt == 10) of the workgroup's SECOND tile, lane 0 of waves 0 and 4 (one of each group): 4 stamps per phase - L segment start,
    // L done (reads / DMAs issued and the counted wait passed), first barrier passed, MFMAs issued - and one behind the last barrier
#define UG_PSTAMP(KT, I) do { if ((threadIdx.x & 255) == 0 && tile_seq == 1 && (KT) == 10) stamp_lds[256 + (threadIdx.x >> 8) * 32 + (I)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define UG_PSTAMP(KT, I) do { } while (0)
#endif
#else
#define UG_STAMP(I) do { } while (0)
#define UG_PSTAMP(KT, I) do { } while (0)
#endif

    int tile = blockIdx.x;
    if (tile >= n_items) return;
    bool stores_in_flight = false;      // the previous tile ended in the full-tile epilogue: exactly 16 C stores per wave behind the prefetch
    TileSrc cur = tile_src(tile, lane);
    if (cur.nk > 0) stage_first(cur);
    for (; tile < n_items; tile += gridDim.x) {
        const int nk = cur.nk;             // K-tiles of THIS work item (shadows the full count)
        if (nk == 0) break;                // padding item (only ever the last one of a workgroup)
        {   // Lane-derived constants of the K loop, re-derived per tile from an opaque copy of the lane id so that they are not live
            // through the epilogue (kept live across the whole tile loop, hipcc spilled a_off / b_off and reloaded them - with a
            // vmcnt(0) that drains the DMA ring - in every K-tile).
            int lane_m = lane;
            asm volatile("" : "+v"(lane_m));
            const int frow = lane_m & 15, fch = lane_m >> 4, fsw = lane_m & 7;
            a_off = (wr * 64 + frow) * 128; b_off = (wc * 32 + frow) * 128;
            ch0 = ((fch ^ fsw) << 4); ch1 = (((4 + fch) ^ fsw) << 4);
        }
        f32x4 acc[2][2][4][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b) acc[i][j][a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
        UG_STAMP(0);
        // In flight from before (issued ahead of the previous tile's epilogue, or below for the first tile):
        // A0(0) B0(0) B1(0) A1(0) [A0(1) B0(1)]. The 8 half-tile slots form a ring: a slot is re-staged two K-tiles ahead as soon as
        // its last ds_read is >= 3 segments old (B0's fragments stay in registers for phase 3, so its slot frees after phase 0):
        //   phase 0 stages B1(kt+1) | phase 1: A1(kt+1) | phase 2: A0(kt+2) | phase 3: B0(kt+2)
        // -> every half-tile has 5-6 phases (~1.3 K-tiles) of lead; vmcnt(8) keeps the four youngest half-tiles in flight.
        // (Measured on the epilogue, end of round 1: it is STORE-bound - skipping its arithmetic changes nothing, skipping its 16 stores per
        // wave saves ~5.5 us per tile-round at K = 3072 (7 %). Not the chip-wide burst: a start ramp that keeps the CUs of an XCD up to
        // 17 us apart for the whole launch (verified with per-CU stamps) gains nothing; not the line pattern either: 8 rows x 128 B per
        // store instruction instead of 16 x 64 B, or the two half-lines back to back, time the same. What is left is the drain time of the
        // stores on the in-order VM counter beyond the ~1.9 K-tiles of slack below.)
        // (Measured and dropped, end of round 2 (profiles/r02g_gemm_rim.log): a reads-in-M schedule - every sub-tile's fragments read by the
        // wave that uses them BETWEEN the MFMAs of the matrix segment one phase earlier (rotating into fragment registers whose last use
        // was just issued, waits and DMA issues shifted one phase, L segments left with the DMA issue only): bit-identical, all GEMM tests
        // green, -1...+1 % on every cfg2 shape with or without forced interleaving. Loop ablations of the same day, 8192^3: without the
        // ds_reads 1518 -> 1732 TFLOP/s, without the MFMAs 2747, without both 3149 (DMA issue + 8 barriers per K-tile alone: 48 % of the
        // loop's time): the fragment reads cost their ~14 % wherever they are issued - operand delivery, not segment placement.)
        // (Measured and dropped: signalling the hand-off barrier 2-4 MFMAs before the end of a segment, so that the other group is
        // released while this one still has MFMAs queued: -9 % - the two groups' MFMAs then share the pipe, matrix beside matrix.)
        // (Measured and dropped, round 5, timing-only build: HALF the barriers - waves 0-3 keep the one between a phase's load and matrix segments, waves 4-7
        // the one behind the matrix segment, so that a group runs matrix -> load without a hand-off and the groups' matrix segments may overlap; alone,
        // with the matrix segment at s_setprio 1, and entering at 1 / raising to 3 behind the first MFMA: 0.1-1.6 %, 1.1-4.1 %, 1.1-2.6 % SLOWER than the
        // eight strict hand-offs (profiles/r05_gemm_loop_ab.log step 10). The alternation itself is worth more than the ~90 cycles a hand-off costs.)
        // (Measured and dropped: 2 segments of 32 MFMAs per K-tile and wave group with all DMA issued by waves 4-7 - 4 barriers per
        // K-tile instead of 8 - ran 5-8 % slower on the full chip and equal on 16 CUs: the hand-offs are not where the loop loses time,
        // and the coarser slot lifetimes cut the DMA lead from ~1.3 to 1.0 K-tiles.)
        // Stores of the previous tile's epilogue (16 per wave after the full-tile epilogue) may still be in flight; on the in-order
        // VM counter they sit BETWEEN the DMAs prefetched ahead of that epilogue and the ones issued below. Waits that only cover
        // prefetched DMAs allow for them (+16), so the store drain (~3 us when every CU bursts its 128 KB at once) overlaps the
        // first K-tiles instead of preceding them. `pre` = the whole ring (K-tiles 0 and 1, 16 DMAs) was prefetched (nk >= 3):
        // nothing is staged in phases 0 / 1 of K-tile 0 and the first wait that covers a post-store DMA is phase 3 of K-tile 1
        // (~1.9 K-tiles of slack); otherwise (12 DMAs ahead) it is phase 0 of K-tile 1. The selection is a scalar branch per wait:
        // peeling the K loop instead made hipcc spill inside it.
#define UG_WAIT_VM(N, RELAXED)                                                                  \
    do {                                                                                        \
        if (RELAXED) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((N) + 16) : "memory");            \
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");                           \
    } while (0)
        const bool pre = nk >= 3, sif = stores_in_flight;
        if (pre) UG_WAIT_VM(12, sif); else if (nk > 1) UG_WAIT_VM(8, sif); else UG_WAIT_VM(4, sif);      // A0(0), B0(0) landed
        UG_BARRIER();
        UG_STAMP(1);
        if (wr == 1) UG_BARRIER();          // waves 4-7 run one barrier behind
        // One K-tile = four phases. STEADY (round 5): the K-tiles 2 .. nk - 3 of the plain kernel - nothing about them depends on the tile (both
        // look-ahead K-tiles exist, no wait has to allow for the previous epilogue's stores, the prefetched ring is used up) - run a copy of the body in
        // which every wait is a constant and every stage is unconditional; the generic copy, whose wait selection is a scalar branch per wait
        // (~45 branches and ~130 SALU instructions per 128 MFMAs: tools/loop_census.py, profiles/r05_loop_census_*.json), runs the first two and
        // the last two K-tiles of a tile only.
        // MODE 0 = generic, MODE 2 = steady (round 5, second step): in the steady state the B1 half-tile's two DMA pieces are issued one L segment
        // EARLIER - at the end of phase 3 of the K-tile before (the phase that reads nothing) instead of the start of phase 0 (the phase that reads
        // twelve fragments: a DMA piece issued among ds_reads costs 100-185 cycles of the segment, 25-60 in a gap without them:
        // MI355X_MICROARCH.md, LDS-DMA piece issue cost). Same issue ORDER, so every counted wait keeps its meaning; only the wait that now follows
        // five young half-tiles instead of four becomes vmcnt(10). The hand-overs are two flags of the generic body: the generic K-tile before the
        // first steady one stages B1 for it in its phase 3 (early_b1), the generic K-tile after the last steady one finds its B1 already staged (skip_b1).
        // (Only the plain kernel runs the steady copy. The convolution variant could - its tap-boundary pointer switches are hooks of the body in
        // either mode and touch the A pointers only - but with its gather registers the second copy of the loop spills 20 VGPRs (84 B of scratch per
        // lane: checked with -Rpass-analysis=kernel-resource-usage) where the single loop fits in 248; the LoRA variant's segment switch replaces the B1
        // pointers one K-tile AFTER the early B1 stage would need them.)
        constexpr bool STEADY_OK = BUF;
        const int steady_from = 2, steady_to = (STEADY_OK && nk >= 6) ? ((nk & 1) ? nk - 4 : nk - 3) : 1;      // steady K-tiles [from, to]: an even count
        auto ktile = [&](const int kt, auto mode_c, auto par_c) __attribute__((always_inline)) {
            constexpr int MODE = decltype(mode_c)::value;
            constexpr bool STEADY = MODE != 0;
            const bool early_b1 = !STEADY && kt + 1 == steady_from && steady_to >= steady_from;
            const bool skip_b1 = !STEADY && kt == steady_to + 1 && steady_to >= steady_from;
            constexpr int PAR = decltype(par_c)::value;             // the K-tile's ring parity when the caller knows it (steady pairs), else -1
            unsigned char* cb = smem + (PAR >= 0 ? PAR : (kt & 1)) * KT_BYTES;
            unsigned char* nb = smem + (PAR >= 0 ? PAR ^ 1 : ((kt & 1) ^ 1)) * KT_BYTES;
            const bool n1 = STEADY || kt + 1 < nk, n2 = STEADY || kt + 2 < nk;
            const bool pre0 = !STEADY && pre && kt == 0;
            const bool x01 = !STEADY && sif && (kt == 0 || (pre && kt == 1)), x3 = !STEADY && sif && kt == 0;
            const int64_t k1 = (int64_t)(kt + 1) * BK, k2 = (int64_t)(kt + 2) * BK;
            if (LORA && kt + 1 == nkA) lora_src(cur, 1);
            if (CONV && n1 && ((kt + 1) & (cv.ktp - 1)) == 0) conv_src(cur, 1, (kt + 1) / cv.ktp);
            // phase 0: quadrant (0,0)
            UG_PSTAMP(kt, 0);
            read_A(cb + SLOT_A0); read_B(breg0, cb + SLOT_B0);
            if (MODE >= 2) UG_WAIT_VM(8, false);                                     // B1(kt) landed; B1(kt+1) was staged in the phase 3 before
            else if (pre0) UG_WAIT_VM(10, x01);
            else if (n1) { if (!skip_b1) stB(nb + SLOT_B1, cur, 1, k1); UG_WAIT_VM(8, x01); }   // B1(kt) landed
            else UG_WAIT_VM(2, x01);
            UG_PSTAMP(kt, 1);
            UG_BARRIER();
            UG_PSTAMP(kt, 2);
            UG_MMA_QUADRANT(0, 0, breg0);
            UG_PSTAMP(kt, 3);
            UG_BARRIER();
            UG_PSTAMP(kt, 4);
            // phase 1: quadrant (0,1)
            // (Measured and dropped the same day: the steady copy issuing this phase's two DMA pieces BEFORE its four reads: +-0.5 % per shape)
            read_B(breg, cb + SLOT_B1);
            if (pre0) UG_WAIT_VM(8, x01);
            else if (n1) { stA(nb + SLOT_A1, cur, 1, k1); UG_WAIT_VM(8, x01); }   // A1(kt) landed
            else UG_WAIT_VM(0, x01);
            UG_PSTAMP(kt, 5);
            UG_BARRIER();
            UG_PSTAMP(kt, 6);
            UG_MMA_QUADRANT(0, 1, breg);
            UG_PSTAMP(kt, 7);
            UG_BARRIER();
            UG_PSTAMP(kt, 8);
            // phase 2: quadrant (1,1)
            read_A(cb + SLOT_A1);
            if (LORA && kt + 2 == nkA) lora_src(cur, 0);
            if (CONV && n2 && ((kt + 2) & (cv.ktp - 1)) == 0) conv_src(cur, 0, (kt + 2) / cv.ktp);
            if (n2) stA(cb + SLOT_A0, cur, 0, k2);
            UG_PSTAMP(kt, 9);
            UG_BARRIER();
            UG_PSTAMP(kt, 10);
            UG_MMA_QUADRANT(1, 1, breg);
            UG_PSTAMP(kt, 11);
            UG_BARRIER();
            UG_PSTAMP(kt, 12);
            // phase 3: quadrant (1,0), B0 from registers
            // (Measured and dropped the same day: A0(kt+2) in this phase too - six pieces, issue order unchanged - is 2-7 % SLOWER: profiles/r05_gemm_loop_ab.log)
            // (... and: A1 + B1(kt+1) in phase 1, A0 + B0(kt+2) here, phases 0 / 2 reads only - four pieces among four reads, four alone: 0.9-2.1 % SLOWER)
            if (MODE == 2) { stB(cb + SLOT_B0, cur, 0, k2); stB(cb + SLOT_B1, cur, 1, k2); UG_WAIT_VM(10, false); }   // + B1(kt+2): its slot's last read was phase 1
            else if (n2 && early_b1) { stB(cb + SLOT_B0, cur, 0, k2); stB(cb + SLOT_B1, cur, 1, k2); UG_WAIT_VM(10, x3); }
            else if (n2) { stB(cb + SLOT_B0, cur, 0, k2); UG_WAIT_VM(8, x3); }    // A0, B0(kt+1) landed
            else if (n1) UG_WAIT_VM(4, x3);
            UG_PSTAMP(kt, 13);
            UG_BARRIER();
            UG_PSTAMP(kt, 14);
            UG_MMA_QUADRANT(1, 0, breg0);
            UG_PSTAMP(kt, 15);
            UG_BARRIER();
            UG_PSTAMP(kt, 16);
        };
        {
            constexpr auto GENERIC = std::integral_constant<int, 0>{};
            constexpr auto MID = std::integral_constant<int, 2>{};
            constexpr auto DYN = std::integral_constant<int, -1>{};
            constexpr auto P0 = std::integral_constant<int, 0>{};
            constexpr auto P1 = std::integral_constant<int, 1>{};
            int kt = 0;
            if constexpr (STEADY_OK) {
                if (steady_to >= steady_from) {
                    for (; kt < steady_from; ++kt) ktile(kt, GENERIC, DYN);
                    // pairs with their ring parity at compile time (steady_from is even): every LDS address is a lane constant + an immediate
                    for (; kt < steady_to; kt += 2) { ktile(kt, MID, P0); ktile(kt + 1, MID, P1); }
                }
            }
            for (; kt < nk; ++kt) ktile(kt, GENERIC, DYN);
        }
#undef UG_WAIT_VM
        if (wr == 0) UG_BARRIER();          // both groups level again; every LDS read of this tile has retired
        UG_STAMP(2);

        // next tile: addresses + first K-tile DMA, then this tile's epilogue runs under it
        const int64_t m0 = cur.m0, n0 = cur.n0;
        const int g = cur.g;
        const int rem = cur.rem, slice = cur.slice;
        stores_in_flight = false;
        // Everything lane-dependent below is re-derived from an opaque copy of the lane id: hoisted out of the tile loop those
        // values stayed live across the main loop and were spilled around it (scratch traffic + a vmcnt(0) ahead of the K loop).
        int lane_e = lane;
        asm volatile("" : "+v"(lane_e));
        const int tid_e = wave * 64 + lane_e;
        // Full tiles take the branch-free epilogue: this lane_e's bias (and, when all 256 rows belong to one sample, gate) columns are
        // fetched NOW, with nothing else on the VM counter, and waited for under the next tile's address arithmetic - before its DMA
        // prefetch is issued, so no later wait in the epilogue has to drain that prefetch.
        constexpr bool RES = EPI == UG_EPI_RES_GATE || EPI == UG_EPI_RES_SCALE;
        constexpr int EPI_A = EPI == UG_EPI_QKV_ROPE ? UG_EPI_BIAS_GELU : EPI;      // what a tile outside the q | k columns runs
        bool qk_tile = false;
        if constexpr (EPI == UG_EPI_QKV_ROPE) qk_tile = n0 < p.qk_until_n;
        bool fast = EPI != UG_EPI_F32 && wide16 && rows_contig && m0 + 256 <= M && n0 + 256 <= N && rem < 0;
        unsigned sample = 0;
        if constexpr (EPI == UG_EPI_RES_GATE) {
            sample = (unsigned)m0 / (unsigned)p.rows_per_sample;
            fast = fast && ((unsigned)m0 + 255u) / (unsigned)p.rows_per_sample == sample;
        }
        float fb[2][2][4] = {}, fg[2][2][4] = {};      // defined on every path: an undef phi became loop-carried and was spilled around the K loop
        if (fast) {
            const bf16_t* bias = p.bias ? (const bf16_t*)p.bias + (int64_t)g * p.bias_gstride : nullptr;
            const bf16_t* gate = nullptr;
            if constexpr (EPI == UG_EPI_RES_GATE) gate = (const bf16_t*)p.gate + (int64_t)g * p.gate_gstride + (int64_t)sample * p.gate_ld;
            u32x2 pb[2][2], pg[2][2];
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    const int64_t n = n0 + j * CJ + wc * CW + nt * 16 + (lane_e >> 4) * 4;
                    pb[j][nt] = (u32x2){0u, 0u}; pg[j][nt] = (u32x2){0u, 0u};
                    if (bias) pb[j][nt] = *(const u32x2*)(bias + n);
                    if constexpr (EPI == UG_EPI_RES_GATE) pg[j][nt] = *(const u32x2*)(gate + n);
                    if constexpr (EPI == UG_EPI_QKV_ROPE) {        // fg[0][nt]: this lane's 4 RMSNorm weights (column within the head)
                        if (qk_tile && j == 0)
                            pg[0][nt] = *(const u32x2*)((const bf16_t*)(2 * n0 >= p.qk_until_n ? p.qk_wk : p.qk_wq) + (QKDH == 64 ? (wc & 1) : wc) * 32 + nt * 16 +
                                                        (lane_e >> 4) * 4);
                    }
                }
            if (tile + (int)gridDim.x < n_items) cur = tile_src(tile + gridDim.x, lane_e); else cur.nk = 0;
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    fb[j][nt][0] = bflo(pb[j][nt].x); fb[j][nt][1] = bfhi(pb[j][nt].x); fb[j][nt][2] = bflo(pb[j][nt].y); fb[j][nt][3] = bfhi(pb[j][nt].y);
                    fg[j][nt][0] = bflo(pg[j][nt].x); fg[j][nt][1] = bfhi(pg[j][nt].x); fg[j][nt][2] = bflo(pg[j][nt].y); fg[j][nt][3] = bfhi(pg[j][nt].y);
                }
            __builtin_amdgcn_sched_barrier(0);
        } else {
            if (tile + (int)gridDim.x < n_items) cur = tile_src(tile + gridDim.x, lane_e); else cur.nk = 0;
        }
        if (cur.nk > 0) stage_first(cur);
        UG_STAMP(3);
        if (rem >= 0) {
            // ---- split-K tail: slab out, ticket, last arriver reduces (cdna guide section 5, "in-launch split-K reduction") ----
            float* my = slabs + ((size_t)rem * nslices + slice) * 65536;
            // UG_GEMM_SLAB_SC (round 6): the slabs travel at SYSTEM scope - `sc0 sc1` stores write through this XCD's L2, `sc0 sc1` loads never hit a
            // stale line of the reader's - instead of plain accesses bracketed by agent-scope fences. On this chip an agent release is `buffer_wbl2 sc1`
            // (write back EVERY dirty line of the XCD's L2: by every slice workgroup) and an acquire `buffer_inv sc1` (drop the L2's contents: under the
            // other workgroups' operand panels); with them the slab round trip measured ~46 us + 6 us per slice (profiles/r06j_shape_rates_cfg1_after_dispatch.log).
#if UG_GEMM_SLAB_SC
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int a = 0; a < 4; ++a)
#pragma unroll
                        for (int b = 0; b < 2; ++b)
                            asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(my + ((((i * 2 + j) * 4 + a) * 2 + b) * 512 + tid_e) * 4), "v"(acc[i][j][a][b]) : "memory");
#else
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int a = 0; a < 4; ++a)
#pragma unroll
                        for (int b = 0; b < 2; ++b)
                            *(f32x4*)(my + ((((i * 2 + j) * 4 + a) * 2 + b) * 512 + tid_e) * 4) = acc[i][j][a][b];
#endif
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // every storing wave: its slab part has reached memory
            __syncthreads();
            unsigned* flag = (unsigned*)(smem + LDS256_BYTES);
            if (tid_e == 0) {
#if !UG_GEMM_SLAB_SC
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // keep: hipcc may drop the fence's own wait
#endif
                *flag = __hip_atomic_fetch_add(tickets + rem, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            __syncthreads();
            const bool last = *flag == (unsigned)(nslices - 1);
            __syncthreads();                                              // flag is re-used by a later item
            if (!last) continue;
            if (tid_e == 0) {
#if !UG_GEMM_SLAB_SC
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
                __hip_atomic_store(tickets + rem, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // all arrivals are in: ready for the next launch
            }
            __syncthreads();
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int a = 0; a < 4; ++a)
#pragma unroll
                        for (int b = 0; b < 2; ++b) acc[i][j][a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
            for (int sl = 0; sl < nslices; ++sl) {
                const float* sp = slabs + ((size_t)rem * nslices + sl) * 65536;
#if UG_GEMM_SLAB_SC
                // system-scope loads, 8 in flight per lane (the accumulators leave no room for a whole slab's 32), same slice and element order as before
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        f32x4 t[4][2];
#pragma unroll
                        for (int a = 0; a < 4; ++a)
#pragma unroll
                            for (int b = 0; b < 2; ++b)
                                asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(t[a][b]) : "v"(sp + ((((i * 2 + j) * 4 + a) * 2 + b) * 512 + tid_e) * 4) : "memory");
                        asm volatile("s_waitcnt vmcnt(0)" : "+v"(t[0][0]), "+v"(t[0][1]), "+v"(t[1][0]), "+v"(t[1][1]), "+v"(t[2][0]), "+v"(t[2][1]), "+v"(t[3][0]), "+v"(t[3][1])::"memory");
#pragma unroll
                        for (int a = 0; a < 4; ++a)
#pragma unroll
                            for (int b = 0; b < 2; ++b) acc[i][j][a][b] += t[a][b];
                    }
#else
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int a = 0; a < 4; ++a)
#pragma unroll
                            for (int b = 0; b < 2; ++b)
                                acc[i][j][a][b] += *(const f32x4*)(sp + ((((i * 2 + j) * 4 + a) * 2 + b) * 512 + tid_e) * 4);
#endif
            }
        }
        if constexpr (EPI == UG_EPI_QKV_ROPE) {
            if (fast && qk_tile) {
                // q | k tile: the tile is two heads wide (j = 0, 1); a row of a head is spread over the 4 waves wc = 0..3 (32 columns each)
                // and, inside a wave, over 4 lane groups of 4 columns x 2 n-tiles. (QKDH = 64, SD3.5: four heads per tile, a head's row over the
                // wave pair wc & ~1; RoPE optional there - JointAttnProcessor2_0 has none.)
                const int lg = lane_e >> 4;
                const int wh = QKDH == 64 ? (wc & 1) : wc;                       // this wave's 32-column slice of its head
                const bool has_rope = QKDH == 128 || p.rope_cs != nullptr;
                float* const part = (float*)(smem + LDS256_BYTES + 16) + ((wr * 64 + (lane_e & 15)) * 8);   // [row 256][head 2][wc 4]
                // (1) the Linear's bf16 output, in place, and the per-row sums of squares: lane groups by permlane swaps, waves through LDS
#pragma unroll
                for (int rg = 0; rg < 8; ++rg)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        float ss = 0.f;
#pragma unroll
                        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                            for (int q = 0; q < 4; q += 2) {
                                float v0 = acc[rg >> 2][j][rg & 3][nt][q] + fb[j][nt][q], v1 = acc[rg >> 2][j][rg & 3][nt][q + 1] + fb[j][nt][q + 1];
                                rbf2(v0, v1);                                     // rounded in pairs (gemm_epilogue.h)
                                acc[rg >> 2][j][rg & 3][nt][q] = v0; acc[rg >> 2][j][rg & 3][nt][q + 1] = v1;
                                ss += v0 * v0; ss += v1 * v1;
                            }
                        part[((rg >> 2) * 128 + (rg & 3) * 16) * 8 + j * 4 + wc] = sum_row_groups(ss);
                    }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                UG_BARRIER();
                // (2) normalise, rotate, store. (cos, sin) pairs of row-group rg+1 are loaded before row-group rg is computed and stored.
                const int colb = (int)n0 + wc * 32 + (lg & 1) * 16 + 8 * (lg >> 1);
                bf16_t* const Cb = (bf16_t*)p.C + colb;
                const float* const csb = p.rope_cs + wh * 32 + lg * 4;
                const unsigned mrow = (unsigned)m0 + wr * 64 + (lane_e & 15);
                // position of a row: ONE division per tile (its divisor made opaque, or the reciprocal is hoisted out of the tile loop and
                // spilled across the K loop); the other 7 rows are < 256 <= rope_rpb further on, so a conditional subtract wraps them.
                unsigned rpb = (unsigned)p.rope_rpb;
                asm volatile("" : "+s"(rpb));
                const unsigned wrap = rpb ? rpb : 0xffffffffu;
                const unsigned rr0 = rpb ? mrow % rpb : mrow;
                u32x4 cbuf[2][2];
                bf16_t* cp[2];
                bf16_t* const c_lane = Cb + (int64_t)(rowmap32((unsigned)m0, (unsigned)p.c_rpb, (unsigned)p.c_bstride) + (unsigned)(wr * 64 + (lane_e & 15))) * p.ldc;
                auto open_rows = [&](int rg) {
                    cp[rg & 1] = c_lane + (int64_t)((rg >> 2) * 128 + (rg & 3) * 16) * p.ldc;      // rows_contig: one scalar row map per tile
                    unsigned rr = rr0 + (rg >> 2) * 128 + (rg & 3) * 16;
                    if (rr >= wrap) rr -= wrap;
                    const float* q = csb + ((int64_t)p.rope_pos0 + rr) * QKDH;
                    if (has_rope) {
                        cbuf[rg & 1][0] = gload16_asm(q);
                        cbuf[rg & 1][1] = gload16_asm_64(q);
                    }
                };
                cbuf[0][0] = cbuf[0][1] = cbuf[1][0] = cbuf[1][1] = (u32x4){0u, 0u, 0u, 0u};
                stores_in_flight = true;
                open_rows(0);
#pragma unroll
                for (int rg = 0; rg < 8; ++rg) {
                    if (rg + 1 < 8) open_rows(rg + 1);
                    if (has_rope) {
                        if (rg == 0 || rg == 7) ug_wait_vm<2>(cbuf[rg & 1][0], cbuf[rg & 1][1]);
                        else ug_wait_vm<4>(cbuf[rg & 1][0], cbuf[rg & 1][1]);
                    }
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const f32x4 t = *(const f32x4*)(part + ((rg >> 2) * 128 + (rg & 3) * 16) * 8 + j * 4);
                        const float tot = QKDH == 128 ? (t[0] + t[1] + t[2] + t[3]) : ((wc & 2) ? t[2] + t[3] : t[0] + t[1]);
                        const float rs = __builtin_amdgcn_rsqf(tot * (1.0f / (float)QKDH) + p.qk_eps);
                        unsigned pk[2][2];
#pragma unroll
                        for (int nt = 0; nt < 2; ++nt) {
                            const u32x4 cs = cbuf[rg & 1][nt];
                            const float c0 = __builtin_bit_cast(float, (unsigned)cs.x), s0 = __builtin_bit_cast(float, (unsigned)cs.y);
                            const float c1 = __builtin_bit_cast(float, (unsigned)cs.z), s1 = __builtin_bit_cast(float, (unsigned)cs.w);
                            float x[4];
#pragma unroll
                            for (int q = 0; q < 4; ++q) x[q] = acc[rg >> 2][j][rg & 3][nt][q] * rs;
                            rbf2(x[0], x[1]); rbf2(x[2], x[3]);                  // x * rsqrt as a bf16 tensor ...
#pragma unroll
                            for (int q = 0; q < 4; ++q) x[q] = x[q] * fg[0][nt][q];
                            rbf2(x[0], x[1]); rbf2(x[2], x[3]);                  // ... then * weight
                            if (has_rope) {
                                pk[nt][0] = pack2bf(x[0] * c0 + (-x[1]) * s0, x[1] * c0 + x[0] * s0);
                                pk[nt][1] = pack2bf(x[2] * c1 + (-x[3]) * s1, x[3] * c1 + x[2] * s1);
                            } else {
                                pk[nt][0] = pack2bf(x[0], x[1]);
                                pk[nt][1] = pack2bf(x[2], x[3]);
                            }
                        }
                        swap16(pk[0][0], pk[1][0]); swap16(pk[0][1], pk[1][1]);
                        u32x4 o; o.x = pk[0][0]; o.y = pk[0][1]; o.z = pk[1][0]; o.w = pk[1][1];
                        __builtin_nontemporal_store(o, (u32x4*)(cp[rg & 1] + j * 128));
                    }
                }
                UG_STAMP(4);
#ifdef UG_DIAG_STAMPS
                ++tile_seq;
#endif
                continue;
            }
        }
        if (fast) {
            // 8 row-groups of 16 rows x 2 column halves; the residual chunks of row-group rg+1 are loaded before row-group rg is
            // computed and stored (older than those stores on the in-order VM counter, so waiting for them does not wait for stores).
            const int lg = lane_e >> 4;
            const int colb = (int)n0 + wc * CW + (lg & 1) * 16 + 8 * (lg >> 1);
            const TileSplit ts = tile_split<EPI>(p, n0);
            bf16_t* const Cb = (bf16_t*)p.C + (int64_t)g * p.c_gstride + colb + ts.cshift;
            const bf16_t* const Rb = RES ? (const bf16_t*)p.R + (int64_t)g * p.r_gstride + colb : nullptr;
            // Row maps: ONE scalar map per tile (rows_contig: a tile's 256 rows are consecutive in C and in R), this lane's row pointer
            // once, then a wave-uniform offset per row-group. (Round 2 mapped every row-group's row with a 32-bit division per map and a
            // 64-bit multiply: ~45 of the ~190 VALU instructions per row-group of this VALU-bound epilogue.)
            const int rl = wr * 64 + (lane_e & 15);
            bf16_t* const c_lane = Cb + (int64_t)(rowmap32((unsigned)m0, (unsigned)p.c_rpb, (unsigned)p.c_bstride) + (unsigned)rl) * p.ldc;
            // NPERM: whole-line stores. Per 16-row group a lane (r = lane & 15, lg) holds o[0] / o[1] = its 16-byte chunk of the row's first / second
            // 64-byte half. Lanes r and r ^ 8 trade through DPP row_ror:8 (same lg, so the chunk position inside a half is unchanged):
            //   store A = rows 0-7:  lanes r < 8 their own o[0] at half 0 | lanes r >= 8 row (r - 8)'s o[1] at half 1
            //   store B = rows 8-15: lanes r >= 8 their own o[0] at half 0 | lanes r < 8 row (r + 8)'s o[1] at half 1
            // -> each instruction writes 8 rows x 128 contiguous bytes. (R may alias C: every residual chunk of the row group was loaded AND waited for
            // by this wave before its first store, and no other wave touches these rows' columns.)
            const int r16 = lane_e & 15;
            bf16_t* const c_laneA = c_lane + (int64_t)((r16 & 7) - r16) * p.ldc + (r16 >> 3) * 32;
            const int dB = 8 * (int)p.ldc + (r16 < 8 ? 32 : -32);      // store B's address = store A's + 8 rows, other half (elements; ldc < 2^27 checked by the launcher's 31-bit row limits)
            const bf16_t* r_lane = nullptr;
            if constexpr (RES) r_lane = Rb + (int64_t)(rowmap32((unsigned)m0, (unsigned)p.r_rpb, (unsigned)p.r_bstride) + (unsigned)rl) * p.ldr;
            // ... and whole-line residual LOADS the same way (half-line loads issue at 20 B/clk per CU, whole lines at 50: tools/probe/store_rate.hip):
            // load A = rows 0-7 (lanes r >= 8 fetch row (r - 8)'s second half), load B = rows 8-15 (lanes r < 8 fetch row (r + 8)'s second half); the two
            // loads are turned back into this lane's own (first-half, second-half) chunks right before use (res_own below).
            const bf16_t* r_laneA = nullptr; int dBr = 0;
            if constexpr (RES && NPERM) { r_laneA = r_lane + (int64_t)((r16 & 7) - r16) * p.ldr + (r16 >> 3) * 32; dBr = 8 * (int)p.ldr + (r16 < 8 ? 32 : -32); }
            // Residual chunks are requested PF row-groups ahead (round 3: all 8, was 2). In-kernel stamps (tools/gemm_stamps.py) showed the R + gate * v
            // epilogue at 8.3 us per tile against 3.7 us for bias only at equal stores: with one row-group of lead every row-group waited out a
            // full memory latency behind the next tile's 16 ring DMAs. The fragment registers of the K loop are dead here, so 8 x 8 registers are free; what the stamps also show: wave 0 finishes earlier (8.3 -> 7.0 us) but then waits longer at the next tile's first barrier - the workgroup's epilogue is bound by the CU's memory path (ring prefetch + R + C = 384 KB), not by one wave's latency chain.
            // (RES_GATE with whole-line stores: 7 - the eighth row-group's 8 registers are what the two extra store pointers and the DPP temporaries need; at 8
            // hipcc spilled one register of the edge-tile path to scratch)
            constexpr int PF = (EPI == UG_EPI_RES_GATE && NPERM && UG_EPI_RES_PREFETCH > 7) ? 7 : UG_EPI_RES_PREFETCH;
            u32x4 rbuf[PF][2];
            auto open_rows = [&](int rg) {
                if constexpr (RES && NPERM) {
                    const bf16_t* rp = r_laneA + (int64_t)((rg >> 2) * 128 + (rg & 3) * 16) * p.ldr;
                    rbuf[rg % PF][0] = gload16_asm(rp);                 // load A: 8 rows x 128 B
                    rbuf[rg % PF][1] = gload16_asm(rp + dBr);           // load B
                } else if constexpr (RES) {
                    const bf16_t* rp = r_lane + (int64_t)((rg >> 2) * 128 + (rg & 3) * 16) * p.ldr;
                    rbuf[rg % PF][0] = gload16_asm(rp);
                    rbuf[rg % PF][1] = gload16_asm_256(rp);
                } else {
                    rbuf[rg % PF][0] = rbuf[rg % PF][1] = (u32x4){0u, 0u, 0u, 0u};
                }
            };
            stores_in_flight = true;
#pragma unroll
            for (int rg = 0; rg < PF - 1; ++rg) open_rows(rg);
#pragma unroll
            for (int rg = 0; rg < 8; ++rg) {
                if (rg + PF - 1 < 8) open_rows(rg + PF - 1);
                if constexpr (RES) {
                    // younger than this row-group's two loads on the in-order VM counter: the loads of the next min(PF - 1, 7 - rg) row-groups
                    // and the stores of the previous min(PF - 1, rg) ones
                    constexpr int LA = PF - 1;
                    const int younger = 2 * ((7 - rg) < LA ? (7 - rg) : LA) + 2 * (rg < LA ? rg : LA);
                    switch (younger) {
                        case 2: ug_wait_vm<2>(rbuf[rg % PF][0], rbuf[rg % PF][1]); break;
                        case 4: ug_wait_vm<4>(rbuf[rg % PF][0], rbuf[rg % PF][1]); break;
                        case 6: ug_wait_vm<6>(rbuf[rg % PF][0], rbuf[rg % PF][1]); break;
                        case 8: ug_wait_vm<8>(rbuf[rg % PF][0], rbuf[rg % PF][1]); break;
                        case 10: ug_wait_vm<10>(rbuf[rg % PF][0], rbuf[rg % PF][1]); break;
                        case 12: ug_wait_vm<12>(rbuf[rg % PF][0], rbuf[rg % PF][1]); break;
                        default: ug_wait_vm<14>(rbuf[rg % PF][0], rbuf[rg % PF][1]); break;
                    }
                }
                const int64_t rgoff = (int64_t)((rg >> 2) * 128 + (rg & 3) * 16) * p.ldc;
                u32x4 res_own[2] = {rbuf[rg % PF][0], rbuf[rg % PF][1]};
                if constexpr (RES && NPERM) {
                    // loads A / B -> this lane's own chunks: first half = (r < 8 ? A : B); second half = row_ror:8 of (r >= 8 ? A : B)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int la = (int)rbuf[rg % PF][0][q], lb = (int)rbuf[rg % PF][1][q];
                        res_own[0][q] = (unsigned)__builtin_amdgcn_update_dpp(la, lb, 0xE4, 0xF, 0xC, false);          // lanes 8-15 of every row take B (same lane: quad_perm identity)
                        const int z = __builtin_amdgcn_update_dpp(lb, la, 0xE4, 0xF, 0xC, false);                         // lanes 8-15 take A, lanes 0-7 keep B
                        res_own[1][q] = (unsigned)__builtin_amdgcn_update_dpp(z, z, 0x128, 0xF, 0xF, false);              // row_ror:8
                    }
                }
                u32x4 o[2];
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    o[j] = (EPI_A == UG_EPI_BIAS_GELU && !ts.gelu)
                        ? epi_chunk_full<UG_EPI_BIAS>(p.alpha, acc[rg >> 2][j][rg & 3][0], acc[rg >> 2][j][rg & 3][1], fb[j][0], fb[j][1],
                                                      fg[j][0], fg[j][1], res_own[j])
                        : epi_chunk_full<EPI_A>(p.alpha, acc[rg >> 2][j][rg & 3][0], acc[rg >> 2][j][rg & 3][1], fb[j][0], fb[j][1],
                                              fg[j][0], fg[j][1], res_own[j]);
                if constexpr (NPERM) {
                    u32x4 xa, xb;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        // v_mov_b32_dpp row_ror:8, bank_mask selects which lanes of every 16-lane row take the neighbour's value (banks 2,3 = lanes 8-15; 0,1 = lanes 0-7)
                        xa[q] = (unsigned)__builtin_amdgcn_update_dpp((int)o[0][q], (int)o[1][q], 0x128, 0xF, 0xC, false);
                        xb[q] = (unsigned)__builtin_amdgcn_update_dpp((int)o[0][q], (int)o[1][q], 0x128, 0xF, 0x3, false);
                    }
                    __builtin_nontemporal_store(xa, (u32x4*)(c_laneA + rgoff));
                    __builtin_nontemporal_store(xb, (u32x4*)(c_laneA + rgoff + dB));
                } else {
                    __builtin_nontemporal_store(o[0], (u32x4*)(c_lane + rgoff));
                    __builtin_nontemporal_store(o[1], (u32x4*)(c_lane + rgoff + 128));
                }
            }
            UG_STAMP(4);
#ifdef UG_DIAG_STAMPS
            ++tile_seq;
#endif
            continue;
        }
        const bf16_t* bias = p.bias ? (const bf16_t*)p.bias + (int64_t)g * p.bias_gstride : nullptr;
        float bv[2][2][4];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                const int64_t n = n0 + j * CJ + wc * CW + nt * 16 + (lane_e >> 4) * 4;
                load_bias4(n < N ? bias : nullptr, n, bv[j][nt]);
            }
        const TileSplit tsl = tile_split<EPI>(p, n0);
        if (EPI != UG_EPI_F32 && wide16) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
                    const int64_t m = m0 + i * 128 + wr * 64 + mt * 16 + (lane_e & 15);
                    const bool row_ok = m < M;                         // lanes l and l^16 share the row: the swaps stay paired
                    RowCtx rc = row_ctx<EPI_A>(p, g, (unsigned)(row_ok ? m : M - 1));
                    rc.coff += tsl.cshift;
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        if (EPI_A == UG_EPI_BIAS_GELU && !tsl.gelu)
                            epi_store_pair16<UG_EPI_BIAS>(p, rc, row_ok, n0 + j * CJ + wc * CW, N, lane_e, acc[i][j][mt][0], acc[i][j][mt][1], bv[j][0], bv[j][1]);
                        else
                            epi_store_pair16<EPI_A>(p, rc, row_ok, n0 + j * CJ + wc * CW, N, lane_e, acc[i][j][mt][0], acc[i][j][mt][1], bv[j][0], bv[j][1]);
                    }
                }
        } else {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
                    const int64_t m = m0 + i * 128 + wr * 64 + mt * 16 + (lane_e & 15);
                    if (m >= M) continue;
                    RowCtx rc = row_ctx<EPI_A>(p, g, (unsigned)m);
                    rc.coff += tsl.cshift;
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int nt = 0; nt < 2; ++nt) {
                            const int64_t n = n0 + j * CJ + wc * CW + nt * 16 + (lane_e >> 4) * 4;
                            if (n >= N) continue;
                            if (EPI_A == UG_EPI_BIAS_GELU && !tsl.gelu) epi_store<UG_EPI_BIAS>(p, rc, n, acc[i][j][mt][nt], bv[j][nt]);
                            else epi_store<EPI_A>(p, rc, n, acc[i][j][mt][nt], bv[j][nt]);
                        }
                }
        }
    }
#ifdef UG_DIAG_STAMPS
    // flush: workgroup b's stamps at slabs[(b * 50 + seq) * 5 + point] as 64-bit ticks (slabs points at the caller's workspace + 4096)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (threadIdx.x == 0 && slabs != nullptr) {
        unsigned long long* out = (unsigned long long*)slabs + (size_t)blockIdx.x * 250;
        for (int i = 0; i < 248; ++i) out[i] = i < tile_seq * 5 ? stamp_lds[i] : 0ull;
        out[248] = stamp_lds[250]; out[249] = __builtin_amdgcn_s_memrealtime();    // entry / exit on the chip-wide real-time counter (per-XCD finish times)
    }
#ifdef UG_DIAG_PHASES
    if (threadIdx.x == 0 && slabs != nullptr) {      // phase stamps of waves 0 and 4 behind the tile stamps of all 256 workgroups
        unsigned long long* out = (unsigned long long*)slabs + (size_t)256 * 250 + (size_t)blockIdx.x * 64;
        for (int i = 0; i < 64; ++i) out[i] = stamp_lds[256 + i];
    }
#endif
#endif
#undef UG_STAMP
#undef UG_PSTAMP
#undef UG_MMA_QUADRANT
#undef UG_BARRIER
}

// UG_GEMM_FORCE_TILE=128|256 pins the kernel choice (tests / A-B timing); default: 256^2 tiles when they fill the chip.
int forced_tile() { return ug_env_int("UG_GEMM_FORCE_TILE", 0); }

template <int EPI>
int launch(const ug_gemm_desc& d, hipStream_t s) {
    const int groups = d.groups > 0 ? d.groups : 1;
    const int64_t t256 = ((d.M + 255) / 256) * ((d.N + 255) / 256) * groups;
    const int64_t t128 = ((d.M + 127) / 128) * ((d.N + 127) / 128) * groups;
    // pick the tile by expected chip fill: 256 CUs x 1 workgroup (256^2) vs 256 x 2 (128^2, whose rate relative to the 256^2 kernel is the
    // weight below: 0.82 on large shapes, but in the cfg2 forward the text-stream projections (M = B x 512) run better on 256^2 tiles:
    // weight 60 % -> 2.011 images/s, 70 %: 2.004-2.008, 82 % (rounds 1-2): 1.995-2.000, 50 %: 2.010)
    const double e256 = (double)t256 / (double)(((t256 + 255) / 256) * 256);
    const double e128 = 0.01 * UG_TUNE("UG_GEMM_E128_PCT", 60) * (double)t128 / (double)(((t128 + 511) / 512) * 512);
    const bool lora = d.lora_r > 0;
    bool big = (!lora || (d.K >= 2 * BK && EPI != UG_EPI_F32)) && d.M >= 192 && d.N >= 192 && e256 >= e128;
    // Round 6 (small-M regime: the reference's own launch shape is batch 1, script/infer.sh:62-63): a FEW tiles with a LONG K loop - the text-stream and
    // 512^2 forms of ff.net.2 (K = 12288) and of the single blocks' proj_out (K = 15360): 24-72 tiles of 256^2 - lost the fill comparison above to the
    // 128^2 kernel, which then ran one round of whole K loops (512 x 3072 x 12288: 187 us = 207 TFLOP/s, profiles/r06i_shape_rates_cfg1.log) although
    // the 256^2 kernel would cut exactly these launches into K-slices that fill the chip (its split-K tail). Priced in microseconds with the measured
    // unit costs - a 256^2 K-tile 1.53 us, a 128^2 K-tile 0.97 us with one workgroup on the CU / 1.27 us with two, the slab round trip 22 + 2.5 us per slice (46 + 6 before the slabs went to system scope, below):
    if (!big && !lora && EPI != UG_EPI_F32 && d.M >= 192 && d.N >= 192 && d.workspace && ug_aligned(d.workspace, 16) && UG_TUNE("UG_GEMM_SPLITK_TAIL", 1) &&
        UG_TUNE("UG_GEMM_SPLITK_SMALLM", 1)) {
        const int G = 256, nkt = (int)(d.K / BK);
        const int rem = (int)(t256 % G);
        if (t256 < G && rem * 2 <= G && nkt >= UG_TUNE("UG_GEMM_SPLITK_MIN_KT", UG_GEMM_SPLITK_MIN_KT_DEFAULT)) {
            const int rem8 = (rem + 7) / 8 * 8;
            int cand = G / rem8; if (cand > 8) cand = 8; if (cand > nkt / 4) cand = nkt / 4;
            if (cand >= 2 && (size_t)d.workspace_bytes >= 4096 + (size_t)rem8 * cand * 65536 * sizeof(float)) {
                const double t_split = 1.53 * nkt / cand + (UG_GEMM_SLAB_SC ? 22.0 + 2.5 * cand : 46.0 + 6.0 * cand);     // slab round trip as measured with / without the system-scope slabs
                const double t_128 = (double)((t128 + 511) / 512) * nkt * (t128 <= G ? 0.97 : 1.27);
                if (t_split < 0.9 * t_128) big = true;
            }
        }
    }
    const int f = forced_tile();
    if (f == 128) big = false;
    if (f == 256 && (!lora || (d.K >= 2 * BK && EPI != UG_EPI_F32))) big = true;
    // the 256^2 kernel's buffer-form DMAs carry UNSIGNED 32-bit byte offsets relative to the tile's first row: the A row map must be monotonic
    // (batch stride >= rows per batch; a broadcasting map such as RowMap(N, 0) goes to the 128^2 kernel, which keeps per-lane pointers), and the
    // offset spans 255 rows plus one batch jump per batch boundary inside the tile: ceil(255 / rows_per_batch) of them
    if (!lora && d.a_rpb > 0 && d.a_bstride < d.a_rpb) big = false;
    const int64_t a_jumps = d.a_rpb > 0 ? (255 + d.a_rpb - 1) / d.a_rpb : 0;
    const int64_t a_jump = d.a_rpb > 0 && d.a_bstride > d.a_rpb ? (d.a_bstride - d.a_rpb) * a_jumps : 0;
    if (!lora && ((255 + a_jump) * d.lda + d.K) * 2 + 256 >= (int64_t)1 << 31) big = false;
    if (!lora && (255 * d.ldw + d.K) * 2 + 256 >= (int64_t)1 << 31) big = false;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)gemm128_kernel<EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        (void)hipFuncSetAttribute((const void*)gemm128_kernel<EPI, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        (void)hipFuncSetAttribute((const void*)gemm256_kernel<EPI, false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS256_BYTES + 16 + UG_STAMP_LDS);
        if (EPI != UG_EPI_F32)
            (void)hipFuncSetAttribute((const void*)gemm256_kernel<EPI, EPI != UG_EPI_F32>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS256_BYTES + 16 + UG_STAMP_LDS);
        attr_set = true;
    }
    if (big) {
        static int ncu = 0;
        if (ncu == 0) {
            int dev = 0; hipDeviceProp_t prop;
            if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ncu = prop.multiProcessorCount;
            if (ncu <= 0) ncu = 256;
        }
        const int total = (int)t256;
        dim3 grid((unsigned)(total < ncu ? total : ncu), 1, 1);
        // 16-byte epilogue accesses need 8-column granularity everywhere the epilogue touches
        const bool res = d.epilogue == UG_EPI_RES_GATE || d.epilogue == UG_EPI_RES_SCALE;
        const int wide16 = (d.N % 8 == 0 && d.ldc % 8 == 0 && d.c_gstride % 8 == 0 && ug_aligned(d.C, 16) &&
                            (!res || (d.ldr % 8 == 0 && d.r_gstride % 8 == 0 && ug_aligned(d.R, 16)))) |
                           ((d.c_rpb % 256 == 0 && (!res || d.r_rpb % 256 == 0) && UG_TUNE("UG_GEMM_EPI_ROWS_CONTIG", 1)) ? 2 : 0);
        // (Measured and dropped, round 3: plain instead of non-temporal C stores in the full-tile epilogue - +-0.5 % on every cfg2 shape,
        // profiles/r03d_gemm_cplain.log: the per-tile store cost is not the cache policy.)
        // UG_GEMM_PWG=1: the one-wave-per-SIMD kernel (gemm_pwg.hip) takes every shape it supports
#ifdef UG_PROBE_BUILD
        if (EPI != UG_EPI_F32 && (wide16 & 1) && !lora && ug_env_int("UG_GEMM_PWG", 0) && d.K <= ug_env_int("UG_PWG_MAXK", 1 << 30) &&
            (ug_env_int("UG_GEMM_PWG", 0) != 4 || (d.M % 256 == 0 && d.N % 256 == 0 && d.K % 128 == 0 && d.K >= 256 && d.a_rpb % 256 == 0 && total >= ncu)))
            return ug_gemm_launch_pwg(d, s);
#endif
        // split-K tail (see the kernel header): needs the caller's workspace for the slabs and tickets
        int full = total, nsl = 1;
        // M-tiles per group of the tile walk. In the cfg2 forward (same box, bench.py x 2 each): 4 -> 2.014 images/s / GEMM 1337 TFLOP/s,
        // 8 (rounds 1-2) 1.991-2.004 / 1322-1330, 6: 2.006, 3 / 5 / 2: 2.002-2.004, 16: 1.965, 32: 1.905 (profiles/r02c_group_m.log)
        const int gm = ((UG_TUNE("UG_GEMM_GROUP_M", 4) & 0xff) << 8) | ((UG_TUNE("UG_GEMM_WALK", 0) & 3) << 16);
        float* slabs = nullptr; unsigned* tickets = nullptr;
        const int G = ncu, rem = total % G;            // total < ncu: every tile is a remainder tile
        const int nkt = (int)(d.K / BK);
        const int split_on = UG_TUNE("UG_GEMM_SPLITK_TAIL", 1);
        // Measured (MI355X): the slab round trip + fences cost ~35 us, so the split only pays when a tile's K loop is long
        // (K = 15360 single-block proj_out: +3.5 %; K = 3072 shapes: -2...-3 %) -> require >= 96 K-tiles.
        if (split_on && !lora && rem > 0 && rem * 2 <= G && d.workspace && nkt >= UG_TUNE("UG_GEMM_SPLITK_MIN_KT", UG_GEMM_SPLITK_MIN_KT_DEFAULT)) {
            const int rem8 = (rem + 7) / 8 * 8;
            int cand = G / rem8; if (cand > 8) cand = 8; if (cand > nkt / 4) cand = nkt / 4;
            const size_t need = 4096 + (size_t)rem8 * cand * 65536 * sizeof(float);
            if (cand >= 2 && (size_t)d.workspace_bytes >= need && ug_aligned(d.workspace, 16)) {
                nsl = cand; full = total - rem;
                if (total < ncu) grid.x = (unsigned)(rem8 * cand);      // one workgroup per K-slice (incl. the 8-alignment padding)
                tickets = (unsigned*)d.workspace;
                slabs = (float*)((char*)d.workspace + 4096);
            }
        }
        if (lora) {     // (EPI_F32 never gets here with LoRA; its second instantiation is the plain kernel again)
            hipLaunchKernelGGL((gemm256_kernel<EPI, EPI != UG_EPI_F32>), grid, dim3(512), LDS256_BYTES + 16 + UG_STAMP_LDS, s, d, (int)(t256 / groups), total, wide16 | gm, full, nsl, slabs, tickets, UgConvGeom{});
        } else {
#ifdef UG_DIAG_STAMPS
            if (nsl == 1 && d.workspace && (size_t)d.workspace_bytes >= 4096 + (size_t)256 * 250 * 8 + (size_t)256 * 64 * 8) slabs = (float*)((char*)d.workspace + 4096);
#endif
            hipLaunchKernelGGL((gemm256_kernel<EPI, false>), grid, dim3(512), LDS256_BYTES + 16 + UG_STAMP_LDS, s, d, (int)(t256 / groups), total, wide16 | gm, full, nsl, slabs, tickets, UgConvGeom{});
        }
    } else {
        const int nM = (int)((d.M + BM - 1) / BM), nN = (int)((d.N + BN - 1) / BN);
        dim3 grid((unsigned)(nM * nN), 1, (unsigned)groups);
        // UG_GEMM128_W8 (build-time): 0 = the 4-wave workgroup everywhere, 1 = the 8-wave form for launches of at most one tile per CU, 2 = everywhere
        if (UG_GEMM128_W8 == 2 || (UG_GEMM128_W8 == 1 && (int64_t)nM * nN * groups <= 256))
            hipLaunchKernelGGL((gemm128_kernel<EPI, 8>), grid, dim3(512), LDS_BYTES, s, d);
        else
            hipLaunchKernelGGL(gemm128_kernel<EPI>, grid, dim3(256), LDS_BYTES, s, d);
    }
    UG_CHECK_LAUNCH("ug_gemm_bf16");
    return UG_OK;
}

// UG_EPI_QKV_ROPE: 256^2 kernel only, whole tiles only (the q | k epilogue reduces rows across the four column waves of a tile)
int launch_qkrope(const ug_gemm_desc& d, hipStream_t s) {
    UG_REQUIRE(d.M % 256 == 0 && d.N % 256 == 0 && d.qk_until_n > 0 && d.qk_until_n % 256 == 0 && d.qk_until_n <= d.N, UG_ERR_UNSUPPORTED,
               "ug_gemm_bf16: UG_EPI_QKV_ROPE needs M, N, qk_until_n multiples of 256 (M=%lld N=%lld qk_until_n=%lld)", (long long)d.M,
               (long long)d.N, (long long)d.qk_until_n);
    UG_REQUIRE(d.groups == 1 && d.lora_r <= 0, UG_ERR_UNSUPPORTED, "ug_gemm_bf16: UG_EPI_QKV_ROPE is neither grouped nor LoRA-extended");
    const int qdh = d.qk_dh == 0 ? 128 : d.qk_dh;
    UG_REQUIRE(qdh == 128 || qdh == 64, UG_ERR_UNSUPPORTED, "ug_gemm_bf16: UG_EPI_QKV_ROPE head width %d not in {64, 128}", qdh);
    UG_REQUIRE(d.qk_wq && d.qk_wk && (d.rope_cs || qdh == 64) && ug_aligned(d.qk_wq, 8) && ug_aligned(d.qk_wk, 8) && ug_aligned(d.rope_cs, 16) &&
               d.ldc % 8 == 0 && ug_aligned(d.C, 16), UG_ERR_BAD_ALIGN, "ug_gemm_bf16: UG_EPI_QKV_ROPE operands missing/misaligned");
    UG_REQUIRE((d.rope_rpb == 0 || d.rope_rpb >= 256) && d.rope_rpb < (1ll << 31) && d.rope_pos0 >= 0 && d.rope_pos0 < (1ll << 31) &&
               (d.gelu_from_n == 0 || d.gelu_from_n >= d.qk_until_n) && (d.c_shift_from_n == 0 || d.c_shift_from_n >= d.qk_until_n),
               UG_ERR_BAD_SHAPE, "ug_gemm_bf16: UG_EPI_QKV_ROPE bad positions / column split");
    constexpr int LDS = LDS256_BYTES + 16 + 256 * 8 * 4 + UG_STAMP_LDS;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)gemm256_kernel<UG_EPI_QKV_ROPE, false, 128>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        (void)hipFuncSetAttribute((const void*)gemm256_kernel<UG_EPI_QKV_ROPE, false, 64>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        attr_set = true;
    }
    static int ncu = 0;
    if (ncu == 0) {
        int dev = 0; hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ncu = prop.multiProcessorCount;
        if (ncu <= 0) ncu = 256;
    }
    const int total = (int)((d.M / 256) * (d.N / 256));
    {   // this epilogue only exists in the 256^2 kernel: its buffer-form A offsets need a monotonic row map and < 2^31 bytes per tile (see launch())
        const int64_t a_jumps = d.a_rpb > 0 ? (255 + d.a_rpb - 1) / d.a_rpb : 0;
        const int64_t a_jump = d.a_rpb > 0 && d.a_bstride > d.a_rpb ? (d.a_bstride - d.a_rpb) * a_jumps : 0;
        UG_REQUIRE((d.a_rpb == 0 || d.a_bstride >= d.a_rpb) && ((255 + a_jump) * d.lda + d.K) * 2 + 256 < ((int64_t)1 << 31) &&
                   (255 * d.ldw + d.K) * 2 + 256 < ((int64_t)1 << 31), UG_ERR_BAD_SHAPE,
                   "ug_gemm_bf16: UG_EPI_QKV_ROPE needs a monotonic A row map (batch stride %lld >= rows per batch %lld) and tiles below 2^31 bytes",
                   (long long)d.a_bstride, (long long)d.a_rpb);
    }
    UG_REQUIRE(d.c_rpb % 256 == 0, UG_ERR_UNSUPPORTED, "ug_gemm_bf16: UG_EPI_QKV_ROPE needs the C row map's rows per batch (%lld) to be a multiple of 256",
               (long long)d.c_rpb);
#ifdef UG_PROBE_BUILD
    // UG_GEMM_PWG=4: the one-wave-per-SIMD probe kernel's port of this epilogue (head width 128 with RoPE, whole rounds, K a multiple of 128)
    if (ug_env_int("UG_GEMM_PWG", 0) == 4 && qdh == 128 && d.rope_cs && d.K % 128 == 0 && d.K >= 256 && d.a_rpb % 256 == 0 && total >= ncu &&
        d.K <= ug_env_int("UG_PWG_MAXK", 1 << 30))
        return ug_gemm_launch_pwg2_qkrope(d, s);
#endif
    const int wgm = 3 | ((UG_TUNE("UG_GEMM_GROUP_M", 4) & 0xff) << 8) | ((UG_TUNE("UG_GEMM_WALK", 0) & 3) << 16);
    const dim3 grid((unsigned)(total < ncu ? total : ncu));
    float* stamps = nullptr;
#ifdef UG_DIAG_STAMPS
    if (d.workspace && (size_t)d.workspace_bytes >= 4096 + (size_t)256 * 250 * 8) stamps = (float*)((char*)d.workspace + 4096);
#endif
    if (qdh == 128)
        hipLaunchKernelGGL((gemm256_kernel<UG_EPI_QKV_ROPE, false, 128>), grid, dim3(512), LDS, s, d, total, total, wgm, total, 1, stamps, (unsigned*)nullptr, UgConvGeom{});
    else
        hipLaunchKernelGGL((gemm256_kernel<UG_EPI_QKV_ROPE, false, 64>), grid, dim3(512), LDS, s, d, total, total, wgm, total, 1, stamps, (unsigned*)nullptr, UgConvGeom{});
    UG_CHECK_LAUNCH("ug_gemm_bf16");
    return UG_OK;
}

}  // namespace

// ug_conv2d_nhwc on the 256^2 kernel (see UgConvGeom): whole 256^2 tiles, bias or residual epilogue, no split-K tail (no workspace at this boundary).
int ug_gemm_launch_conv256(const ug_gemm_desc& d, const UgConvGeom& cv, hipStream_t s) {
    if (d.M % 256 || d.N % 256 || d.K % BK || cv.ktp < 2 || (cv.ktp & (cv.ktp - 1)) || d.K / BK < 3) return UG_ERR_UNSUPPORTED;
    static int ncu = 0;
    if (ncu == 0) {
        int dev = 0; hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ncu = prop.multiProcessorCount;
        if (ncu <= 0) ncu = 256;
    }
    constexpr int LDS = LDS256_BYTES + 16 + UG_STAMP_LDS;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)gemm256_kernel<UG_EPI_BIAS, false, 128, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        (void)hipFuncSetAttribute((const void*)gemm256_kernel<UG_EPI_RES_SCALE, false, 128, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        attr_set = true;
    }
    const int total = (int)((d.M / 256) * (d.N / 256));
    const int wgm = 3 | ((UG_TUNE("UG_GEMM_GROUP_M", 4) & 0xff) << 8);       // 16-byte epilogue accesses, one row map per tile (C and R are plain [M][N])
    const dim3 grid((unsigned)(total < ncu ? total : ncu));
    if (d.epilogue == UG_EPI_RES_SCALE)
        hipLaunchKernelGGL((gemm256_kernel<UG_EPI_RES_SCALE, false, 128, true>), grid, dim3(512), LDS, s, d, total, total, wgm, total, 1, (float*)nullptr, (unsigned*)nullptr, cv);
    else
        hipLaunchKernelGGL((gemm256_kernel<UG_EPI_BIAS, false, 128, true>), grid, dim3(512), LDS, s, d, total, total, wgm, total, 1, (float*)nullptr, (unsigned*)nullptr, cv);
    UG_CHECK_LAUNCH("ug_conv2d_nhwc(256)");
    return UG_OK;
}

extern "C" int64_t ug_gemm_workspace_bytes(void) { return 4096 + (int64_t)256 * 65536 * (int64_t)sizeof(float); }

extern "C" int ug_gemm_bf16(const ug_gemm_desc* dp, ug_stream_t stream) {
    UG_REQUIRE(dp != nullptr, UG_ERR_BAD_SHAPE, "ug_gemm_bf16: null descriptor");
    ug_gemm_desc d = *dp;
    if (d.groups <= 0) d.groups = 1;
    UG_REQUIRE(d.M >= 0 && d.N > 0 && d.K > 0, UG_ERR_BAD_SHAPE, "ug_gemm_bf16: bad M/N/K %lld/%lld/%lld",
               (long long)d.M, (long long)d.N, (long long)d.K);
    if (d.M == 0) return UG_OK;
    UG_REQUIRE(d.M < (1ll << 31) && d.N < (1ll << 31) && d.a_bstride < (1ll << 31) && d.c_bstride < (1ll << 31) && d.r_bstride < (1ll << 31) &&
               d.a_rpb < (1ll << 31) && d.c_rpb < (1ll << 31) && d.r_rpb < (1ll << 31) && d.rows_per_sample < (1ll << 31),
               UG_ERR_UNSUPPORTED, "ug_gemm_bf16: row counts must fit 31 bits");
    UG_REQUIRE(d.K % BK == 0, UG_ERR_UNSUPPORTED, "ug_gemm_bf16: K=%lld must be a multiple of %d", (long long)d.K, BK);
    UG_REQUIRE(d.N % 4 == 0, UG_ERR_UNSUPPORTED, "ug_gemm_bf16: N=%lld must be a multiple of 4", (long long)d.N);
    UG_REQUIRE(d.gelu_from_n >= 0 && d.c_shift_from_n >= 0 && d.gelu_from_n % 256 == 0 && d.c_shift_from_n % 256 == 0 && d.c_shift % 8 == 0 &&
               (d.c_shift_from_n > 0 || d.c_shift == 0) && d.ldc >= d.N + (d.c_shift > 0 ? d.c_shift : 0),
               UG_ERR_BAD_SHAPE, "ug_gemm_bf16: column split needs gelu_from_n, c_shift_from_n multiples of 256, c_shift a multiple of 8 and ldc >= N + c_shift");
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
        UG_REQUIRE(d.R && d.ldr % 4 == 0 && ug_aligned(d.R, 8) && d.r_gstride % 4 == 0, UG_ERR_BAD_ALIGN, "ug_gemm_bf16: residual missing/misaligned");
    }
    if (d.epilogue == UG_EPI_RES_GATE)
        UG_REQUIRE(d.gate && d.rows_per_sample > 0 && d.gate_ld % 4 == 0 && ug_aligned(d.gate, 8) && d.gate_gstride % 4 == 0, UG_ERR_BAD_SHAPE,
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
        case UG_EPI_QKV_ROPE: return launch_qkrope(d, s);
        default: UG_FAIL(UG_ERR_UNSUPPORTED, "ug_gemm_bf16: unknown epilogue %d", d.epilogue);
    }
}
