// Flash-style fused attention forward for the joint text|image|condition sequences of the UniGen MM-DiT blocks.
//   O = softmax(Q K^T * scale) V, non-causal, no mask, bf16 in/out, fp32 scores / statistics / accumulators.
// Replaces F.scaled_dot_product_attention at src/UniGenUtils.py:601 (JointAttnRopeProcessor) and inside diffusers
// FluxAttnProcessor2_0 (base blocks, src/UniGenTransformer.py:1129,1151). L = 4608 / 8192 / 8704 at 1024^2.
//
// Structure (gfx950, wave64): one workgroup = 8 waves = 256 query rows of one (batch, head); each wave owns 32 query
// rows, Q fragments live in registers. K/V tiles of 64 keys are staged HBM -> registers -> LDS (issue early, write late),
// double buffered, in an XOR-swizzled 256-byte-row image that is conflict-free for both the row reads (K, ds_read_b128)
// and the transposed reads (V, ds_read_b64_tr_b16).
//   S^T = K Q^T   with v_mfma_f32_32x32x16_bf16: the query index lands on the LANE, so the softmax row statistics are
//                 lane-local (one exchange with lane^32 per tile for the max).
//   O^T = V^T P^T : the S^T accumulator registers 8s..8s+7, packed to bf16, ARE the B operand of k-step s (permuted k
//                 order, matched by the key order of the transposed V reads) - P never touches LDS or other lanes.
#include "ug_common.h"
#include <stdlib.h>
#include <type_traits>

namespace {

constexpr int KVB = 64;      // keys per tile
constexpr bool UG_STAGGER_Q_IN_LDS = false;   // stagger variant: Q fragments from LDS (32 fewer VGPRs) or registers (a third less LDS read traffic in QK^T)

typedef __attribute__((address_space(3))) bf16x4* lds_b64_ptr;
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
// LDS-DMA as inline asm: hipcc's waitcnt pass then does not know DMAs are in flight (with the builtin it put `s_waitcnt vmcnt(0)` ahead of
// the first ds_read behind every barrier, i.e. one segment after the issue instead of two); the kernel states the one wait itself.
// M0 = LDS byte address of the wave's 1 KiB run (lane l lands at + 16 l); one wait state between the SALU write of M0 and the DMA.
// (M0 is a reserved register for hipcc - it never keeps a value there across statements and rejects it on a clobber list - so writing it here is safe.)
__device__ __forceinline__ unsigned lds_addr(const unsigned char* l) { return (unsigned)(size_t)(lptr_t)l; }
__device__ __forceinline__ const void* uniform_ptr(const void* p) {      // pin a wave-uniform pointer into an SGPR pair
    const unsigned long long a = (unsigned long long)p;
    unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    asm volatile("s_nop 4" : "+s"(lo), "+s"(hi));      // VALU-written SGPR -> VMEM base: 5 wait states, not padded inside an asm statement
    return (const void*)(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ void glds16_off(const void* base /* uniform_ptr() */, unsigned off_bytes, unsigned lds) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(off_bytes), "s"(base), "s"(lds) : "memory");
}
// buffer form of the same DMA: SGPR resource (base of the (batch, head)'s K or V) + per-lane byte offset + SGPR byte offset (the tile / run part)
__device__ __forceinline__ void bufds16(u32x4 rsrc, unsigned voff, unsigned soff, unsigned lds) {
    asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds" ::"v"(voff), "s"(rsrc), "s"(soff), "s"(lds) : "memory");
}
__device__ __forceinline__ void glds16_ptr(const void* g, unsigned lds) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g), "s"(lds) : "memory");
}
__device__ __forceinline__ void glds4_ptr(const void* g, unsigned lds) {          // 4 bytes per lane: lane l lands at lds + 4 l
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off" ::"v"(g), "s"(lds) : "memory");
}

// Row swizzle of the K/V tile images. f(row) is XORed into the 16-byte chunk index.
//   DH = 128 (256-byte rows): f = ((row & 3) << 2) | ((row >> 2) & 3)          (cdna guide T10, image (b))
//   DH =  64 (128-byte rows, two rows per 256-byte bank row): f = swap_bits_0_2((row >> 1) & 7): the 8 same-parity rows of a
//            ds_read_b128 lane group get 8 distinct chunks, and rows r, r+2 of a transposed-read block land in different
//            64-byte quarters -> both read kinds are conflict-free.
template <int DH>
__device__ __forceinline__ int row_swz(int row) {
    if constexpr (DH == 128) {
        return ((row & 3) << 2) | ((row >> 2) & 3);
    } else {
        const int v = (row >> 1) & 7;
        return ((v & 1) << 2) | (v & 2) | ((v >> 2) & 1);
    }
}
// byte offset of 16-byte chunk ch of row `row` in a [rows][DH x bf16] tile image
template <int DH>
__device__ __forceinline__ int img_off(int row, int ch) {
    return 2 * DH * row + 16 * (ch ^ row_swz<DH>(row));
}

__device__ __forceinline__ bf16x8 tr_read_pair(const unsigned char* lo, const unsigned char* hi) {
    const bf16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_b64_ptr)lo);
    const bf16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_b64_ptr)hi);
    return (bf16x8){a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
}

// max of three; without IEEE mode hipcc does not add a NaN-quieting self-max per operand (the scores are finite or -inf here)
__device__ __forceinline__ float ug_max3(float a, float b, float c) {
    return __builtin_fmaxf(__builtin_fmaxf(a, b), c);     // v_max3_f32 (attention.hip is built with -fno-honor-nans -mno-amdgpu-ieee)
}
// max over the two 32-lane halves (lane l and l ^ 32), in every lane: one v_permlane32_swap (VALU) instead of the ds_bpermute +
// lgkmcnt(0) that __shfl_xor compiles to (which also waits for every LDS read in flight)
__device__ __forceinline__ float ug_max_halves(float x) {
    const unsigned u = __float_as_uint(x);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return ug_max3(__uint_as_float(r[0]), __uint_as_float(r[1]), x);
}

// STAGGER (8 waves, the default): waves 0-3 and 4-7 - the two waves of every SIMD - run one segment apart. A wave alternates a
// matrix-only segment X(t) = P.V(t) then S^T(t+1) = K Q^T (32 MFMAs, an explicit fenced stream) with a VALU-only segment Y(t+1) = the
// online softmax of tile t+1, so a wave's softmax runs under its partner's MFMAs instead of both waves hitting the matrix pipe, then
// the VALU, together (the lock-step loop, STAGGER = false, kept for A/B: both phases then serialise and a tile costs the sum).
// Two barriers per tile; every thread fetches its share of K(t+2), V(t+1) at the start of an even segment and publishes it to LDS at
// the end of the following odd one, into buffers nobody reads in those two segments.
template <int DH, int NW, bool STAGGER, int PRIO = 1, bool WIDE = false, bool DMA = false, int KV = 64, int OCC = 2, int LSUM = 0, int AIS = 0, bool BUFD = false>   // head dim 128 | 64; waves per workgroup: 8 (256 query rows, 1 / CU) or 4 (128 rows, 2 / CU)
// BUFD (round 6; head width 64 / OCC 4 only): the whole-tile LDS-DMAs in BUFFER form. The stamps put group B's softmax segment 770 cycles above group A's
// (2263 vs 1497), all of it the 4 DMA issues per wave and tile; AIS = 1 showed the cost follows the issuer (A's X 1063 -> 1937), i.e. it is the issue
// sequence itself: per tile ~20 VALU instructions of lane-offset re-derivation (kept out of registers in round 3), a 64-bit VALU pointer bump,
// two v_readfirstlane + s_nop 4 per operand, on a SIMD whose VALU the four waves' softmax already saturates. Here the per-lane byte offset is ONE
// VGPR held through the loop (run 1's is that ^ 16: needs K and V to share a row stride that is a multiple of 16 elements - the dispatcher checks),
// the (batch, head) base is an SGPR resource, the tile / run offset an SGPR: per tile 4 x (s_mov m0 + buffer_load ... lds), one v_xor, scalar adds.
// AIS (round 6, DMA stagger only): WHO issues the LDS-DMAs of K(t+2) / V(t+1), and where. The per-segment stamps (tools/attn_stamps.py, profiles/r06b_*)
// show the loop's period is the SUM of the two groups' softmax segments - the matrix segment X is the shorter one of every pair (dh 64: Y 1497 /
// X 1063 cycles for group A, Y 2264 / X 1115 for group B) - and that group B's Y is 770 cycles longer than A's only because it opens with the
// tile's 4 DMA issues per wave (~140 cycles each + their address arithmetic), while group A then sits 1300 cycles at the barrier behind its X.
//   0: group B at the start of its softmax segment Y(t) (rounds 2-5);
//   1: group A at the END of its matrix segment X(t) - the same global segment 2t+2, so every buffer-reuse and landing deadline is unchanged
//      (K(t)'s and V(t-1)'s last readers finished in segment 2t+1; the data is waited for at the end of A's next softmax segment, 2t+3, and first
//      read in 2t+4) - i.e. inside the time A would spend waiting for B's softmax anyway;
//   2: split: group A issues K(t+2) at the end of X(t), group B V(t+1) at the start of Y(t) (each waits for its own).
// LSUM (round 6, stagger only): the softmax row sums leave the VALU. With the scores' scale / exp2 / max / pack the running sum `l += p` is one of ~5
// VALU instructions per score and the softmax segment Y is what the barriers wait for (tools/attn_stamps.py); here each lane's probabilities are
// summed on the matrix pipe instead, inside X, from the SAME packed bf16 fragments P.V consumes: v_mfma_f32_4x4x4_16b_bf16 (16 blocks of 4x4x4) with
// A = ones makes D[b][i][j] = sum_k B[b][k][j], i.e. every lane gets the sum of the four bf16 values IT passes as B (lane = block b, column j),
// accumulated over the tile's 8 half-fragments: 8 two-pass MFMAs per tile and wave (+12.5 % matrix-pipe cycles) for 32 v_add_f32 (-20 % of the
// softmax segment's issue cycles). The denominator is then the sum of the ROUNDED probabilities - the ones the numerator multiplies - not of their
// fp32 originals (relative difference <= 2^-9 / sqrt(keys), below the output's own bf16 rounding).
// KV: keys per tile. 64 everywhere in rounds 1-2; round 3 adds KV = 128 for head dim 64 (UniGenSD3): a 128-key tile of 128-byte rows is the
// same 16 KiB image, the same register budget (S^T 64 + P 32 + O 32 + Q 16 against 32 + 16 + 64 + 32 at dh 128 / 64 keys) and the same 32
// MFMAs per matrix segment as the dh 128 kernel, so the per-segment costs (two barriers, the max exchange, the lazy-rescale test, fences,
// the first-read latency) are paid once per 128 keys instead of once per 64 (at dh 64 the kernel ran at 55-60 %
// of its VALU-issue bound).
// PRIO (stagger only): 0 = no priority games; 1 = s_setprio 1 around the matrix stream of every X segment; 2 = ONE static s_setprio 1 for
// the younger wave group (waves 4-7) before the loop (cdna guide T5, static form). WIDE: 16-byte epilogue stores (T21).
// DMA (stagger only): K / V tiles go HBM -> LDS with global_load_lds_dwordx4 (no staging registers, no ds_write): the swizzled image is
// produced on the SOURCE side (lane l of an instruction lands at byte 16 l of a 1 KiB run = 4 rows at dh 128, so it fetches chunk
// (l % 16) ^ f(row) of its row), and group B (waves 4-7) issues all of it at the start of its softmax segment, two segments ahead of use.
// OCC: waves per SIMD the register allocation must allow. 2 = one 8-wave workgroup per CU (all shipped forms). OCC = 4 (round 3, head dim 64
// only, A/B): <= 128 registers so that TWO workgroups share a CU (64 KiB of LDS each) - four waves per SIMD fill each other's barrier and
// latency bubbles in the VALU-bound dh 64 loop; Q fragments then come from LDS (16 registers fewer).
__global__ __launch_bounds__(64 * NW, OCC) void flash_attn_kernel(
    const bf16_t* __restrict__ q, int64_t q_rs, int64_t q_bs, const bf16_t* __restrict__ k, int64_t k_rs, int64_t k_bs,
    const bf16_t* __restrict__ v, int64_t v_rs, int64_t v_bs, bf16_t* __restrict__ o, int64_t o_rs, int64_t o_bs,
    int heads, int Lq, int Lkv, int nQ, float c /* softmax_scale * log2(e) */, float* __restrict__ lse_out /* nullable */, int64_t lse_ld) {
    constexpr int KVB = KV;                          // shadows the file-level constant (the other kernels keep 64)
    constexpr int NKB = KVB / 32;                    // 32-key blocks of S^T per tile
    constexpr int NKS = KVB / 16;                    // k-steps of O^T += V^T P^T per tile
    constexpr int RB = 2 * DH;                       // row bytes
    constexpr int NCH = DH / 8;                      // 16-byte chunks per row
    constexpr int TILE = KVB * RB;                   // bytes of one K (or V) tile image
    constexpr int QS = DH / 16;                      // k-steps of S^T = K Q^T
    constexpr int NDB = DH / 32;                     // 32-wide d blocks of O^T
    constexpr int QROWS = 32 * NW, NT = 64 * NW, NST = (KVB * NCH) / NT > 0 ? (KVB * NCH) / NT : 1;   // staging chunks of K (and of V) per thread and tile
    static_assert((KVB * NCH) / NT >= 1 || DMA, "tile smaller than the workgroup: register staging cannot cover it (the LDS-DMA form can)");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // [2][K tile | V tile]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    /* -DUG_ATTN_STAMPS (tools/attn_stamps.py, a separate library build; never the product): wave 0 / wave 4 of every workgroup record s_memtime at
     * the kernel's entry, the end of the prologue, the end of the tile loop and the end of the epilogue, s_memrealtime at both ends and the CU they
     * ran on, into the buffer the caller passes as `lse_out` (24 dwords per workgroup and wave group, incl. the four per-segment accumulators of the stagger loop; the log-sum-exp is then not written). */
#ifdef UG_ATTN_STAMPS
    unsigned long long ug_st[4], ug_rt0 = __builtin_amdgcn_s_memrealtime();
#define UG_ASTAMP(I) do { ug_st[I] = __builtin_amdgcn_s_memtime(); } while (0)
    UG_ASTAMP(0);
    // per-segment accumulators of the stagger loop (round 6): cycles a wave spends in Y (softmax, incl. group B's DMA issue), at the barrier behind
    // it, in X (P.V + K.Q^T, incl. group B's DMA wait) and at the barrier behind that, summed over the tiles
    unsigned long long ug_seg[4] = {0, 0, 0, 0}, ug_t = 0;
#define UG_SEG0() do { ug_t = __builtin_amdgcn_s_memtime(); } while (0)
#define UG_SEG(I) do { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); ug_seg[I] += n_ - ug_t; ug_t = n_; } while (0)
#else
#define UG_ASTAMP(I) do { } while (0)
#define UG_SEG0() do { } while (0)
#define UG_SEG(I) do { } while (0)
#endif

    // XCD-aware block order: all query tiles of one (batch, head) run on one XCD so its K/V stay in that L2.
    const int nwg = gridDim.x;
    const int qd = nwg >> 3, rm = nwg & 7;
    const int xcd = blockIdx.x & 7, kk = blockIdx.x >> 3;
    const int logical = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + kk;
    const int qt = logical % nQ;
    const int bh = logical / nQ;
    const int head = bh % heads, b = bh / heads;

    const bf16_t* Qb = q + (int64_t)b * q_bs + head * DH;
    const bf16_t* Kb = k + (int64_t)b * k_bs + head * DH;
    const bf16_t* Vb = v + (int64_t)b * v_bs + head * DH;

    // ---- Q fragments (B operand of S^T = K Q^T): lane (r, h) holds Q[q = r][d = 16 s + 8 h + j] ----
    const int q_row = qt * QROWS + wave * 32 + r;
    const int q_ld = q_row < Lq ? q_row : Lq - 1;
    // Lock-step variant: Q fragments stay in registers. X/Y stagger: they live in LDS (same swizzled row image as K, one
    // ds_read_b128 per k-step) because S^T must survive a barrier next to the P.V operands and 32 fewer VGPRs avoid spills.
    constexpr int QBASE = 2 * 2 * KVB * RB;            // byte offset of the Q image behind the two K|V buffers
    constexpr bool QLDS = STAGGER && (UG_STAGGER_Q_IN_LDS || OCC == 4);
    bf16x8 qf[QLDS ? 1 : QS];
    const int q_lds = QBASE + RB * (wave * 32 + r);
    const int qx = h ^ row_swz<DH>(r);                  // wave * 32 keeps row_swz unchanged (multiple of 16)
    if constexpr (!QLDS) {
#pragma unroll
        for (int s = 0; s < QS; ++s) qf[s] = *(const bf16x8*)(Qb + (int64_t)q_ld * q_rs + 16 * s + 8 * h);
        // Retire the Q loads HERE: the empty asm takes every fragment as a read-write operand, so hipcc must have the loaded
        // values in hand before it (it waits vmcnt there) and treats them as fresh afterwards. Without it the loads are sunk to
        // the loop header and every iteration re-waits for them with vmcnt(7..0), draining the K/V prefetch issued at its top.
        if constexpr (QS == 8)
            asm volatile("" : "+v"(qf[0]), "+v"(qf[1]), "+v"(qf[2]), "+v"(qf[3]), "+v"(qf[4]), "+v"(qf[5]), "+v"(qf[6]), "+v"(qf[7]));
        else
            asm volatile("" : "+v"(qf[0]), "+v"(qf[1]), "+v"(qf[2]), "+v"(qf[3]));
    } else {
        // each lane copies the 16-byte chunks (16 s + 8 h) of its own query row; only this wave reads them back
#pragma unroll
        for (int s = 0; s < QS; ++s) {
            const u32x4 v4 = *(const u32x4*)(Qb + (int64_t)q_ld * q_rs + 16 * s + 8 * h);
            *(u32x4*)(smem + q_lds + 16 * ((2 * s) ^ qx)) = v4;
        }
    }

    // ---- staging assignment: thread -> 2 chunks of K and 2 of V per tile ----
    int st_row[NST], st_ch[NST], st_off[NST];
#pragma unroll
    for (int u = 0; u < NST; ++u) {
        const int cid = tid + NT * u;
        st_row[u] = cid / NCH; st_ch[u] = cid % NCH;
        st_off[u] = img_off<DH>(st_row[u], st_ch[u]);
    }
    u32x4 kreg[NST], vreg[NST];
    auto stage_load = [&](int kv0) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < NST; ++u) {
            int key = kv0 + st_row[u]; if (key > Lkv - 1) key = Lkv - 1;
            kreg[u] = *(const u32x4*)(Kb + (int64_t)key * k_rs + st_ch[u] * 8);
            vreg[u] = *(const u32x4*)(Vb + (int64_t)key * v_rs + st_ch[u] * 8);
        }
    };
    auto stage_write = [&](int buf) __attribute__((always_inline)) {
        unsigned char* Kbuf = smem + buf * 2 * TILE;
        unsigned char* Vbuf = Kbuf + TILE;
#pragma unroll
        for (int u = 0; u < NST; ++u) {
            *(u32x4*)(Kbuf + st_off[u]) = kreg[u];
            *(u32x4*)(Vbuf + st_off[u]) = vreg[u];
        }
    };

    // ---- per-lane LDS read offsets ----
    // K row read: row = kb*32 + r, chunk = 2s + h  ->  RB*row + 16*((2s) ^ kx),  kx = h ^ f(r)   (f ignores the kb*32 part)
    const int k_rowoff = RB * r;
    const int kx = h ^ row_swz<DH>(r);
    // V transposed read: group g = lane>>4 (16 lanes), i = lane&15. Block rows = keys 16ks + 4h + (i>>2) (+8 for the
    // second half of the k-step), columns d = 32db + 16(g&1) + 4(i&3)..+3. Lane receives column d = 32db + (lane&31).
    const int i16 = lane & 15, g16 = lane >> 4;
    const int v_key = 4 * h + (i16 >> 2);
    const int v_lowch = 2 * (g16 & 1) + ((i16 & 3) >> 1);
    const int v_b8 = 8 * (i16 & 1);
    int voff_lo[NDB], voff_hi[NDB];
#pragma unroll
    for (int db = 0; db < NDB; ++db) {
        const int ch = 4 * db + v_lowch;
        voff_lo[db] = RB * v_key + 16 * (ch ^ row_swz<DH>(v_key)) + v_b8;              // f(16 ks + key) == f(key)
        voff_hi[db] = RB * (v_key + 8) + 16 * (ch ^ row_swz<DH>(v_key + 8)) + v_b8;
    }

    f32x16 oacc[NDB];
#pragma unroll
    for (int db = 0; db < NDB; ++db)
#pragma unroll
        for (int i = 0; i < 16; ++i) oacc[db][i] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;
    static_assert(!LSUM || STAGGER, "the matrix-pipe row sum lives in the stagger loop's X segment");
    f32x4 lacc = {0.f, 0.f, 0.f, 0.f}, lacc2 = {0.f, 0.f, 0.f, 0.f};   // LSUM: register 0 of each = half of this lane's running row sum (rows 1-3 of its 4x4 block: unused
                                                                       // copies); LSUM = 2: two chains, so that no MFMA of a k-step waits for the one issued just before it (4 more registers)
    bf16x4 ones4 = {(short)0x3f80, (short)0x3f80, (short)0x3f80, (short)0x3f80};
    if constexpr (LSUM) asm volatile("" : "+v"(ones4));        // one VGPR pair for the loop, not re-materialised per use

    const int ntiles = (Lkv + KVB - 1) / KVB;
    bf16x8 pf[NKB][2];                                 // P^T fragments of the tile between its S and P stages
    // CUR = buffer parity as a compile-time constant: every LDS address below is then a loop-invariant VGPR + an immediate offset
    // (with a runtime parity hipcc re-materialised ~50 address adds per tile, a quarter of the VALU work of the loop).
    f32x16 sacc[NKB];                                  // S^T of the tile between its QK^T and its softmax
    auto do_QK = [&](int t, auto cur_c) __attribute__((always_inline)) {
        constexpr int CUR = decltype(cur_c)::value;
        const int kv0 = t * KVB;
        const unsigned char* Kbuf = smem + CUR * 2 * TILE;
        // ---- S^T[key][q]: all 8 K fragments of key block 0 first, then block-0 MFMAs with the block-1 reads between them ----
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
            for (int i = 0; i < 16; ++i) sacc[kb][i] = 0.f;
        if constexpr (!STAGGER) {
            bf16x8 kf[NKB][QS];
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
                for (int s = 0; s < QS; ++s) kf[kb][s] = *(const bf16x8*)(Kbuf + kb * 32 * RB + k_rowoff + 16 * ((2 * s) ^ kx));
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
                for (int s = 0; s < QS; ++s) sacc[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[kb][s], qf[s], sacc[kb], 0, 0, 0);
            if constexpr (NKB == 2) {
                __builtin_amdgcn_sched_group_barrier(0x100, QS, 0);       // ds_reads of key block 0
#pragma unroll
                for (int s = 0; s < QS; ++s) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);    // 1 MFMA (block 0)
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);    // 1 ds_read (block 1)
                }
                __builtin_amdgcn_sched_group_barrier(0x008, QS, 0);       // MFMAs of block 1
            }
        } else {
            // Q comes from LDS too: per k-step one Q fragment and the two key blocks' K fragments, read two steps ahead of their
            // MFMAs (9 fragments = 36 VGPRs live instead of 24 fragments if hipcc hoisted every read).
            bf16x8 ql[QS], kf[NKB][QS];
#pragma unroll
            for (int s = 0; s < QS; ++s) {
                if constexpr (QLDS) ql[s] = *(const bf16x8*)(smem + q_lds + 16 * ((2 * s) ^ qx)); else ql[s] = qf[s];
#pragma unroll
                for (int kb = 0; kb < NKB; ++kb) kf[kb][s] = *(const bf16x8*)(Kbuf + kb * 32 * RB + k_rowoff + 16 * ((2 * s) ^ kx));
            }
#pragma unroll
            for (int s = 0; s < QS; ++s)
#pragma unroll
                for (int kb = 0; kb < NKB; ++kb) sacc[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[kb][s], ql[s], sacc[kb], 0, 0, 0);
            if constexpr (NKB == 2) {
                __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);        // fragments of k-steps 0, 1
#pragma unroll
                for (int s = 0; s < QS - 2; ++s) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);    // MFMAs of step s
                    __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);    // fragments of step s + 2
                }
                __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
            }
        }
        if (kv0 + KVB > Lkv) {   // ragged last tile: keys >= Lkv do not exist
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int key = kv0 + kb * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
                    if (key >= Lkv) sacc[kb][i] = -INFINITY;
                }
        }
    };
    auto do_SM = [&]() __attribute__((always_inline)) {
        // ---- online softmax, all lane-local (this lane: query r, 32 of the tile's 64 keys; lane^32 has the rest) ----
        float tmax = sacc[0][0];
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
            for (int i = 0; i < 16; ++i) tmax = fmaxf(tmax, sacc[kb][i]);
        tmax = ug_max_halves(tmax);
        // Lazy reference point: a row's running max moves only when the tile maximum exceeds it by more than 2^8 in the exponent
        // (softmax is shift-invariant; P and l stay below 2^8 per element: exact in fp32, same relative precision in bf16). With the
        // exact max some row of the wave moves in most tiles and the 64-register rescale below ran nearly every iteration.
        const bool up = (tmax - m_run) * c > 8.0f;
        const float m_new = up ? tmax : m_run;
        if (!__all(!up)) {
            const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c);
            if constexpr (LSUM) { lacc[0] *= alpha; lacc2[0] *= alpha; } else l_run *= alpha;
#pragma unroll
            for (int db = 0; db < NDB; ++db)
#pragma unroll
                for (int i = 0; i < 16; ++i) oacc[db][i] *= alpha;
            m_run = m_new;
        }
        const float mc = m_run * c;
        // (Measured and dropped, round 2, bit-identical: exponentiating key block 1 - or only its last 8 scores per lane - inside X(t), 3-7 VALU
        // instructions behind each of its first 8 P.V MFMAs: -8 % / -4 % at dh 128, -10 % / -7 % at dh 64. Ablations of the same day (tools/attn_ab.py
        // on diagnostic builds): without the K/V DMAs +7-10 %, without this softmax +18 % (+37 % at dh 64), without both +28 % (+53 %): the matrix
        // segments alone take 78 % of the loop's time at dh 128, and VALU work moved into them costs more than it saves here.)
        // (Measured and dropped: the row sum from the packed bf16 probabilities, two per v_dot2c_f32_bf16: -3.5 % at dh 128. Considered and
        // rejected on accuracy: Q pre-multiplied by scale * log2(e) in bf16 with the accumulators initialised to -m (no fma per score): the
        // attention error against fp32 grows from 1.6e-3 to 2.3e-3, 4x on peaked rows.)
        // (Measured and dropped, same box: the scale / shift and the row sums two elements per instruction, v_pk_fma_f32 / v_pk_add_f32 -
        // 5 % SLOWER at dh 128 (1086 vs 1146, 1118 vs 1179 TFLOP/s), +1 % at dh 64: the packed forms buy no issue cycles here.)
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
            float p[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                p[i] = __builtin_amdgcn_exp2f(fmaf(sacc[kb][i], c, -mc));
                if constexpr (!LSUM) l_run += p[i];
            }
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                u32x4 w;
                w.x = pack2bf(p[8 * s2 + 0], p[8 * s2 + 1]); w.y = pack2bf(p[8 * s2 + 2], p[8 * s2 + 3]);
                w.z = pack2bf(p[8 * s2 + 4], p[8 * s2 + 5]); w.w = pack2bf(p[8 * s2 + 6], p[8 * s2 + 7]);
                pf[kb][s2] = __builtin_bit_cast(bf16x8, w);
            }
        }
    };
    auto do_P = [&](int t, auto cur_c) __attribute__((always_inline)) {
        constexpr int CUR = decltype(cur_c)::value;
        const unsigned char* Vbuf = smem + CUR * 2 * TILE + TILE;
        // ---- O^T[d][q] += V^T[d][key] P^T[key][q]: the V fragments of d-block db+1 are read between the MFMAs of block db ----
        {
            bf16x8 vf[NDB][NKS];
#pragma unroll
            for (int db = 0; db < NDB; ++db)
#pragma unroll
                for (int ks = 0; ks < NKS; ++ks)
                    vf[db][ks] = tr_read_pair(Vbuf + ks * 16 * RB + voff_lo[db], Vbuf + ks * 16 * RB + voff_hi[db]);
#pragma unroll
            for (int db = 0; db < NDB; ++db)
#pragma unroll
                for (int ks = 0; ks < NKS; ++ks)
                    oacc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[db][ks], pf[ks >> 1][ks & 1], oacc[db], 0, 0, 0);
            if constexpr (NKS == 4) {
                __builtin_amdgcn_sched_group_barrier(0x100, 8, 1);            // 8 tr reads (d-block 0)
#pragma unroll
                for (int i = 0; i < 4 * (NDB - 1); ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 1);        // 1 MFMA
                    __builtin_amdgcn_sched_group_barrier(0x100, 2, 1);        // 2 tr reads of the next d-block
                }
                __builtin_amdgcn_sched_group_barrier(0x008, 4, 1);            // last d-block
            }
        }
    };
    // X(t) of the stagger variant as an explicit stream (sched_barrier after every piece): P.V(t) - 16 MFMAs, k-step outer so the 4
    // (8 at dh = 128... NDB) accumulators rotate - then S^T(t+1) = K Q^T - 16 MFMAs. Every LDS fragment is read two or three steps ahead
    // of its MFMA and the first K / Q fragments of the second half are requested under the last P.V MFMAs: this wave is alone on the
    // matrix pipe in this segment (its SIMD partner is in the VALU-only Y), so an exposed ds_read latency is an idle pipe. hipcc's
    // own order (sched_group_barrier hints included) ran the segment at 70-90 cycles per MFMA.
    auto qx_frag = [&](int s) __attribute__((always_inline)) -> bf16x8 { if constexpr (QLDS) return *(const bf16x8*)(smem + q_lds + 16 * ((2 * s) ^ qx)); else return qf[s]; };
    auto do_X = [&](int t, auto cur_c, bool have_qk, auto&& hook, auto kofs_c) __attribute__((always_inline)) {
        constexpr int CUR = decltype(cur_c)::value;
        constexpr int KOFS = decltype(kofs_c)::value;       // >= 0 (AIS 5): byte offset of the ring slot that holds K(t+1), a compile-time constant like CUR
        const unsigned char* Vbuf = smem + CUR * 2 * TILE + TILE;
        const unsigned char* Kbuf = KOFS >= 0 ? smem + KOFS : smem + (CUR ^ 1) * 2 * TILE;
        bf16x8 vf[NKS][NDB];
        auto rdv = [&](int ks) __attribute__((always_inline)) {
#pragma unroll
            for (int db = 0; db < NDB; ++db) vf[ks][db] = tr_read_pair(Vbuf + ks * 16 * RB + voff_lo[db], Vbuf + ks * 16 * RB + voff_hi[db]);
        };
        bf16x8 kf[NKB][QS];
        auto rdk = [&](int s) __attribute__((always_inline)) {
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb) kf[kb][s] = *(const bf16x8*)(Kbuf + kb * 32 * RB + k_rowoff + 16 * ((2 * s) ^ kx));
        };
        constexpr int QPRE = QS / 4;                   // k-steps of K.Q^T whose fragments are requested under each of the last two P.V steps
        rdv(0); rdv(1);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (PRIO == 1) __builtin_amdgcn_s_setprio(1);                   // the matrix stream outranks the partner wave's softmax VALU at issue
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
#pragma unroll
            for (int db = 0; db < NDB; ++db)
                oacc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[ks][db], pf[ks >> 1][ks & 1], oacc[db], 0, 0, 0);
            if constexpr (LSUM) {        // this lane's 8 probabilities of the k-step, summed on the matrix pipe (see the template's header)
                const bf16x8 pw = pf[ks >> 1][ks & 1];
                lacc = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(ones4, __builtin_shufflevector(pw, pw, 0, 1, 2, 3), lacc, 0, 0, 0);
                if constexpr (LSUM == 2) lacc2 = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(ones4, __builtin_shufflevector(pw, pw, 4, 5, 6, 7), lacc2, 0, 0, 0);
                else lacc = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(ones4, __builtin_shufflevector(pw, pw, 4, 5, 6, 7), lacc, 0, 0, 0);
            }
            hook(ks);                                                              // AIS 3: one LDS-DMA piece behind this k-step's MFMAs (no-op otherwise)
            __builtin_amdgcn_sched_barrier(0);
            if (ks + 2 < NKS) rdv(ks + 2);
            else if (have_qk) {                                                    // the first 2 QPRE k-steps of the second half
#pragma unroll
                for (int j = 0; j < QPRE; ++j) rdk(QPRE * (ks - (NKS - 2)) + j);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (!have_qk) { if constexpr (PRIO == 1) __builtin_amdgcn_s_setprio(0); return; }
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
            for (int i = 0; i < 16; ++i) sacc[kb][i] = 0.f;
#pragma unroll
        for (int s = 0; s < QS; ++s) {
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb) sacc[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[kb][s], qx_frag(s), sacc[kb], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (s + 2 * QPRE < QS) { rdk(s + 2 * QPRE); __builtin_amdgcn_sched_barrier(0); }
        }
        if constexpr (PRIO == 1) __builtin_amdgcn_s_setprio(0);
        const int kv0 = (t + 1) * KVB;
        if (kv0 + KVB > Lkv) {   // ragged last tile: keys >= Lkv do not exist
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int key = kv0 + kb * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
                    if (key >= Lkv) sacc[kb][i] = -INFINITY;
                }
        }
    };
    if constexpr (!STAGGER) {
        stage_load(0);
        stage_write(0);
        __syncthreads();
        auto tile = [&](int t, auto cur_c) __attribute__((always_inline)) {
            constexpr int CUR = decltype(cur_c)::value;
            if (t + 1 < ntiles) stage_load((t + 1) * KVB);
            do_QK(t, cur_c);
            do_SM();
            do_P(t, cur_c);
            if (t + 1 < ntiles) stage_write(CUR ^ 1);
            __syncthreads();
        };
        for (int t = 0; t < ntiles; t += 2) {
            tile(t, std::integral_constant<int, 0>{});
            if (t + 1 < ntiles) tile(t + 1, std::integral_constant<int, 1>{});
        }
    } else {
        static_assert(!STAGGER || NW == 8 || (NW == 16 && DMA), "the stagger pairs the waves of one SIMD: w, w + 4 (and w + 8, w + 12 in the 16-wave form)");
        // X / Y stagger. A wave alternates a MATRIX-only segment X(t) = P.V of tile t followed by S^T = K.Q^T of tile t+1, and a
        // VALU-only segment Y(t+1) = online softmax of tile t+1. Waves 0-3 (group A) and 4-7 (group B) - the two waves of every
        // SIMD - run one segment apart, so in every segment a SIMD has one wave feeding the matrix pipe and one feeding the VALU
        // (PMC on the lock-step loop: matrix pipe busy 42 %, VALU 45 %, hardly overlapping).
        //   global segment:   0       1       2       3       4
        //   group A:        QK(0)    Y(0)    X(0)    Y(1)    X(1) ...
        //   group B:          -     QK(0)    Y(0)    X(0)    Y(1) ...
        // K(t+1) and V(t) are first needed in segment 2t+2: every thread fetches its share at the START of even segment 2t and
        // publishes it at the END of odd segment 2t+1 (into buffers nobody reads in 2t / 2t+1). Loads cross barriers: raw s_barrier.
        auto seg_barrier = [&]() __attribute__((always_inline)) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        };
        // K(kt), V(vt) -> registers -> LDS. Branch-free on purpose: tiles past the end are clamped re-reads published into buffers whose
        // last readers are done. With a per-load `if (tile < ntiles)` every global_load sat in its own basic block behind an
        // `s_waitcnt vmcnt(0)`: the four loads of a fetch ran one after the other at full memory latency (~4400 cycles per fetch
        // segment, found with s_memtime stamps) - that, not the segment structure, is why this variant first measured 602 TFLOP/s.
        // Addresses: a wave-uniform tile base (SALU) + a per-thread 32-bit element offset computed once - a fetch is then 2 NST loads and
        // no VALU (with `key * stride` in 64 bits per load, a fetch cost the matrix segment ~700 cycles before its first MFMA). The
        // ragged last tile and tiles past the end take the clamped path.
        unsigned koff[NST], voff[NST];
#pragma unroll
        for (int u = 0; u < NST; ++u) {
            koff[u] = (unsigned)(st_row[u] * (int)k_rs + st_ch[u] * 8);
            voff[u] = (unsigned)(st_row[u] * (int)v_rs + st_ch[u] * 8);
        }
        auto fetch = [&](int kt, int vt) __attribute__((always_inline)) {
            if (kt * KVB + KVB <= Lkv && vt * KVB + KVB <= Lkv) {       // wave-uniform: both tiles whole
                const bf16_t* kbase = Kb + (int64_t)kt * KVB * k_rs;
                const bf16_t* vbase = Vb + (int64_t)vt * KVB * v_rs;
#pragma unroll
                for (int u = 0; u < NST; ++u) {
                    kreg[u] = *(const u32x4*)(kbase + koff[u]);
                    vreg[u] = *(const u32x4*)(vbase + voff[u]);
                }
            } else {
#pragma unroll
                for (int u = 0; u < NST; ++u) {
                    int key = kt * KVB + st_row[u]; if (key > Lkv - 1) key = Lkv - 1;
                    kreg[u] = *(const u32x4*)(Kb + (int64_t)key * k_rs + st_ch[u] * 8);
                    key = vt * KVB + st_row[u]; if (key > Lkv - 1) key = Lkv - 1;
                    vreg[u] = *(const u32x4*)(Vb + (int64_t)key * v_rs + st_ch[u] * 8);
                }
            }
        };
        auto publish = [&](int kt, int vt) __attribute__((always_inline)) {
#pragma unroll
            for (int u = 0; u < NST; ++u) {
                *(u32x4*)(smem + (kt & 1) * 2 * TILE + st_off[u]) = kreg[u];
                *(u32x4*)(smem + (vt & 1) * 2 * TILE + TILE + st_off[u]) = vreg[u];
            }
        };
        const bool groupA = __builtin_amdgcn_readfirstlane(wave) < NW / 2;
        // LDS-DMA staging: a tile image is NI runs of 1 KiB (RPI rows each); wave wb of group B owns runs wb * NIW .. + NIW - 1
        // (measured and dropped: every wave issuing NI / 8 runs, group A's half at the start of its own softmax segment 2t+1 and waited for
        // at its end - same bits, -0.5 % at dh 128, -12 % at dh 64: group B's issue cost is not what bounds the segment pairs; and group A
        // issuing all of them one at a time behind the MFMAs of the first 2 NIW steps of its matrix segment: -11 % / -4 %, ~46 cycles of
        // matrix-segment time per DMA)
        constexpr int RPI = 1024 / RB, NI = TILE / 1024, NIW = NI / (NW / 2);
        static_assert(NIW >= 1, "fewer 1 KiB runs in a tile than issuing waves");
        const int wb = __builtin_amdgcn_readfirstlane(wave) & (NW / 2 - 1);
        unsigned dko[NIW], dvo[NIW];
#pragma unroll
        for (int u = 0; u < NIW; ++u) {
            const int row = (wb * NIW + u) * RPI + lane / NCH;
            const int ch = (lane % NCH) ^ row_swz<DH>(row);
            dko[u] = (unsigned)(row * (int)k_rs + ch * 8) * 2u;        // bytes
            dvo[u] = (unsigned)(row * (int)v_rs + ch * 8) * 2u;
        }
        constexpr bool BUFD64 = BUFD && OCC == 4 && DH == 64;      // the one-live-VGPR form; any other BUFD kernel keeps its per-run lane offsets (dko / dvo) and only moves
                                                                 // the tile part of the address from a 64-bit VALU pointer + v_readfirstlane pair into the scalar offset
        static_assert(!BUFD || DMA, "BUFD is a form of the LDS-DMA staging");
        static_assert(!BUFD64 || NIW == 2 || NIW == 1, "the head-width-64 form derives run 1 from run 0");
        unsigned bvo = 0;                              // BUFD: lane offset of run 0 inside a wave's pair of 1 KiB runs (rows lane / 8, swizzled chunk)
        u32x4 rsK = {0u, 0u, 0u, 0u}, rsV = {0u, 0u, 0u, 0u};
        if constexpr (BUFD) {
            if constexpr (BUFD64) {
                const int sw = (((lane >> 4) & 1) << 2) | ((lane >> 4) & 2);
                bvo = (unsigned)((lane >> 3) * (int)k_rs) * 2u + (unsigned)(((lane & 7) ^ sw) << 4);
                if constexpr (NIW == 1) bvo ^= (unsigned)(wb & 1) << 4;           // one run per wave: the run's parity (row bit 3) is the swizzle's chunk bit 0
                asm volatile("" : "+v"(bvo));
            }
            const unsigned long long ka = (unsigned long long)uniform_ptr(Kb), va = (unsigned long long)uniform_ptr(Vb);
            rsK = (u32x4){(unsigned)ka, (unsigned)(ka >> 32), 0xffffffffu, 0x00020000u};
            rsV = (u32x4){(unsigned)va, (unsigned)(va >> 32), 0xffffffffu, 0x00020000u};
        }
        auto dma_tile = [&](const bf16_t* base, int64_t rs, const unsigned (&off)[NIW], int tile, unsigned dst, auto is_k) {
            if (tile * KVB + KVB <= Lkv) {             // whole tile: wave-uniform base (SGPR pair) + per-lane 32-bit byte offset
                if constexpr (BUFD && !BUFD64) {
                    const unsigned so = (unsigned)(tile * KVB * (int)rs) * 2u;                             // scalar: the tile part; the run rows are in the lane offsets
#pragma unroll
                    for (int u = 0; u < NIW; ++u) {
                        if constexpr (decltype(is_k)::value) bufds16(rsK, off[u], so, dst + u * 1024); else bufds16(rsV, off[u], so, dst + u * 1024);
                    }
                } else if constexpr (BUFD) {
                    const unsigned so = (unsigned)((tile * KVB + wb * NIW * RPI) * (int)rs) * 2u;          // scalar: tile and run-pair part of the byte offset
                    if constexpr (decltype(is_k)::value) {
                        bufds16(rsK, bvo, so, dst);
                        if constexpr (NIW == 2) bufds16(rsK, bvo ^ 16u, so + (unsigned)(RPI * (int)rs) * 2u, dst + 1024);
                    } else {
                        bufds16(rsV, bvo, so, dst);
                        if constexpr (NIW == 2) bufds16(rsV, bvo ^ 16u, so + (unsigned)(RPI * (int)rs) * 2u, dst + 1024);
                    }
                } else if constexpr (OCC == 4 && DH == 64 && NIW == 2) {
                    // Lane offsets re-derived at the issue (not kept live through the loop: registers are what this form is short of), cheaply:
                    // run u of wave wb covers rows (2 wb + u) * 8 + lane / 8, so the wave / run part of the row goes into the scalar base and
                    // row_swz<64> reduces to a lane term with bit 0 = u: the second run's chunk is the first one's ^ 1. ~10 VALU per tile
                    // instead of ~48 (round 3: the generic re-derivation was ~12 % of the issuing waves' VALU instructions).
                    int lane_r = lane;
                    asm volatile("" : "+v"(lane_r));
                    const int sw = (((lane_r >> 4) & 1) << 2) | ((lane_r >> 4) & 2);
                    const unsigned c0 = (unsigned)(((lane_r & 7) ^ sw) << 4), rp = (unsigned)((lane_r >> 3) * (int)rs) * 2u;
                    const char* tw = (const char*)uniform_ptr(base + ((int64_t)tile * KVB + wb * NIW * RPI) * rs);
                    glds16_off(tw, rp + c0, dst);
                    glds16_off(tw + (int64_t)RPI * rs * 2, rp + (c0 ^ 16u), dst + 1024);
                } else if constexpr (OCC == 4) {
                    const void* tb = uniform_ptr(base + (int64_t)tile * KVB * rs);
                    int lane_r = lane;
                    asm volatile("" : "+v"(lane_r));
#pragma unroll
                    for (int u = 0; u < NIW; ++u) {
                        const int row = (wb * NIW + u) * RPI + lane_r / NCH;
                        const int ch = (lane_r % NCH) ^ row_swz<DH>(row);
                        glds16_off(tb, (unsigned)(row * (int)rs + ch * 8) * 2u, dst + u * 1024);
                    }
                } else {
                    const void* tb = uniform_ptr(base + (int64_t)tile * KVB * rs);
#pragma unroll
                    for (int u = 0; u < NIW; ++u) glds16_off(tb, off[u], dst + u * 1024);
                }
            } else {                                   // ragged last tile: rows past the end re-read the last key (masked in S^T)
                int lane_r = lane;
                asm volatile("" : "+v"(lane_r));      // row / chunk re-derived here, not kept live through the loop
#pragma unroll
                for (int u = 0; u < NIW; ++u) {
                    const int row = (wb * NIW + u) * RPI + lane_r / NCH;
                    const int ch = (lane_r % NCH) ^ row_swz<DH>(row);
                    int key = tile * KVB + row; if (key > Lkv - 1) key = Lkv - 1;
                    glds16_ptr(base + (int64_t)key * rs + ch * 8, dst + u * 1024);
                }
            }
        };
        // AIS 5: K tiles live in a ring of THREE slots - the two K halves of the double buffer and the (unused: Q stays in registers) Q image area behind it
        auto kslot = [&](int s3) __attribute__((always_inline)) -> int { return s3 < 2 ? s3 * 2 * TILE : 4 * TILE; };
        static_assert(AIS != 5 || (DMA && !QLDS && !BUFD64), "AIS 5 keeps its third K slot where the Q image would be");
        auto dma_fetch = [&](int kt, int vt) __attribute__((always_inline)) {         // tiles past the end are simply not fetched
            const unsigned l0 = __builtin_amdgcn_readfirstlane(lds_addr(smem)) + wb * NIW * 1024;
            if (kt < ntiles) dma_tile(Kb, k_rs, dko, kt, l0 + (AIS == 5 ? kslot(kt % 3) : (kt & 1) * 2 * TILE), std::true_type{});
            if (vt < ntiles) dma_tile(Vb, v_rs, dvo, vt, l0 + (vt & 1) * 2 * TILE + TILE, std::false_type{});
        };
        static_assert(AIS == 0 || DMA, "AIS re-assigns the LDS-DMA issue");
        static_assert((AIS != 3 && AIS != 4) || (BUFD64 && NKS == 4 && NIW == 2), "AIS 3 / 4 spread the buffer-form pieces over the P.V k-steps");
        auto dma_piece = [&](int kt, int vt, int j) __attribute__((always_inline)) {
            if constexpr (BUFD64) {
                const unsigned l0 = __builtin_amdgcn_readfirstlane(lds_addr(smem)) + wb * NIW * 1024;
                const int tile = j < 2 ? kt : vt;
                if (tile >= ntiles) return;
                const unsigned dst = l0 + (tile & 1) * 2 * TILE + (j < 2 ? 0 : TILE);
                if (tile * KVB + KVB <= Lkv) {
                    const unsigned so = (unsigned)((tile * KVB + wb * NIW * RPI) * (int)k_rs) * 2u;
                    if (j == 0) bufds16(rsK, bvo, so, dst);
                    else if (j == 1) bufds16(rsK, bvo ^ 16u, so + (unsigned)(RPI * (int)k_rs) * 2u, dst + 1024);
                    else if (j == 2) bufds16(rsV, bvo, so, dst);
                    else bufds16(rsV, bvo ^ 16u, so + (unsigned)(RPI * (int)k_rs) * 2u, dst + 1024);
                } else if (j == 0) dma_tile(Kb, k_rs, dko, tile, dst, std::true_type{});
                else if (j == 2) dma_tile(Vb, v_rs, dvo, tile, dst, std::false_type{});
            }
        };
        auto dma_wait = [&]() __attribute__((always_inline)) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); };
        if constexpr (PRIO == 2) { if (!groupA) __builtin_amdgcn_s_setprio(1); }
        if constexpr (DMA && (AIS == 1 || AIS == 3)) {         // group A is the issuer throughout
            if (groupA) { dma_fetch(0, ntiles); dma_wait(); }      // K(0) only
            seg_barrier();
            if (groupA) dma_fetch(1, 0);                           // waited for at the end of A's Y(0)
        } else if constexpr (DMA) {
            if (!groupA) { dma_fetch(0, ntiles); dma_wait(); }     // K(0) only
            seg_barrier();
            if (!groupA) dma_fetch(1, 0);
        } else {
            fetch(0, ntiles);                          // K(0) only
            publish(0, ntiles);
            seg_barrier();
            fetch(1, 0);                               // segment 0 (even): K(1), V(0) in flight
        }
        if (!groupA) seg_barrier();                    // B idles through segment 0
        UG_ASTAMP(1);
        do_QK(0, std::integral_constant<int, 0>{});    // A: segment 0 | B: segment 1
        if (!groupA) { if constexpr (DMA && AIS != 1 && AIS != 3) dma_wait(); else if constexpr (!DMA) publish(1, 0); }    // end of segment 1 (B)
        seg_barrier();
        // one tile = Y(t) | X(t); buffer parity is a compile-time constant (two tiles per trip)
        auto tile = [&](int t, auto cur_c, auto k3_c) __attribute__((always_inline)) {
            constexpr int K3 = decltype(k3_c)::value;            // t % 3 (AIS 5: the K ring slot of tile t; the loop is unrolled over 6 tiles so that it is a constant)
            // Y(t): A in odd segment 2t+1 (publishes K(t+1), V(t) at its end) | B in even segment 2t+2 (fetches K(t+2), V(t+1) at its start)
            if (!groupA) { if constexpr (DMA && AIS == 0) dma_fetch(t + 2, t + 1); else if constexpr (DMA && (AIS == 2 || AIS == 4 || AIS == 5)) dma_fetch(ntiles, t + 1); else if constexpr (!DMA) fetch(t + 2, t + 1); }
            // AIS 5: group A issues K(t+2) at the START of its softmax segment (2t+1) into ring slot (t+2) % 3, whose last readers (QK(t-1): segments 2t-2, 2t-1) are
            // done - with two slots it would be the buffer group B's X(t-1) reads K(t) from right now; group B keeps V(t+1) (start of ITS softmax segment, 2t+2)
            if constexpr (DMA && AIS == 5) { if (groupA) dma_fetch(t + 2, ntiles); }
            if constexpr (PRIO == 3) __builtin_amdgcn_s_setprio(1);                  // the softmax segment outranks the partner's matrix stream at issue
            do_SM();
            // P^T is "used" here: hipcc otherwise sinks the (pure) scale / exp2 / pack chain across the barrier to its first use, the
            // P.V MFMAs - i.e. out of this VALU-only segment into the matrix-only one, which then ran at ~60 cycles per MFMA
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb) { asm volatile("" : "+v"(pf[kb][0])); asm volatile("" : "+v"(pf[kb][1])); }
            if constexpr (LSUM) asm volatile("" : "+v"(m_run)); else asm volatile("" : "+v"(l_run), "+v"(m_run));
            if constexpr (PRIO == 3) __builtin_amdgcn_s_setprio(0);
            // group A publishes K(t+1), V(t) and at once re-fills the staging registers with K(t+2), V(t+1): its VALU segment has slack
            // (the partner's matrix segment is longer), whereas a fetch at the head of its own X(t) delayed the first MFMA
            if constexpr (!DMA) { if (groupA) { publish(t + 1, t); fetch(t + 2, t + 1); } }
            if constexpr (DMA && AIS == 5) {
                // K(t+1) (issued one tile ago) must have landed before X(t); K(t+2), issued at the top of this segment, may stay in flight: NIW pieces
                if (groupA) { if (t + 2 < ntiles) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NIW) : "memory"); else dma_wait(); }
            } else if constexpr (DMA && AIS != 0) { if (groupA) dma_wait(); }      // A's DMAs of the end of X(t-1): K(t+1) (and V(t), AIS 1), first read in X(t)
            UG_SEG(0);
            seg_barrier();
            UG_SEG(1);
            // X(t) = P.V(t) then K.Q^T(t+1): A in even segment 2t+2 | B in odd segment 2t+3 (publish). (Measured and dropped: group B
            // reading its first V^T fragments ahead of the barrier, inside its softmax segment: -4 %, -10 % with two k-steps.)
            if constexpr (DMA && AIS == 5) {
                constexpr int KN = (K3 + 1) % 3;
                do_X(t, cur_c, t + 1 < ntiles, [](int) {}, std::integral_constant<int, (KN < 2 ? KN * 2 * TILE : 4 * TILE)>{});
            } else if constexpr (DMA && AIS == 4) {
                // split: group A issues the two K(t+2) pieces behind the first two P.V k-steps of its matrix segment, group B the V(t+1) tile at the start of its softmax segment
                do_X(t, cur_c, t + 1 < ntiles, [&](int ks) __attribute__((always_inline)) { if (groupA && ks < 2) dma_piece(t + 2, ntiles, ks); }, std::integral_constant<int, -1>{});
            } else if constexpr (DMA && AIS == 3) {
                // group A: the four pieces of K(t+2) / V(t+1), one behind the MFMAs of each P.V k-step of ITS matrix segment (buffer form: two scalar
                // instructions + the DMA each); a ragged or missing tile takes the whole-tile path behind step 0 (K) / step 2 (V)
                do_X(t, cur_c, t + 1 < ntiles, [&](int ks) __attribute__((always_inline)) { if (groupA) dma_piece(t + 2, t + 1, ks); }, std::integral_constant<int, -1>{});
            } else {
                do_X(t, cur_c, t + 1 < ntiles, [](int) {}, std::integral_constant<int, -1>{});
            }
            if (!groupA) { if constexpr (DMA && AIS != 1 && AIS != 3) dma_wait(); else if constexpr (!DMA) publish(t + 2, t + 1); }
            if constexpr (DMA && AIS == 1) { if (groupA) dma_fetch(t + 2, t + 1); }
            if constexpr (DMA && AIS == 2) { if (groupA) dma_fetch(t + 2, ntiles); }
            UG_SEG(2);
            seg_barrier();
            UG_SEG(3);
        };
        UG_SEG0();
        if constexpr (AIS == 5) {
            for (int t = 0; t < ntiles; t += 6) {                  // buffer parity AND K ring slot as compile-time constants: period 6
                tile(t, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
                if (t + 1 < ntiles) tile(t + 1, std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{});
                if (t + 2 < ntiles) tile(t + 2, std::integral_constant<int, 0>{}, std::integral_constant<int, 2>{});
                if (t + 3 < ntiles) tile(t + 3, std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{});
                if (t + 4 < ntiles) tile(t + 4, std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{});
                if (t + 5 < ntiles) tile(t + 5, std::integral_constant<int, 1>{}, std::integral_constant<int, 2>{});
            }
        } else {
            for (int t = 0; t < ntiles; t += 2) {
                tile(t, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
                if (t + 1 < ntiles) tile(t + 1, std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{});
            }
        }
        UG_ASTAMP(2);
        if (groupA) seg_barrier();                     // A's trailing (empty) segment pairs with B's last one
    }

    // ---- epilogue: O[q][d] = O^T / l ----
    if constexpr (LSUM) l_run = lacc[0] + lacc2[0];
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    // training: log2 sum_k 2^(c s) of the row for the backward kernels (m_run is the row's reference point, shared by both lane halves)
#ifndef UG_ATTN_STAMPS
    if (lse_out != nullptr && h == 0 && q_row < Lq) lse_out[(int64_t)bh * lse_ld + q_row] = __builtin_amdgcn_logf(l_tot) + m_run * c;
#endif
    const float inv = 1.0f / l_tot;
    if constexpr (WIDE) {
        // Lane (r, h) holds, per 8-column group g4 of a 32-wide d block, columns 8 g4 + 4 h .. + 3 of its query row (8 bytes). One
        // v_permlane32_swap per dword on the group pair (k, k + 1) moves the upper half-wave's group-k data down and the lower half's
        // group-(k + 1) data up: lanes 0-31 then hold columns 8k .. 8k + 7 and lanes 32-63 columns 8k + 8 .. 8k + 15 of the row: ONE
        // 16-byte store per pair instead of two 8-byte ones (cdna guide T21: the store tail is issue-bound). Rows past Lq only skip the store.
        bf16_t* Orow = o + (int64_t)b * o_bs + (int64_t)(q_row < Lq ? q_row : Lq - 1) * o_rs + head * DH + 8 * h;
#pragma unroll
        for (int db = 0; db < NDB; ++db)
#pragma unroll
            for (int k2 = 0; k2 < 4; k2 += 2) {
                unsigned ax = pack2bf(oacc[db][4 * k2 + 0] * inv, oacc[db][4 * k2 + 1] * inv), ay = pack2bf(oacc[db][4 * k2 + 2] * inv, oacc[db][4 * k2 + 3] * inv);
                unsigned bx = pack2bf(oacc[db][4 * k2 + 4] * inv, oacc[db][4 * k2 + 5] * inv), by = pack2bf(oacc[db][4 * k2 + 6] * inv, oacc[db][4 * k2 + 7] * inv);
                auto rx = __builtin_amdgcn_permlane32_swap(ax, bx, false, false);
                auto ry = __builtin_amdgcn_permlane32_swap(ay, by, false, false);
                u32x4 w; w.x = rx[0]; w.y = ry[0]; w.z = rx[1]; w.w = ry[1];
                if (q_row < Lq) *(u32x4*)(Orow + 32 * db + 8 * k2) = w;
            }
    } else if (q_row < Lq) {
        bf16_t* Orow = o + (int64_t)b * o_bs + (int64_t)q_row * o_rs + head * DH;
#pragma unroll
        for (int db = 0; db < NDB; ++db)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                u32x2 w;
                w.x = pack2bf(oacc[db][4 * g4 + 0] * inv, oacc[db][4 * g4 + 1] * inv);
                w.y = pack2bf(oacc[db][4 * g4 + 2] * inv, oacc[db][4 * g4 + 3] * inv);
                *(u32x2*)(Orow + 32 * db + 8 * g4 + 4 * h) = w;
            }
    }
#ifdef UG_ATTN_STAMPS
    UG_ASTAMP(3);
    if (lse_out != nullptr && lane == 0 && (wave & (NW / 2 - 1)) == 0) {
        unsigned long long* d = (unsigned long long*)lse_out + ((int64_t)blockIdx.x * 2 + wave / (NW / 2)) * 12;
        d[8] = ug_seg[0]; d[9] = ug_seg[1]; d[10] = ug_seg[2]; d[11] = ug_seg[3];
        d[0] = ug_st[0]; d[1] = ug_st[1]; d[2] = ug_st[2]; d[3] = ug_st[3]; d[4] = ug_rt0; d[5] = __builtin_amdgcn_s_memrealtime();
        d[6] = ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32) | (unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 4);   // XCC_ID | HW_ID
        d[7] = (unsigned long long)logical;
    }
#endif
#undef UG_ASTAMP
#undef UG_SEG0
#undef UG_SEG
}



#ifdef UG_PROBE_BUILD   // round 4: +3.5...+4.2 % alone, -5.6 % inside the forward (see the dispatcher): probe library only
// =====================================================================================================================
// Round 4: the X|Y stagger kernel on v_mfma_f32_16x16x32_bf16 - the bf16 shape this chip clocks ~1.12-1.15x higher than 32x32x16 at equal
// cycles per FLOP (MI355X_MICROARCH "DVFS give-back" item 7; tools/probe/coexec4.hip priced this loop's skeleton at +3...+6 %).
// Same workgroup (8 waves x 32 query rows), same 64-key tiles, same swizzled LDS image, same LDS-DMA staging, same segment structure;
// what changes is the operand geometry (lane = (i, g), i = lane & 15, g = lane >> 4):
//   S^T tile (kt, qb) = K[16 keys] Q^T[16 queries], 32 of d per MFMA:  A = K row key(kt, i), chunk 4 s + g (ds_read_b128);  B = Q row 16 qb + i
//     from registers;  D: lane holds S^T[key(kt, 4 g + r)][query 16 qb + i], r = 0..3.
//   O^T tile (db, qb) = V^T[16 d] P^T[32 keys]:  B = this lane's OWN S^T registers of the key-tile pair (2 ks, 2 ks + 1), packed to bf16 - element
//     j of lane group g is key slot (kt = 2 ks + (j >> 2), row 4 g + (j & 3)) - no LDS, no cross-lane traffic for P;  A = V^T, two
//     ds_read_b64_tr_b16 per (ks, db): the 4-key blocks key(2 ks, 4 g ..) and key(2 ks + 1, 4 g ..) of columns 16 db .. + 15.
//   Every LDS fragment feeds TWO MFMAs (qb = 0, 1), so LDS reads, VGPRs and MFMA cycles per tile equal the 32x32x16 kernel's
//   (16 ds_read_b128 + 32 tr reads, 64 x 16 instead of 32 x 32 MFMA cycles per tile and wave).
//   key(kt, rho) = 16 kt + ((rho - 4) & 15): the rotation makes BOTH read kinds conflict-free on the shared image at head width 128 - a
//   ds_read_b128 lane group {i in 0-3, 12-15 of g; i in 4-11 of g + 1} covers all 16 slots of the bank row iff the rows read by lanes 4-11 are
//   closed under row ^ 4 (f(row) ^ 1 = f(row ^ 4) for the image's f), and a transposed read's 32-lane half takes rows {12-15, 0-3} or
//   {4-7, 8-11}, whose slot pairs f(row) >> 1 are distinct. (With key = 16 kt + rho the row reads are 2-way, cdna guide T10.)
//   Softmax: a query's scores are spread over the 4 lane groups - row max and final row sum take one v_permlane16_swap + one v_permlane32_swap.
// =====================================================================================================================
__device__ __forceinline__ float ug_max_groups(float x) {      // max over lanes l, l ^ 16, l ^ 32, l ^ 48, in every lane
    unsigned u = __float_as_uint(x);
    const auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    x = __builtin_fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
    u = __float_as_uint(x);
    const auto b = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return __builtin_fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}
__device__ __forceinline__ float ug_sum_groups(float x) {
    unsigned u = __float_as_uint(x);
    const auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    x = __uint_as_float(a[0]) + __uint_as_float(a[1]);
    u = __float_as_uint(x);
    const auto b = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}

template <int DH, int PRIO>
__global__ __launch_bounds__(512, 2) void flash_attn_m16_kernel(
    const bf16_t* __restrict__ q, int64_t q_rs, int64_t q_bs, const bf16_t* __restrict__ k, int64_t k_rs, int64_t k_bs,
    const bf16_t* __restrict__ v, int64_t v_rs, int64_t v_bs, bf16_t* __restrict__ o, int64_t o_rs, int64_t o_bs,
    int heads, int Lq, int Lkv, int nQ, float c /* softmax_scale * log2(e) */, float* __restrict__ lse_out /* nullable */, int64_t lse_ld) {
    constexpr int KVB = 64, RB = 2 * DH, NCH = DH / 8, TILE = KVB * RB;
    constexpr int NS = DH / 32;                      // k-steps (32 of d) of S^T = K Q^T
    constexpr int NDB = DH / 16;                     // 16-wide d blocks of O^T
    constexpr int NKT = KVB / 16;                    // 16-key tiles of S^T per K/V tile
    constexpr int NKS = KVB / 32;                    // k-steps (32 keys) of O^T += V^T P^T
    constexpr int NU = NKS * (NDB / 4);              // P.V steps of 8 MFMAs (4 d blocks x 2 query blocks)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // [2][K tile | V tile]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 15, g = lane >> 4;

    const int nwg = gridDim.x;
    const int qd = nwg >> 3, rm = nwg & 7;
    const int xcd = blockIdx.x & 7, kk = blockIdx.x >> 3;
    const int logical = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + kk;
    const int qt = logical % nQ;
    const int bh = logical / nQ;
    const int head = bh % heads, b = bh / heads;
    const bf16_t* Qb = q + (int64_t)b * q_bs + head * DH;
    const bf16_t* Kb = k + (int64_t)b * k_bs + head * DH;
    const bf16_t* Vb = v + (int64_t)b * v_bs + head * DH;

    // ---- Q fragments (B operand of S^T): lane (i, g) holds Q[16 qb + i][32 s + 8 g + j] ----
    bf16x8 qf[2][NS];
    int q_row[2];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        q_row[qb] = qt * 256 + wave * 32 + 16 * qb + i;
        const int q_ld = q_row[qb] < Lq ? q_row[qb] : Lq - 1;
#pragma unroll
        for (int s = 0; s < NS; ++s) qf[qb][s] = *(const bf16x8*)(Qb + (int64_t)q_ld * q_rs + 32 * s + 8 * g);
    }
    // retire the Q loads here (see flash_attn_kernel): otherwise every loop iteration re-waits for them and drains the K/V prefetch
    if constexpr (NS == 4)
        asm volatile("" : "+v"(qf[0][0]), "+v"(qf[0][1]), "+v"(qf[0][2]), "+v"(qf[0][3]), "+v"(qf[1][0]), "+v"(qf[1][1]), "+v"(qf[1][2]), "+v"(qf[1][3]));
    else
        asm volatile("" : "+v"(qf[0][0]), "+v"(qf[0][1]), "+v"(qf[1][0]), "+v"(qf[1][1]));

    // ---- per-lane LDS read offsets ----
    const int pi_i = (i - 4) & 15;                   // S^T-tile row i of this lane <-> image row 16 kt + pi_i
    const int k_rowoff = RB * pi_i;
    const int fk = row_swz<DH>(pi_i);
    // transposed V read: lane 4 qq + pp of group g supplies row rho = (4 g + qq - 4) & 15 (+ 16 kt), columns 16 db + 4 pp .. + 3
    const int qq = i >> 2, pp = i & 3;
    const int rho = (4 * g + qq - 4) & 15;
    int voff[NDB];
#pragma unroll
    for (int db = 0; db < NDB; ++db) voff[db] = RB * rho + 16 * ((2 * db + (pp >> 1)) ^ row_swz<DH>(rho)) + 8 * (pp & 1);

    f32x4 oacc[NDB][2];
#pragma unroll
    for (int db = 0; db < NDB; ++db)
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) oacc[db][qb] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float m_run[2] = {-INFINITY, -INFINITY}, l_run[2] = {0.f, 0.f};
    const int ntiles = (Lkv + KVB - 1) / KVB;
    bf16x8 pf[NKS][2];                                 // P^T fragments of the tile between its softmax and its P.V
    f32x4 sacc[NKT][2];                                // S^T of the tile between its K Q^T and its softmax

    auto mask_ragged = [&](int kv0) __attribute__((always_inline)) {
        if (kv0 + KVB > Lkv) {
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int key = kv0 + 16 * kt + ((4 * g + r - 4) & 15);
                    if (key >= Lkv) { sacc[kt][0][r] = -INFINITY; sacc[kt][1][r] = -INFINITY; }
                }
        }
    };
    auto do_QK0 = [&]() __attribute__((always_inline)) {     // tile 0 (buffer 0): no P.V before it
        const unsigned char* Kbuf = smem;
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) { sacc[kt][0] = (f32x4){0.f, 0.f, 0.f, 0.f}; sacc[kt][1] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt) {
                const bf16x8 kf = *(const bf16x8*)(Kbuf + kt * 16 * RB + k_rowoff + 16 * ((4 * s + g) ^ fk));
                sacc[kt][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[0][s], sacc[kt][0], 0, 0, 0);
                sacc[kt][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[1][s], sacc[kt][1], 0, 0, 0);
            }
        mask_ragged(0);
    };
    auto do_SM = [&]() __attribute__((always_inline)) {
        float tmax[2];
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            float t = sacc[0][qb][0];
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt) {
                t = ug_max3(t, sacc[kt][qb][0], sacc[kt][qb][1]);
                t = ug_max3(t, sacc[kt][qb][2], sacc[kt][qb][3]);
            }
            tmax[qb] = ug_max_groups(t);
        }
        // lazy reference point, as in flash_attn_kernel: a row's running max moves only when the tile max exceeds it by more than 2^8
        const bool up0 = (tmax[0] - m_run[0]) * c > 8.0f, up1 = (tmax[1] - m_run[1]) * c > 8.0f;
        if (!__all(!(up0 || up1))) {
            const float mn0 = up0 ? tmax[0] : m_run[0], mn1 = up1 ? tmax[1] : m_run[1];
            const float a0 = __builtin_amdgcn_exp2f((m_run[0] - mn0) * c), a1 = __builtin_amdgcn_exp2f((m_run[1] - mn1) * c);
            l_run[0] *= a0; l_run[1] *= a1;
#pragma unroll
            for (int db = 0; db < NDB; ++db)
#pragma unroll
                for (int r = 0; r < 4; ++r) { oacc[db][0][r] *= a0; oacc[db][1][r] *= a1; }
            m_run[0] = mn0; m_run[1] = mn1;
        }
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            const float mc = m_run[qb] * c;
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
                float p[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    p[j] = __builtin_amdgcn_exp2f(fmaf(sacc[2 * ks + (j >> 2)][qb][j & 3], c, -mc));
                    l_run[qb] += p[j];
                }
                u32x4 w;
                w.x = pack2bf(p[0], p[1]); w.y = pack2bf(p[2], p[3]); w.z = pack2bf(p[4], p[5]); w.w = pack2bf(p[6], p[7]);
                pf[ks][qb] = __builtin_bit_cast(bf16x8, w);
            }
        }
    };
    // X(t) = P.V(t) then S^T(t + 1) = K Q^T as an explicit, fenced stream: NU steps of 8 P.V MFMAs, then NS steps of 8 K Q^T MFMAs; every LDS
    // fragment is requested two steps (256 MFMA cycles) ahead of its MFMAs
    auto do_X = [&](int t, auto cur_c, bool have_qk) __attribute__((always_inline)) {
        constexpr int CUR = decltype(cur_c)::value;
        const unsigned char* Vbuf = smem + CUR * 2 * TILE + TILE;
        const unsigned char* Kbuf = smem + (CUR ^ 1) * 2 * TILE;
        bf16x8 vf[NU][4], kf[NS][NKT];
        auto rdv = [&](int u) __attribute__((always_inline)) {
            const int ks = u / (NDB / 4), db0 = 4 * (u % (NDB / 4));
#pragma unroll
            for (int d = 0; d < 4; ++d)
                vf[u][d] = tr_read_pair(Vbuf + (2 * ks) * 16 * RB + voff[db0 + d], Vbuf + (2 * ks + 1) * 16 * RB + voff[db0 + d]);
        };
        auto rdk = [&](int s) __attribute__((always_inline)) {
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt) kf[s][kt] = *(const bf16x8*)(Kbuf + kt * 16 * RB + k_rowoff + 16 * ((4 * s + g) ^ fk));
        };
        rdv(0);
        if constexpr (NU > 1) rdv(1);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (PRIO == 1) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int ks = u / (NDB / 4), db0 = 4 * (u % (NDB / 4));
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                oacc[db0 + d][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf[u][d], pf[ks][0], oacc[db0 + d][0], 0, 0, 0);
                oacc[db0 + d][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf[u][d], pf[ks][1], oacc[db0 + d][1], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (u + 2 < NU) rdv(u + 2);
            else if (have_qk && u + 2 - NU < NS) rdk(u + 2 - NU);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (!have_qk) { if constexpr (PRIO == 1) __builtin_amdgcn_s_setprio(0); return; }
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) { sacc[kt][0] = (f32x4){0.f, 0.f, 0.f, 0.f}; sacc[kt][1] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int s = 0; s < NS; ++s) {
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt) {
                sacc[kt][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[s][kt], qf[0][s], sacc[kt][0], 0, 0, 0);
                sacc[kt][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[s][kt], qf[1][s], sacc[kt][1], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (s + 2 < NS) { rdk(s + 2); __builtin_amdgcn_sched_barrier(0); }
        }
        if constexpr (PRIO == 1) __builtin_amdgcn_s_setprio(0);
        mask_ragged((t + 1) * KVB);
    };

    // ---- X | Y stagger with LDS-DMA staging: identical orchestration to flash_attn_kernel<.., STAGGER, .., DMA> ----
    auto seg_barrier = [&]() __attribute__((always_inline)) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
    const bool groupA = __builtin_amdgcn_readfirstlane(wave) < 4;
    constexpr int RPI = 1024 / RB, NI = TILE / 1024, NIW = NI / 4;
    const int wb = __builtin_amdgcn_readfirstlane(wave) & 3;
    unsigned dko[NIW], dvo[NIW];
#pragma unroll
    for (int u = 0; u < NIW; ++u) {
        const int row = (wb * NIW + u) * RPI + lane / NCH;
        const int ch = (lane % NCH) ^ row_swz<DH>(row);
        dko[u] = (unsigned)(row * (int)k_rs + ch * 8) * 2u;        // bytes
        dvo[u] = (unsigned)(row * (int)v_rs + ch * 8) * 2u;
    }
    auto dma_tile = [&](const bf16_t* base, int64_t rs, const unsigned (&off)[NIW], int tile, unsigned dst) {
        if (tile * KVB + KVB <= Lkv) {
            const void* tb = uniform_ptr(base + (int64_t)tile * KVB * rs);
#pragma unroll
            for (int u = 0; u < NIW; ++u) glds16_off(tb, off[u], dst + u * 1024);
        } else {                                   // ragged last tile: rows past the end re-read the last key (masked in S^T)
            int lane_r = lane;
            asm volatile("" : "+v"(lane_r));
#pragma unroll
            for (int u = 0; u < NIW; ++u) {
                const int row = (wb * NIW + u) * RPI + lane_r / NCH;
                const int ch = (lane_r % NCH) ^ row_swz<DH>(row);
                int key = tile * KVB + row; if (key > Lkv - 1) key = Lkv - 1;
                glds16_ptr(base + (int64_t)key * rs + ch * 8, dst + u * 1024);
            }
        }
    };
    auto dma_fetch = [&](int kt, int vt) __attribute__((always_inline)) {
        const unsigned l0 = __builtin_amdgcn_readfirstlane(lds_addr(smem)) + wb * NIW * 1024;
        if (kt < ntiles) dma_tile(Kb, k_rs, dko, kt, l0 + (kt & 1) * 2 * TILE);
        if (vt < ntiles) dma_tile(Vb, v_rs, dvo, vt, l0 + (vt & 1) * 2 * TILE + TILE);
    };
    auto dma_wait = [&]() __attribute__((always_inline)) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); };
    if (!groupA) { dma_fetch(0, ntiles); dma_wait(); }     // K(0) only
    seg_barrier();
    if (!groupA) dma_fetch(1, 0);
    if (!groupA) seg_barrier();                    // B idles through segment 0
    do_QK0();                                      // A: segment 0 | B: segment 1
    if (!groupA) dma_wait();
    seg_barrier();
    auto tile = [&](int t, auto cur_c) __attribute__((always_inline)) {
        if (!groupA) dma_fetch(t + 2, t + 1);
        if constexpr (PRIO == 3) __builtin_amdgcn_s_setprio(1);
        do_SM();
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) { asm volatile("" : "+v"(pf[ks][0])); asm volatile("" : "+v"(pf[ks][1])); }
        asm volatile("" : "+v"(l_run[0]), "+v"(l_run[1]), "+v"(m_run[0]), "+v"(m_run[1]));
        if constexpr (PRIO == 3) __builtin_amdgcn_s_setprio(0);
        seg_barrier();
        do_X(t, cur_c, t + 1 < ntiles);
        if (!groupA) dma_wait();
        seg_barrier();
    };
    for (int t = 0; t < ntiles; t += 2) {
        tile(t, std::integral_constant<int, 0>{});
        if (t + 1 < ntiles) tile(t + 1, std::integral_constant<int, 1>{});
    }
    if (groupA) seg_barrier();                     // A's trailing (empty) segment pairs with B's last one

    // ---- epilogue: O[q][d] = O^T / l. Lane (i, g) holds, per (db, qb), columns 16 db + 4 g .. + 3 of row 16 qb + i ----
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        const float l_tot = ug_sum_groups(l_run[qb]);
        if (lse_out != nullptr && g == 0 && q_row[qb] < Lq) lse_out[(int64_t)bh * lse_ld + q_row[qb]] = __builtin_amdgcn_logf(l_tot) + m_run[qb] * c;
        const float inv = 1.0f / l_tot;
        // v_permlane16_swap on the d-block pair (db, db + 1): afterwards an even lane group holds columns 16 db + 4 g .. + 7 (its own block-db
        // data and group g + 1's), an odd one columns 16 (db + 1) + 4 (g - 1) .. + 7: ONE 16-byte store per pair (cdna guide T21)
        bf16_t* Orow = o + (int64_t)b * o_bs + (int64_t)(q_row[qb] < Lq ? q_row[qb] : Lq - 1) * o_rs + head * DH + ((g & 1) ? 16 + 4 * (g - 1) : 4 * g);
#pragma unroll
        for (int db = 0; db < NDB; db += 2) {
            unsigned ax = pack2bf(oacc[db][qb][0] * inv, oacc[db][qb][1] * inv), ay = pack2bf(oacc[db][qb][2] * inv, oacc[db][qb][3] * inv);
            unsigned bx = pack2bf(oacc[db + 1][qb][0] * inv, oacc[db + 1][qb][1] * inv), by = pack2bf(oacc[db + 1][qb][2] * inv, oacc[db + 1][qb][3] * inv);
            const auto rx = __builtin_amdgcn_permlane16_swap(ax, bx, false, false);
            const auto ry = __builtin_amdgcn_permlane16_swap(ay, by, false, false);
            u32x4 w; w.x = rx[0]; w.y = ry[0]; w.z = rx[1]; w.w = ry[1];
            if (q_row[qb] < Lq) *(u32x4*)(Orow + 16 * db) = w;
        }
    }
}

#endif   // UG_PROBE_BUILD

#ifdef UG_PROBE_BUILD   // measured 4 % behind the 8-wave stagger: kept for A/B in the probe library only
// =====================================================================================================================
// One wave per SIMD ("pwg"): 4 waves x 64 query rows, up to 512 registers per lane, software-pipelined inside the wave.
//
// The 8-wave loop above keeps the matrix pipe ~42 % busy: its two waves per SIMD reach the MFMA segments and the softmax
// segments together. Here a single in-order wave per SIMD overlaps the two itself:
//     iteration t:   S1 = [ online softmax of S(t) (VALU)  interleaved with  O^T += V^T P^T of tile t-1 (32 MFMAs) ]
//                    S2 = [ S^T(t+1) = K Q^T (32 MFMAs) ]
// Per 64-key tile a wave issues 64 MFMAs for 64 query rows (every K / V fragment read from LDS feeds two MFMAs - half the LDS
// traffic per FLOP of the 32-row waves) against ~260 VALU instructions placed in the MFMA gaps of S1.
// LDS: 2 K slots + 2 V slots (64 KB), one barrier per tile: iteration t writes K(t+2) and V(t) (fetched to registers one
// iteration earlier) into the slots whose last readers finished before the barrier at its top, and fetches K(t+3), V(t+1).
// =====================================================================================================================
template <int DH>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void flash_attn_pwg_kernel(
    const bf16_t* __restrict__ q, int64_t q_rs, int64_t q_bs, const bf16_t* __restrict__ k, int64_t k_rs, int64_t k_bs,
    const bf16_t* __restrict__ v, int64_t v_rs, int64_t v_bs, bf16_t* __restrict__ o, int64_t o_rs, int64_t o_bs,
    int heads, int Lq, int Lkv, int nQ, float c /* softmax_scale * log2(e) */) {
    constexpr int RB = 2 * DH, NCH = DH / 8, TILE = KVB * RB, QS = DH / 16, NDB = DH / 32;
    constexpr int NT = 256, NST = (KVB * NCH) / NT;
    static_assert(NST >= 1, "tile smaller than the workgroup");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // K slot 0 | K slot 1 | V slot 0 | V slot 1
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;

    const int nwg = gridDim.x;
    const int qd = nwg >> 3, rm = nwg & 7;
    const int xcd = blockIdx.x & 7, kk = blockIdx.x >> 3;
    const int logical = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + kk;
    const int qt = logical % nQ;
    const int bh = logical / nQ;
    const int head = bh % heads, b = bh / heads;
    const bf16_t* Qb = q + (int64_t)b * q_bs + head * DH;
    const bf16_t* Kb = k + (int64_t)b * k_bs + head * DH;
    const bf16_t* Vb = v + (int64_t)b * v_bs + head * DH;

    // ---- Q fragments of the wave's two 32-row blocks (B operand of S^T = K Q^T): lane (r, h) holds Q[r][16 s + 8 h + j]. They are parked
    // in AGPRs (hipcc does not feed MFMA B operands from AGPRs) and copied to VGPRs one k-step ahead of their MFMAs, 8 v_accvgpr_read per
    // step in the MFMA shadow. Re-reading them from LDS instead doubled the QK^T segment's LDS traffic to ~75 % of the LDS array. ----
    bf16x8 qf[2][QS];
    int q_row[2];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        q_row[qb] = qt * 256 + wave * 64 + qb * 32 + r;
        const int q_ld = q_row[qb] < Lq ? q_row[qb] : Lq - 1;
#pragma unroll
        for (int s = 0; s < QS; ++s) qf[qb][s] = *(const bf16x8*)(Qb + (int64_t)q_ld * q_rs + 16 * s + 8 * h);
    }
#pragma unroll
    for (int qb = 0; qb < 2; ++qb)      // retire the loads here (see the 8-wave kernel) and pin the class
#pragma unroll
        for (int s = 0; s < QS; ++s) asm volatile("" : "+a"(qf[qb][s]));
    // ---- staging: thread -> NST chunks of K and of V per tile ----
    int st_off[NST], st_row[NST], st_col[NST];
#pragma unroll
    for (int u = 0; u < NST; ++u) {
        const int cid = tid + NT * u;
        st_row[u] = cid / NCH;
        st_col[u] = (cid % NCH) * 8;
        st_off[u] = img_off<DH>(st_row[u], cid % NCH);
    }
    u32x4 kreg[NST], vreg[NST];
    auto fetch_k = [&](int kv0) {
#pragma unroll
        for (int u = 0; u < NST; ++u) { int key = kv0 + st_row[u]; if (key > Lkv - 1) key = Lkv - 1; kreg[u] = *(const u32x4*)(Kb + (int64_t)key * k_rs + st_col[u]); }
    };
    auto fetch_v = [&](int kv0) {
#pragma unroll
        for (int u = 0; u < NST; ++u) { int key = kv0 + st_row[u]; if (key > Lkv - 1) key = Lkv - 1; vreg[u] = *(const u32x4*)(Vb + (int64_t)key * v_rs + st_col[u]); }
    };
    auto write_k = [&](int slot) {
#pragma unroll
        for (int u = 0; u < NST; ++u) *(u32x4*)(smem + slot * TILE + st_off[u]) = kreg[u];
    };
    auto write_v = [&](int slot) {
#pragma unroll
        for (int u = 0; u < NST; ++u) *(u32x4*)(smem + (2 + slot) * TILE + st_off[u]) = vreg[u];
    };

    // ---- per-lane LDS read offsets (same images as the 8-wave kernel). Every swizzled offset is BASE ^ constant: the XOR only touches
    // bits 4-7, which the row term (multiple of 256) and the 8-byte term leave free. The segments re-derive their 8-16 addresses from
    // an opaque copy of the base (one v_xor each) - kept as loop invariants, hipcc held ~40 address registers and spilled them. ----
    static_assert(DH == 128, "XOR-folded offsets assume 256-byte rows");
    const int kx = h ^ row_swz<DH>(r);
    const int k_base = RB * r + 16 * kx;                                   // K fragment s, key block kb: kb * 32 * RB + (k_base ^ 32 s)
    const int i16 = lane & 15, g16 = lane >> 4;
    const int v_key = 4 * h + (i16 >> 2);
    const int v_lowch = 2 * (g16 & 1) + ((i16 & 3) >> 1);
    const int v_b8 = 8 * (i16 & 1);
    const int vlo_base = RB * v_key + 16 * (v_lowch ^ row_swz<DH>(v_key)) + v_b8;          // d-block db, k-step ks: ks * 16 * RB + (base ^ 64 db)
    const int vhi_base = RB * (v_key + 8) + 16 * (v_lowch ^ row_swz<DH>(v_key + 8)) + v_b8;

    f32x16 oacc[2][NDB];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb)
#pragma unroll
        for (int db = 0; db < NDB; ++db)
#pragma unroll
            for (int i = 0; i < 16; ++i) oacc[qb][db][i] = 0.f;
    // Register classes are pinned through empty asm operands: O^T accumulators, Q fragments and the staging registers are touched
    // only by MFMA / memory instructions and live in AGPRs; S^T and P^T are read and written by VALU and stay in VGPRs. Left to
    // itself hipcc accumulated S^T in AGPRs and moved ~500 registers per tile through v_accvgpr_read / _write.
    auto pin_o = [&]() {
#pragma unroll
        for (int qb = 0; qb < 2; ++qb)
#pragma unroll
            for (int db = 0; db < NDB; ++db) asm("" : "+a"(oacc[qb][db]));
    };
    pin_o();
    f32x16 sacc[2][2];                                   // [query block][key block] of the tile between its QK^T and its softmax
    float m_run[2] = {-INFINITY, -INFINITY}, l_run[2] = {0.f, 0.f}, alpha[2] = {1.f, 1.f};
    bf16x8 pf[2][2][2][2];                               // P^T fragments [tile parity][query block][key block][half]
    const int ntiles = (Lkv + KVB - 1) / KVB;

    // The segments below are written as explicit instruction streams: sched_barrier(0) after every piece keeps hipcc from
    // re-clumping them (left to the scheduler - with or without sched_group_barrier - the softmax VALU work ended up in runs of 60-70
    // instructions between MFMAs, and a lone wave per SIMD has nobody to cover a stalled pipe). Measured with s_memtime stamps: a
    // segment that is VALU-bound costs its VALU cycles + ~16 per MFMA, one that is MFMA-bound 32 per MFMA. Hence three segments:
    //   A: P.V(t-1), 32 MFMAs  | row maxima of both query blocks, then the 32 exp elements of query block 0
    //   B: S^T(t+1) of query block 0, 16 MFMAs into the S registers block 0 just released | the 32 exp elements of query block 1
    //   C: S^T(t+1) of query block 1, 16 MFMAs | the staging pieces (8 LDS writes, 8 global loads)
#define UG_FENCE() __builtin_amdgcn_sched_barrier(0)
    float mc[2], lsum[2];
    // one exp element of the lane: query block qb = e / 32, key block (e / 16) % 2; packs a finished group of 8 into pf[PP]
    auto sm_elems = [&](int e0, int e1, auto pp_c, float (&p)[2][32]) __attribute__((always_inline)) {
        constexpr int PP = decltype(pp_c)::value;
#pragma unroll
        for (int e = e0; e < e1; ++e) {
            const int qb = e >> 5, kb = (e >> 4) & 1, x = e & 15;
            const float pe = __builtin_amdgcn_exp2f(fmaf(sacc[qb][kb][x], c, -mc[qb]));
            p[qb][e & 31] = pe;
            lsum[qb] += pe;
            if ((e & 7) == 7) {
                const int g = e >> 3, s2 = g & 1;
                const float* pg = &p[qb][(g & 3) * 8];
                u32x4 w;
                w.x = pack2bf(pg[0], pg[1]); w.y = pack2bf(pg[2], pg[3]); w.z = pack2bf(pg[4], pg[5]); w.w = pack2bf(pg[6], pg[7]);
                pf[PP][qb][kb][s2] = __builtin_bit_cast(bf16x8, w);
            }
            if ((e & 31) == 31) l_run[qb] = l_run[qb] * alpha[qb] + lsum[qb];
        }
    };
    float pbuf[2][32];
    // Segment A. Returns whether any lane's reference point moved (then O is rescaled by alpha after this segment).
    auto do_A = [&](auto slot_c, auto pp_c, auto pv_c) __attribute__((always_inline)) -> int {
        constexpr int SLOT = decltype(slot_c)::value, PP = decltype(pp_c)::value;
        constexpr bool HAVE_PV = decltype(pv_c)::value;
        const unsigned char* Vbuf = smem + (2 + SLOT) * TILE;
        int vl0 = vlo_base, vh0 = vhi_base;
        asm volatile("" : "+v"(vl0), "+v"(vh0));
        // P.V order: k-step outer, then d-block, then query block: the 8 accumulators rotate
        bf16x8 vf[4][NDB];
        auto rdv = [&](int ks) {
#pragma unroll
            for (int db = 0; db < NDB; ++db)
                vf[ks][db] = tr_read_pair(Vbuf + ks * 16 * RB + (vl0 ^ (64 * db)), Vbuf + ks * 16 * RB + (vh0 ^ (64 * db)));
        };
        if constexpr (HAVE_PV) { rdv(0); UG_FENCE(); }
        int moved = 0;
        float tmax[2];
        lsum[0] = 0.f; lsum[1] = 0.f;
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            if constexpr (HAVE_PV) {
                const int ks = i >> 3, db = (i >> 1) & 3, qb = i & 1;
                oacc[qb][db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[ks][db], pf[PP ^ 1][qb][ks >> 1][ks & 1], oacc[qb][db], 0, 0, 0);
                UG_FENCE();
                if ((i & 7) == 1 && ks + 1 < 4) { rdv(ks + 1); UG_FENCE(); }
            }
            if (i < 8) {
                // maxima: piece i covers 8 of the 32 values of query block i / 4
                const int qb = i >> 2, part = i & 3, kb = part >> 1, o8 = 8 * (part & 1);
                const f32x16& sv = sacc[qb][kb];
                float m8 = ug_max3(sv[o8], sv[o8 + 1], sv[o8 + 2]);
                m8 = ug_max3(m8, sv[o8 + 3], sv[o8 + 4]);
                m8 = ug_max3(m8, sv[o8 + 5], sv[o8 + 6]);
                tmax[qb] = part == 0 ? ug_max3(m8, sv[o8 + 7], sv[o8 + 7]) : ug_max3(tmax[qb], m8, sv[o8 + 7]);
                if (part == 3) {
                    // Lazy running max: the reference point of a row moves only when the tile maximum exceeds it by more than 2^8 in
                    // the exponent (softmax is shift-invariant; P and l then stay below 2^8 per element, exact in fp32 / same relative
                    // precision in bf16). With an exact max some row of the 64 moves in nearly every tile and the O^T rescale - a
                    // round trip of 128 accumulators through VGPRs, ~2500 cycles - ran every iteration.
                    const float tm = ug_max_halves(tmax[qb]);
                    const bool up = (tm - m_run[qb]) * c > 8.0f;
                    const float m_new = up ? tm : m_run[qb];
                    moved |= !__all(!up);
                    alpha[qb] = __builtin_amdgcn_exp2f((m_run[qb] - m_new) * c);     // 1 where the point stayed, 0 on the first tile
                    m_run[qb] = m_new;
                    mc[qb] = m_new * c;
                }
            } else {
                const int j = i - 8;                     // 24 pieces x 4/3 elements: query block 0
                sm_elems((j * 4) / 3, ((j + 1) * 4) / 3, pp_c, pbuf);
            }
            UG_FENCE();
        }
        if constexpr (HAVE_PV) pin_o();
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) { asm volatile("" : "+v"(pf[PP][0][kb][0])); asm volatile("" : "+v"(pf[PP][0][kb][1])); }
        return moved;
    };
    // S^T(t)[query block QB] = K Q^T from K slot SLOT (16 MFMAs, k-step outer, the 2 key-block accumulators alternate). These MFMAs are
    // inline asm: S^T accumulates in VGPRs (where the softmax reads it) and the Q fragment comes straight from its AGPRs. As a builtin,
    // hipcc accumulates in AGPRs in a 512-register kernel and copies Q to VGPRs: +128 v_accvgpr_read and 64 more live registers per
    // tile. hipcc does not see an MFMA here, so the hazards are ours: operands come from ds_read / AGPRs (waitcnt is tracked through
    // the asm operands) and the results are first read by VALU behind the s_nops that close segment C or behind the next barrier.
    // FILL(ks) is called after the MFMAs of k-step ks: the VALU / memory work this segment shadows.
    auto do_QK = [&](int t, auto slot_c, auto qb_c, auto&& fill) __attribute__((always_inline)) {
        constexpr int SLOT = decltype(slot_c)::value, QB = decltype(qb_c)::value;
        const unsigned char* Kbuf = smem + SLOT * TILE;
        int kb0 = k_base;
        asm volatile("" : "+v"(kb0));
        constexpr int AHEAD = 3;
        bf16x8 kf[QS][2];
        auto rd = [&](int ks) {
            kf[ks][0] = *(const bf16x8*)(Kbuf + (kb0 ^ (32 * ks)));
            kf[ks][1] = *(const bf16x8*)(Kbuf + 32 * RB + (kb0 ^ (32 * ks)));
        };
#pragma unroll
        for (int ks = 0; ks < AHEAD; ++ks) rd(ks);
        UG_FENCE();
#pragma unroll
        for (int ks = 0; ks < QS; ++ks) {
            if (ks + AHEAD < QS) rd(ks + AHEAD);
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                if (ks == 0) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(sacc[QB][kb]) : "v"(kf[ks][kb]), "a"(qf[QB][ks]));
                else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(sacc[QB][kb]) : "v"(kf[ks][kb]), "a"(qf[QB][ks]));
            }
            UG_FENCE();
            fill(ks);
            UG_FENCE();
        }
    };
    auto mask_tail = [&](int t) __attribute__((always_inline)) {        // ragged last tile: keys >= Lkv do not exist
        asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 7" ::: "memory");   // the last MFMAs' results (8 passes) before any VALU read
        if (t * KVB + KVB > Lkv) {
#pragma unroll
            for (int qb = 0; qb < 2; ++qb)
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const int key = t * KVB + kb * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
                        if (key >= Lkv) sacc[qb][kb][i] = -INFINITY;
                    }
        }
    };
    // one staging piece of segment C of iteration t_it (computing S^T(t_it + 1)): pieces 0-7 publish K(t_it+2) / V(t_it) into the
    // slots of parity t_it & 1, pieces 8-15 fetch K(t_it+3) / V(t_it+1). Tiles past the end are clamped re-reads nobody consumes.
    auto stage_piece = [&](int pc, int t_it, int wslot) __attribute__((always_inline)) {
        constexpr int W = 2 * NST;
        if (pc < NST) *(u32x4*)(smem + wslot * TILE + st_off[pc]) = kreg[pc];
        else if (pc < W) *(u32x4*)(smem + (2 + wslot) * TILE + st_off[pc - NST]) = vreg[pc - NST];
        else if (pc < 2 * W) {
            const bool isk = pc < W + NST;
            const int u = isk ? pc - W : pc - W - NST;
            int tile = isk ? t_it + 3 : t_it + 1; if (tile > ntiles - 1) tile = ntiles - 1;              // wave-uniform
            const int rmax = Lkv - 1 - tile * KVB;
            const int row = st_row[u] < rmax ? st_row[u] : rmax;
            if (isk) kreg[u] = *(const u32x4*)(Kb + (int64_t)tile * KVB * k_rs + (unsigned)(row * (int)k_rs + st_col[u]));
            else vreg[u] = *(const u32x4*)(Vb + (int64_t)tile * KVB * v_rs + (unsigned)(row * (int)v_rs + st_col[u]));
        }
    };

    // ---- prologue: K(0), K(1) in LDS, S(0) computed, K(2) and V(0) in registers ----
    fetch_k(0);
    write_k(0);
    fetch_k(KVB);                                        // clamped re-read when there is a single tile
    write_k(1);
    __syncthreads();
    auto nofill = [&](int) {};
    do_QK(0, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, nofill);
    do_QK(0, std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{}, nofill);
    mask_tail(0);
    fetch_k(2 * KVB);
    fetch_v(0);

    auto rescale_o = [&]() __attribute__((always_inline)) {   // rare (a row's reference point moved): one accumulator at a time through VGPRs
#pragma unroll
        for (int qb = 0; qb < 2; ++qb)
#pragma unroll
            for (int db = 0; db < NDB; ++db) {
                f32x16 x = oacc[qb][db];
                asm volatile("" : "+v"(x));
#pragma unroll
                for (int i = 0; i < 16; ++i) x[i] *= alpha[qb];
                asm volatile("" : "+a"(x));
                oacc[qb][db] = x;
            }
    };
    // Iteration t: barrier | A | (rescale) | B | C. C also publishes K(t+2) -> K slot t & 1 and V(t) -> V slot t & 1 (their last
    // readers, QK(t) and P.V(t-2), finished before the barrier) and fetches K(t+3), V(t+1).
    auto iter = [&](int t, auto par_c, auto pv_c) __attribute__((always_inline)) {
        constexpr int PAR = decltype(par_c)::value;     // t & 1
        using CUR = std::integral_constant<int, PAR>;
        using OTH = std::integral_constant<int, PAR ^ 1>;
        __syncthreads();
        // S^T is "redefined" here so that its softmax cannot be hoisted above the barrier, away from the P.V MFMAs it must shadow
#pragma unroll
        for (int qb = 0; qb < 2; ++qb)
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) asm volatile("" : "+v"(sacc[qb][kb]));
        const int moved = do_A(OTH{}, CUR{}, pv_c);
        if (decltype(pv_c)::value && moved) rescale_o();
        if (t + 1 < ntiles) {
            // B: 2 exp elements of query block 1 per MFMA (hipcc hoists most of this pure chain up into segment A; pinning it here
            // measured 3 % slower)
            do_QK(t + 1, OTH{}, std::integral_constant<int, 0>{}, [&](int ks) __attribute__((always_inline)) { sm_elems(32 + 4 * ks, 36 + 4 * ks, CUR{}, pbuf); });
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) { asm volatile("" : "+v"(pf[PAR][1][kb][0])); asm volatile("" : "+v"(pf[PAR][1][kb][1])); }
            // C: 2 staging pieces per k-step
            do_QK(t + 1, OTH{}, std::integral_constant<int, 1>{}, [&](int ks) __attribute__((always_inline)) { stage_piece(2 * ks, t, PAR); stage_piece(2 * ks + 1, t, PAR); });
            mask_tail(t + 1);
        } else {
            sm_elems(32, 64, CUR{}, pbuf);               // last tile: the rest of its softmax, and V(t) is still to be published
            write_v(PAR);
        }
    };
    iter(0, std::integral_constant<int, 0>{}, std::false_type{});       // no P.V yet (O is zero: nothing to rescale)
    for (int t = 1; t < ntiles; t += 2) {
        iter(t, std::integral_constant<int, 1>{}, std::true_type{});
        if (t + 1 < ntiles) iter(t + 1, std::integral_constant<int, 0>{}, std::true_type{});
    }
#undef UG_FENCE
    __syncthreads();                                     // V(ntiles-1) visible
    auto tail_PV = [&](auto par_c) {                     // O^T += V^T P^T of the last tile
        constexpr int PAR = decltype(par_c)::value;
        const unsigned char* Vbuf = smem + (2 + PAR) * TILE;
        int vl0 = vlo_base, vh0 = vhi_base;
        asm volatile("" : "+v"(vl0), "+v"(vh0));
#pragma unroll
        for (int db = 0; db < NDB; ++db)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const bf16x8 vfr = tr_read_pair(Vbuf + ks * 16 * RB + (vl0 ^ (64 * db)), Vbuf + ks * 16 * RB + (vh0 ^ (64 * db)));
#pragma unroll
                for (int qb = 0; qb < 2; ++qb)
                    oacc[qb][db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vfr, pf[PAR][qb][ks >> 1][ks & 1], oacc[qb][db], 0, 0, 0);
            }
    };
    if ((ntiles - 1) & 1) tail_PV(std::integral_constant<int, 1>{}); else tail_PV(std::integral_constant<int, 0>{});

    // ---- epilogue: O[q][d] = O^T / l ----
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        const float l_tot = l_run[qb] + __shfl_xor(l_run[qb], 32, 64);
        const float inv = 1.0f / l_tot;
        if (q_row[qb] < Lq) {
            bf16_t* Orow = o + (int64_t)b * o_bs + (int64_t)q_row[qb] * o_rs + head * DH;
#pragma unroll
            for (int db = 0; db < NDB; ++db)
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    u32x2 w;
                    w.x = pack2bf(oacc[qb][db][4 * g4 + 0] * inv, oacc[qb][db][4 * g4 + 1] * inv);
                    w.y = pack2bf(oacc[qb][db][4 * g4 + 2] * inv, oacc[qb][db][4 * g4 + 3] * inv);
                    *(u32x2*)(Orow + 32 * db + 8 * g4 + 4 * h) = w;
                }
        }
    }
}


#endif   // UG_PROBE_BUILD

// =====================================================================================================================
// Attention BACKWARD (the autograd of F.scaled_dot_product_attention, src/UniGenUtils.py:601) on the forward kernel's
// tiling. One kernel, four modes; a workgroup OWNS 256 rows of one side (each wave 32, their fragments in registers as the B operand, the owned
// row index on the LANE) and STREAMS 64-row tiles of the other side through the forward's swizzled LDS image (row reads for the score-like
// products, transposed reads for the accumulating product):
//   LSE : owns queries,  streams K       : S^T = K Q^T                         -> lse2[q] = log2 sum_k 2^(c S)          (lane-local statistics)
//   DQ  : owns queries,  streams K, V    : S^T = K Q^T, dP^T = V dO^T, dS^T = P^T (dP^T - delta)  -> dQ^T += K^T dS^T, x scale at the end
//   DK  : owns keys,     streams Q, dO   : S   = Q K^T, dP   = dO V^T, dS   = P   (dP   - delta)  -> dK^T += Q^T dS,     x scale at the end
//   DV  : owns keys,     streams Q, dO   : S   = Q K^T, P                                               -> dV^T += dO^T P
// with P = 2^(c S - lse2[q]), c = scale log2(e), delta[q] = sum_d dO[q][d] O[q][d]. In every mode the 32x32 accumulator of a score-like product
// has the streamed index on its rows and the owned index on its lanes, so its registers, packed to bf16, ARE the B operand of the accumulating
// product (the forward's P^T trick); lse / delta are per lane when queries are owned and per accumulator row when they are streamed. S is
// recomputed per mode (8 product units against the minimum of 5) - still ~4x less time than moving fp32 score matrices through HBM.
// Streamed tiles (and, when queries are streamed, their 64 lse2 / delta values) reach LDS by LDS-DMA one tile ahead; at dh 128 the LDS reads
// of every matrix phase are software-pipelined by hand. Measured (tools/attn_bwd_ab.py, profiles/r02f_attn_bwd.log): 501 -> 595 TFLOP/s of the
// algorithmic 10 B H Lq Lkv dh at dh 128 (4608^2), 395-420 -> 511-551 at dh 64.
// Tried and removed (commit "Attention backward: hand-pipelined LDS reads ...", same log): an X|Y staggered variant as in the forward (wave
// groups one segment apart, S of the whole tile and Z crossing the barriers in registers, three LDS-DMA buffers). Same bits, but at dh 128 it
// needs ~300 registers (hipcc spilled 100: 211 TFLOP/s) and at dh 64 it measured 431-465 against the lock-step kernel's 453-492 of that day:
// stamps showed each group's segment stretching by 450-900 cycles beside its partner's although a VALU-only and an MFMA-only wave co-execute
// perfectly in isolation (tools/probe/coexec*.hip) - unexplained, left for a later round. Starting waves 4-7 one matrix phase late per tile
// inside the lock-step kernel: +3 % / -6 % (dh 128 / 64) before the pipelining, -3 % after it.

// =====================================================================================================================
enum { BWD_LSE = 0, BWD_DQ = 1, BWD_DK = 2, BWD_DV = 3 };

template <int DH, int MODE, bool DMA>
__global__ __launch_bounds__(512, 2) void attn_bwd_kernel(
    const bf16_t* __restrict__ own1, int64_t o1_rs, int64_t o1_bs, const bf16_t* __restrict__ own2, int64_t o2_rs, int64_t o2_bs,
    const bf16_t* __restrict__ st1, int64_t s1_rs, int64_t s1_bs, const bf16_t* __restrict__ st2, int64_t s2_rs, int64_t s2_bs,
    float* __restrict__ lse2, const float* __restrict__ delta, int64_t stat_ld /* queries per (b, h) row of lse2 / delta */,
    bf16_t* __restrict__ out, int64_t out_rs, int64_t out_bs, int heads, int Lown, int Lst, int nOwn, float c, float scale) {
    constexpr int RB = 2 * DH, NCH = DH / 8, TILE = KVB * RB, QS = DH / 16, NDB = DH / 32, NT = 512, NST = (KVB * NCH) / NT;
    constexpr bool OWN_Q = MODE == BWD_LSE || MODE == BWD_DQ;          // queries owned (statistics lane-local) or streamed
    constexpr bool TWO = MODE == BWD_DQ || MODE == BWD_DK;             // second score-like product (dP)
    constexpr int BUFSZ = 2 * TILE + (DMA ? 512 : 0);                      // DMA: + lse2[64] | delta[64] of the streamed rows (queries streamed)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // [2][tile of st1 | tile of st2 (| statistics)]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int ot = blockIdx.x % nOwn, bh = blockIdx.x / nOwn;
    const int head = bh % heads, b = bh / heads;
    const bf16_t* O1 = own1 + (int64_t)b * o1_bs + head * DH;
    const bf16_t* O2 = TWO ? own2 + (int64_t)b * o2_bs + head * DH : nullptr;
    const bf16_t* S1 = st1 + (int64_t)b * s1_bs + head * DH;
    const bf16_t* S2 = (MODE != BWD_LSE) ? st2 + (int64_t)b * s2_bs + head * DH : nullptr;
    const int own_row = ot * 256 + wave * 32 + r;
    const int own_ld = own_row < Lown ? own_row : Lown - 1;
    bf16x8 f1[QS], f2[TWO ? QS : 1];
#pragma unroll
    for (int s = 0; s < QS; ++s) {
        f1[s] = *(const bf16x8*)(O1 + (int64_t)own_ld * o1_rs + 16 * s + 8 * h);
        if constexpr (TWO) f2[s] = *(const bf16x8*)(O2 + (int64_t)own_ld * o2_rs + 16 * s + 8 * h);
    }
    const float* stat_l = lse2 + (int64_t)bh * stat_ld;
    const float* stat_d = (MODE == BWD_DQ || MODE == BWD_DK) ? delta + (int64_t)bh * stat_ld : nullptr;
    float my_lse = 0.f, my_delta = 0.f;
    if constexpr (MODE == BWD_DQ) { my_lse = stat_l[own_ld]; my_delta = stat_d[own_ld]; }
    // staging: thread -> NST chunks of each streamed tile
    int st_row[NST], st_ch[NST], st_off[NST];
#pragma unroll
    for (int u = 0; u < NST; ++u) {
        const int cid = tid + NT * u;
        st_row[u] = cid / NCH; st_ch[u] = cid % NCH;
        st_off[u] = img_off<DH>(st_row[u], st_ch[u]);
    }
    u32x4 r1[DMA ? 1 : NST], r2[DMA ? 1 : NST];
    // LDS-DMA staging (DMA): a tile image is NI runs of 1 KiB (RPI rows each), wave w owns runs w * NIW .. + NIW - 1 of both streamed tiles; the
    // DMAs of tile t + 1 are issued at the top of tile t and waited for (vmcnt(0)) ahead of the barrier that ends it - no staging registers,
    // no ds_write. The swizzle is applied on the source side as in the forward.
    constexpr int RPI = 1024 / RB, NI = TILE / 1024, NIW = NI / 8;
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    unsigned d1o[NIW], d2o[NIW];
#pragma unroll
    for (int u = 0; u < NIW; ++u) {
        const int row = (wv * NIW + u) * RPI + lane / NCH;
        const int ch = (lane % NCH) ^ row_swz<DH>(row);
        d1o[u] = (unsigned)(row * (int)s1_rs + ch * 8) * 2u;
        d2o[u] = (unsigned)(row * (int)s2_rs + ch * 8) * 2u;
    }
    auto dma_stream = [&](const bf16_t* base, int64_t rs, const unsigned (&off)[NIW], int row0, unsigned dst) {
        if (row0 + KVB <= Lst) {
            const void* tb = uniform_ptr(base + (int64_t)row0 * rs);
#pragma unroll
            for (int u = 0; u < NIW; ++u) glds16_off(tb, off[u], dst + u * 1024);
        } else {                                       // ragged last tile: rows past the end re-read the last row (masked below)
            int lane_r = lane;
            asm volatile("" : "+v"(lane_r));
#pragma unroll
            for (int u = 0; u < NIW; ++u) {
                const int row = (wv * NIW + u) * RPI + lane_r / NCH;
                const int ch = (lane_r % NCH) ^ row_swz<DH>(row);
                int sr = row0 + row; if (sr > Lst - 1) sr = Lst - 1;
                glds16_ptr(base + (int64_t)sr * rs + ch * 8, dst + u * 1024);
            }
        }
    };
    auto stage_load = [&](int row0, int buf) {
        if constexpr (DMA) {
            const unsigned lb = __builtin_amdgcn_readfirstlane(lds_addr(smem)) + buf * BUFSZ, l0 = lb + wv * NIW * 1024;
            dma_stream(S1, s1_rs, d1o, row0, l0);
            if constexpr (MODE != BWD_LSE) dma_stream(S2, s2_rs, d2o, row0, l0 + TILE);
            if constexpr (!OWN_Q) {                    // the tile's 64 lse2 / delta values ride along (rows padded to a multiple of 64, zeros)
                if (wv == 0) glds4_ptr(stat_l + row0 + lane, lb + 2 * TILE);
                if (MODE == BWD_DK && wv == 1) glds4_ptr(stat_d + row0 + lane, lb + 2 * TILE + 256);
            }
        } else {
#pragma unroll
            for (int u = 0; u < NST; ++u) {
                int row = row0 + st_row[u]; if (row > Lst - 1) row = Lst - 1;
                r1[u] = *(const u32x4*)(S1 + (int64_t)row * s1_rs + st_ch[u] * 8);
                if constexpr (MODE != BWD_LSE) r2[u] = *(const u32x4*)(S2 + (int64_t)row * s2_rs + st_ch[u] * 8);
            }
        }
    };
    auto stage_write = [&](int buf) {
        if constexpr (DMA) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
#pragma unroll
            for (int u = 0; u < NST; ++u) {
                *(u32x4*)(smem + buf * BUFSZ + st_off[u]) = r1[u];
                if constexpr (MODE != BWD_LSE) *(u32x4*)(smem + buf * BUFSZ + TILE + st_off[u]) = r2[u];
            }
        }
    };
    // per-lane LDS read offsets: every swizzled offset is BASE ^ constant, re-derived from opaque copies of three bases (no per-fragment registers)
    const int k_base = RB * r + 16 * (h ^ row_swz<DH>(r));                 // row fragment s of 32-row block kb: kb * 32 * RB + (k_base ^ 32 s)
    const int i16 = lane & 15, g16 = lane >> 4;
    const int t_key = 4 * h + (i16 >> 2), t_lowch = 2 * (g16 & 1) + ((i16 & 3) >> 1), t_b8 = 8 * (i16 & 1);
    const int tlo_base = RB * t_key + 16 * (t_lowch ^ row_swz<DH>(t_key)) + t_b8;              // d-block db, k-step ks: ks * 16 * RB + (base ^ 64 db)
    const int thi_base = RB * (t_key + 8) + 16 * (t_lowch ^ row_swz<DH>(t_key + 8)) + t_b8;
    // dh 128: the LDS reads of each matrix phase are software-pipelined by hand - the fragments of step j + PD are requested before the MFMAs
    // of step j (left to hipcc every MFMA pair sits right behind its own ds_read and s_waitcnt): 527 -> 548 TFLOP/s. At dh 64 (half the MFMAs
    // per read burst) the same pipeline measured 6 % slower than hipcc's order, which stays.
    constexpr bool PIPE = DH == 128;
    constexpr int PDS = 3, PDA = 1;
    f32x16 acc[MODE == BWD_LSE ? 1 : NDB];
    if constexpr (MODE != BWD_LSE) {
#pragma unroll
        for (int db = 0; db < NDB; ++db)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[db][i] = 0.f;
    }
    float m_run = -INFINITY, l_run = 0.f;
    const int ntiles = (Lst + KVB - 1) / KVB;
    stage_load(0, 0);
    stage_write(0);
    __syncthreads();
    for (int t = 0; t < ntiles; ++t) {
        const int cur = t & 1;
        if (t + 1 < ntiles) stage_load((t + 1) * KVB, cur ^ 1);
        const unsigned char* B1 = smem + cur * BUFSZ;
        const unsigned char* B2 = B1 + TILE;
        bf16x8 zf[2][2];
        float tmax = -INFINITY;
        f32x16 x1k[MODE == BWD_LSE ? 2 : 1];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            f32x16 x1, x2;
#pragma unroll
            for (int i = 0; i < 16; ++i) { x1[i] = 0.f; x2[i] = 0.f; }
            if constexpr (PIPE) {
                int kb0 = k_base;
                asm volatile("" : "+v"(kb0));
                bf16x8 ab[PDS + 1][2];
                auto rd = [&](int s5) {
                    ab[s5 % (PDS + 1)][0] = *(const bf16x8*)(B1 + kb * 32 * RB + (kb0 ^ (32 * s5)));
                    if constexpr (TWO) ab[s5 % (PDS + 1)][1] = *(const bf16x8*)(B2 + kb * 32 * RB + (kb0 ^ (32 * s5)));
                };
#pragma unroll
                for (int j = 0; j < PDS && j < QS; ++j) rd(j);
#pragma unroll
                for (int s5 = 0; s5 < QS; ++s5) {
                    if (s5 + PDS < QS) rd(s5 + PDS);
                    __builtin_amdgcn_sched_barrier(0);
                    x1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab[s5 % (PDS + 1)][0], f1[s5], x1, 0, 0, 0);
                    if constexpr (TWO) x2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab[s5 % (PDS + 1)][1], f2[s5], x2, 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {
#pragma unroll
                for (int s5 = 0; s5 < QS; ++s5) {
                    const bf16x8 a1 = *(const bf16x8*)(B1 + kb * 32 * RB + (k_base ^ (32 * s5)));
                    x1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, f1[s5], x1, 0, 0, 0);
                    if constexpr (TWO) {
                        const bf16x8 a2 = *(const bf16x8*)(B2 + kb * 32 * RB + (k_base ^ (32 * s5)));
                        x2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, f2[s5], x2, 0, 0, 0);
                    }
                }
            }
            // streamed row of accumulator element i
            const int srow0 = t * KVB + kb * 32 + 4 * h;
            if constexpr (MODE == BWD_LSE) {
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    if (srow0 + (i & 3) + 8 * (i >> 2) >= Lst) x1[i] = -INFINITY;
                    tmax = fmaxf(tmax, x1[i]);
                }
                x1k[kb] = x1;
            } else {
                float z[16];
                float sl[16], sd[16];
                if constexpr (!OWN_Q && DMA) {         // statistics of the streamed rows from the LDS copy (broadcast reads), no global load in the loop
                    const float* st = (const float*)(B1 + 2 * TILE);
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const f32x4 a = *(const f32x4*)(st + kb * 32 + 4 * h + 8 * g);
                        sl[4 * g] = a[0]; sl[4 * g + 1] = a[1]; sl[4 * g + 2] = a[2]; sl[4 * g + 3] = a[3];
                        if constexpr (MODE == BWD_DK) {
                            const f32x4 d4 = *(const f32x4*)(st + 64 + kb * 32 + 4 * h + 8 * g);
                            sd[4 * g] = d4[0]; sd[4 * g + 1] = d4[1]; sd[4 * g + 2] = d4[2]; sd[4 * g + 3] = d4[3];
                        }
                    }
                }
                if constexpr (!OWN_Q && !DMA) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        int q0 = srow0 + 8 * g; if (q0 > (int)stat_ld - 4) q0 = (int)stat_ld - 4;       // stat rows are padded to a multiple of 64
                        const f32x4 a = *(const f32x4*)(stat_l + q0);
                        sl[4 * g] = a[0]; sl[4 * g + 1] = a[1]; sl[4 * g + 2] = a[2]; sl[4 * g + 3] = a[3];
                        if constexpr (MODE == BWD_DK) {
                            const f32x4 d4 = *(const f32x4*)(stat_d + q0);
                            sd[4 * g] = d4[0]; sd[4 * g + 1] = d4[1]; sd[4 * g + 2] = d4[2]; sd[4 * g + 3] = d4[3];
                        }
                    }
                }
                // z = P (DV) or P (dP - delta) (DQ, DK: the softmax scale is applied once, to the accumulator, in the epilogue); rows past the
                // end exist only in the ragged last tile
                if (t * KVB + KVB <= Lst) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const float p = __builtin_amdgcn_exp2f(fmaf(x1[i], c, -(OWN_Q ? my_lse : sl[i])));
                        z[i] = TWO ? p * (x2[i] - (OWN_Q ? my_delta : sd[i])) : p;
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const bool valid = srow0 + (i & 3) + 8 * (i >> 2) < Lst;
                        const float p = __builtin_amdgcn_exp2f(fmaf(x1[i], c, -(OWN_Q ? my_lse : sl[i])));
                        const float v = TWO ? p * (x2[i] - (OWN_Q ? my_delta : sd[i])) : p;
                        z[i] = valid ? v : 0.f;
                    }
                }
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    u32x4 w;
                    w.x = pack2bf(z[8 * s2 + 0], z[8 * s2 + 1]); w.y = pack2bf(z[8 * s2 + 2], z[8 * s2 + 3]);
                    w.z = pack2bf(z[8 * s2 + 4], z[8 * s2 + 5]); w.w = pack2bf(z[8 * s2 + 6], z[8 * s2 + 7]);
                    zf[kb][s2] = __builtin_bit_cast(bf16x8, w);
                }
            }
        }
        if constexpr (MODE == BWD_LSE) {
            tmax = ug_max_halves(tmax);
            const float m_new = fmaxf(m_run, tmax);
            float sum = 0.f;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int i = 0; i < 16; ++i) sum += __builtin_amdgcn_exp2f((x1k[kb][i] - m_new) * c);
            l_run = l_run * __builtin_amdgcn_exp2f((m_run - m_new) * c) + sum;
            m_run = m_new;
        } else {
            // acc^T[d][own] += T^T[d][streamed] Z[streamed][own], T = st1 (DQ: K, DK: Q) or st2 (DV: dO)
            const unsigned char* Tb = (MODE == BWD_DV) ? B2 : B1;
            if constexpr (PIPE) {
                int lo0 = tlo_base, hi0 = thi_base;
                asm volatile("" : "+v"(lo0), "+v"(hi0));
                bf16x8 tfb[PDA + 1][NDB];
                auto rd = [&](int ks) {
#pragma unroll
                    for (int db = 0; db < NDB; ++db) tfb[ks % (PDA + 1)][db] = tr_read_pair(Tb + ks * 16 * RB + (lo0 ^ (64 * db)), Tb + ks * 16 * RB + (hi0 ^ (64 * db)));
                };
#pragma unroll
                for (int j = 0; j < PDA && j < 4; ++j) rd(j);
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    if (ks + PDA < 4) rd(ks + PDA);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int db = 0; db < NDB; ++db) acc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tfb[ks % (PDA + 1)][db], zf[ks >> 1][ks & 1], acc[db], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {
#pragma unroll
                for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                    for (int db = 0; db < NDB; ++db) {
                        const bf16x8 tf = tr_read_pair(Tb + ks * 16 * RB + (tlo_base ^ (64 * db)), Tb + ks * 16 * RB + (thi_base ^ (64 * db)));
                        acc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tf, zf[ks >> 1][ks & 1], acc[db], 0, 0, 0);
                    }
            }
        }
        if (t + 1 < ntiles) stage_write(cur ^ 1);
        __syncthreads();
    }
    if constexpr (MODE == BWD_LSE) {
        const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
        if (own_row < Lown && h == 0) lse2[(int64_t)bh * stat_ld + own_row] = __builtin_amdgcn_logf(l_tot) + m_run * c;     // v_log_f32 = log2
    } else if (own_row < Lown) {
        bf16_t* Orow = out + (int64_t)b * out_bs + (int64_t)own_row * out_rs + head * DH;
#pragma unroll
        for (int db = 0; db < NDB; ++db)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const float es = TWO ? scale : 1.f;
                u32x2 w;
                w.x = pack2bf(acc[db][4 * g4 + 0] * es, acc[db][4 * g4 + 1] * es);
                w.y = pack2bf(acc[db][4 * g4 + 2] * es, acc[db][4 * g4 + 3] * es);
                *(u32x2*)(Orow + 32 * db + 8 * g4 + 4 * h) = w;
            }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Fused dK / dV (round 3): the DK and DV modes above both recompute S = Q K^T over the same (keys, queries) - 8 product units per attention
// against the minimum of 5. Fusing them on the 256-key ownership needs 128 accumulator + 64 owned-operand registers beside the score tiles:
// over the 256 a two-wave-per-SIMD kernel has. This kernel splits the work of a 32-key block between the TWO waves of a pair instead:
//   a workgroup owns 128 keys; waves p and p + 4 (p = 0..3) own the same 32 keys (K and V fragments in registers, key on the lane);
//   score phase : wave half h = wave >> 2 takes the 32 streamed queries kb = h of the 64-row tile: S, dP -> P and dS = P (dP - delta), packed to bf16 -
//                 the B operands of the accumulating products. Half 0 accumulates dK, half 1 accumulates dV: a wave keeps the operand of ITS gradient
//                 and publishes the other one lane-linear in LDS (2 KiB per wave, double-buffered);
//   accumulation: after ONE barrier per tile a wave has its partner's operand too and accumulates its gradient over all 64 queries, all head dims:
//                 dK^T += Q^T dS (half 0) or dV^T += dO^T P (half 1) (transposed reads of the streamed tile, fragments one k-step ahead).
// Per 32 keys x 64 queries: 16 + 16 (scores) + 16 + 16 (accumulation) MFMAs = the five product units, no duplicated product, 64 accumulator
// registers per wave. The streamed tiles (image and statistics of the DK mode) sit in a ring of THREE stages: tile t + 2 is requested right after the
// barrier of tile t and waited for ahead of the barrier of tile t + 1, so that single barrier certifies the exchange, the landing and the free stage.
// Measured (tools/attn_bwd_ab.py, profiles/r03y_attn_bwd_fuse_*.log; whole backward incl. the DQ mode): dh 128 605 -> 711 TFLOP/s of the algorithmic
// 10 B H Lq Lkv dh at 4608^2 (+17.6 %; 8704^2 +17 %, 1000^2 +8 %), dh 64 531 -> 578 / 574 -> 612; same bits as the two modes (same products, same order).
// Steps on the way: the first form (both gradients per wave on half the head dims, 4 KiB exchange, two barriers) +0.8 %; its transposed reads one step
// ahead +2 %; the per-element row masks out of the whole-tile path (32 v_cndmask per tile: the score phase is VALU-bound) +13.6 %; this form +17.6 %.
// Tried and dropped: the dV waves scoring tile t + 1 BEFORE accumulating tile t while the dK waves do the opposite (the two waves of a SIMD then
// alternate VALU-heavy and MFMA-only phases): 37 spilled registers at dh 128 (421 TFLOP/s), and 522 vs 578 at dh 64 without a single spill.
// Also dropped: scalar branches instead of the wave-uniform `half ? a : b` selects of the operands (24 v_cndmask per tile): the duplicated
// accumulation block costs 46 spilled registers at dh 128 (476 TFLOP/s).
// ---------------------------------------------------------------------------------------------------------------------
template <int DH>
__global__ __launch_bounds__(512, 2) void attn_bwd_dkv_kernel(
    const bf16_t* __restrict__ kk, int64_t k_rs, int64_t k_bs, const bf16_t* __restrict__ vv, int64_t v_rs, int64_t v_bs,
    const bf16_t* __restrict__ qq, int64_t q_rs, int64_t q_bs, const bf16_t* __restrict__ dd, int64_t d_rs, int64_t d_bs,
    const float* __restrict__ lse2, const float* __restrict__ delta, int64_t stat_ld, bf16_t* __restrict__ dk, int64_t dk_rs, int64_t dk_bs,
    bf16_t* __restrict__ dv, int64_t dv_rs, int64_t dv_bs, int heads, int Lkv, int Lq, int nOwn, float c, float scale) {
    constexpr int RB = 2 * DH, NCH = DH / 8, TILE = KVB * RB, QS = DH / 16, NDB = DH / 32;
    constexpr int BUFSZ = 2 * TILE + 512, NSTG = 3, ZOFF = NSTG * BUFSZ;   // [3][Q tile | dO tile | lse2[64] | delta[64]] then [2][8 waves] x 2 KiB of exchanged operands
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pair = wave & 3, half = wave >> 2;           // half 0: queries kb = 0 of a tile, accumulates dK; half 1: queries kb = 1, accumulates dV
    const int r = lane & 31, h = lane >> 5;
    const int ot = blockIdx.x % nOwn, bh = blockIdx.x / nOwn;
    const int head = bh % heads, b = bh / heads;
    const bf16_t* Kb = kk + (int64_t)b * k_bs + head * DH;
    const bf16_t* Vb = vv + (int64_t)b * v_bs + head * DH;
    const bf16_t* Qb = qq + (int64_t)b * q_bs + head * DH;
    const bf16_t* Db = dd + (int64_t)b * d_bs + head * DH;
    const int own_row = ot * 128 + pair * 32 + r;
    const int own_ld = own_row < Lkv ? own_row : Lkv - 1;
    bf16x8 fk[QS], fv[QS];
#pragma unroll
    for (int s = 0; s < QS; ++s) {
        fk[s] = *(const bf16x8*)(Kb + (int64_t)own_ld * k_rs + 16 * s + 8 * h);
        fv[s] = *(const bf16x8*)(Vb + (int64_t)own_ld * v_rs + 16 * s + 8 * h);
    }
    const float* stat_l = lse2 + (int64_t)bh * stat_ld;
    const float* stat_d = delta + (int64_t)bh * stat_ld;
    // LDS-DMA staging as in attn_bwd_kernel (wave w owns runs w * NIW .. + NIW - 1 of both streamed tiles), but a ring of three stages: tile t + 2 is
    // requested right after the barrier of tile t (its stage was last read by the accumulation of tile t - 1, which every wave has left by then) and is
    // waited for ahead of the barrier of tile t + 1 - one barrier per tile certifies both the exchange and the landing.
    constexpr int RPI = 1024 / RB, NI = TILE / 1024, NIW = NI / 8;
    unsigned d1o[NIW], d2o[NIW];
#pragma unroll
    for (int u = 0; u < NIW; ++u) {
        const int row = (wave * NIW + u) * RPI + lane / NCH;
        const int ch = (lane % NCH) ^ row_swz<DH>(row);
        d1o[u] = (unsigned)(row * (int)q_rs + ch * 8) * 2u;
        d2o[u] = (unsigned)(row * (int)d_rs + ch * 8) * 2u;
    }
    auto dma_stream = [&](const bf16_t* base, int64_t rs, const unsigned (&off)[NIW], int row0, unsigned dst) __attribute__((always_inline)) {
        if (row0 + KVB <= Lq) {
            const void* tb = uniform_ptr(base + (int64_t)row0 * rs);
#pragma unroll
            for (int u = 0; u < NIW; ++u) glds16_off(tb, off[u], dst + u * 1024);
        } else {                                       // ragged last tile: rows past the end re-read the last row (masked below)
            int lane_r = lane;
            asm volatile("" : "+v"(lane_r));
#pragma unroll
            for (int u = 0; u < NIW; ++u) {
                const int row = (wave * NIW + u) * RPI + lane_r / NCH;
                const int ch = (lane_r % NCH) ^ row_swz<DH>(row);
                int sr = row0 + row; if (sr > Lq - 1) sr = Lq - 1;
                glds16_ptr(base + (int64_t)sr * rs + ch * 8, dst + u * 1024);
            }
        }
    };
    auto stage_load = [&](int row0, int stg) __attribute__((always_inline)) {
        const unsigned lb = __builtin_amdgcn_readfirstlane(lds_addr(smem)) + stg * BUFSZ, l0 = lb + wave * NIW * 1024;
        dma_stream(Qb, q_rs, d1o, row0, l0);
        dma_stream(Db, d_rs, d2o, row0, l0 + TILE);
        if (wave == 0) glds4_ptr(stat_l + row0 + lane, lb + 2 * TILE);          // the tile's 64 lse2 / delta values (rows padded to a multiple of 64, zeros)
        if (wave == 1) glds4_ptr(stat_d + row0 + lane, lb + 2 * TILE + 256);
    };
    const int k_base = RB * r + 16 * (h ^ row_swz<DH>(r));                 // row fragment s of 32-row block kb: kb * 32 * RB + (k_base ^ 32 s)
    const int i16 = lane & 15, g16 = lane >> 4;
    const int t_key = 4 * h + (i16 >> 2), t_lowch = 2 * (g16 & 1) + ((i16 & 3) >> 1), t_b8 = 8 * (i16 & 1);
    const int tlo_base = RB * t_key + 16 * (t_lowch ^ row_swz<DH>(t_key)) + t_b8;              // d-block db, k-step ks: ks * 16 * RB + (base ^ 64 db)
    const int thi_base = RB * (t_key + 8) + 16 * (t_lowch ^ row_swz<DH>(t_key + 8)) + t_b8;
    f32x16 acc[NDB];                                   // dK^T (half 0) or dV^T (half 1): [d][own key]
#pragma unroll
    for (int db = 0; db < NDB; ++db)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[db][i] = 0.f;
    const int ntiles = (Lq + KVB - 1) / KVB;
    stage_load(0, 0);
    if (ntiles > 1) stage_load(KVB, 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int stg = 0;                                       // stage of tile t
    for (int t = 0; t < ntiles; ++t) {
        const unsigned char* B1 = smem + stg * BUFSZ;
        const unsigned char* B2 = B1 + TILE;
        unsigned char* const zmine = smem + ZOFF + (t & 1) * 16384 + wave * 2048 + lane * 16;
        const unsigned char* const zpart = smem + ZOFF + (t & 1) * 16384 + (wave ^ 4) * 2048 + lane * 16;
        // ---- score phase: the 32 streamed queries kb = half; this wave keeps the operand of ITS gradient and publishes the other one ----
        bf16x8 zown[2];
        {
            f32x16 x1, x2;
#pragma unroll
            for (int i = 0; i < 16; ++i) { x1[i] = 0.f; x2[i] = 0.f; }
            int kb0 = k_base;
            asm volatile("" : "+v"(kb0));
            constexpr int PDS = 2;
            bf16x8 ab[PDS + 1][2];
            auto rd = [&](int s5) __attribute__((always_inline)) {
                ab[s5 % (PDS + 1)][0] = *(const bf16x8*)(B1 + half * 32 * RB + (kb0 ^ (32 * s5)));
                ab[s5 % (PDS + 1)][1] = *(const bf16x8*)(B2 + half * 32 * RB + (kb0 ^ (32 * s5)));
            };
#pragma unroll
            for (int j = 0; j < PDS && j < QS; ++j) rd(j);
#pragma unroll
            for (int s5 = 0; s5 < QS; ++s5) {
                if (s5 + PDS < QS) rd(s5 + PDS);
                __builtin_amdgcn_sched_barrier(0);
                x1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab[s5 % (PDS + 1)][0], fk[s5], x1, 0, 0, 0);
                x2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab[s5 % (PDS + 1)][1], fv[s5], x2, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            const float* st = (const float*)(B1 + 2 * TILE);
            float sl[16], sd[16];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 a = *(const f32x4*)(st + half * 32 + 4 * h + 8 * g);
                const f32x4 d4 = *(const f32x4*)(st + 64 + half * 32 + 4 * h + 8 * g);
                sl[4 * g] = a[0]; sl[4 * g + 1] = a[1]; sl[4 * g + 2] = a[2]; sl[4 * g + 3] = a[3];
                sd[4 * g] = d4[0]; sd[4 * g + 1] = d4[1]; sd[4 * g + 2] = d4[2]; sd[4 * g + 3] = d4[3];
            }
            float pk[16], pv[16];
            if (t * KVB + KVB <= Lq) {                 // rows past the end exist only in the ragged last tile
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    pv[i] = __builtin_amdgcn_exp2f(fmaf(x1[i], c, -sl[i]));
                    pk[i] = pv[i] * (x2[i] - sd[i]);
                }
            } else {
                const int srow0 = t * KVB + half * 32 + 4 * h;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const float p = __builtin_amdgcn_exp2f(fmaf(x1[i], c, -sl[i]));
                    const bool valid = srow0 + (i & 3) + 8 * (i >> 2) < Lq;
                    pv[i] = valid ? p : 0.f;
                    pk[i] = valid ? p * (x2[i] - sd[i]) : 0.f;
                }
            }
            bf16x8 zk[2], zv[2];
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                u32x4 w;
                w.x = pack2bf(pk[8 * s2 + 0], pk[8 * s2 + 1]); w.y = pack2bf(pk[8 * s2 + 2], pk[8 * s2 + 3]);
                w.z = pack2bf(pk[8 * s2 + 4], pk[8 * s2 + 5]); w.w = pack2bf(pk[8 * s2 + 6], pk[8 * s2 + 7]);
                zk[s2] = __builtin_bit_cast(bf16x8, w);
                w.x = pack2bf(pv[8 * s2 + 0], pv[8 * s2 + 1]); w.y = pack2bf(pv[8 * s2 + 2], pv[8 * s2 + 3]);
                w.z = pack2bf(pv[8 * s2 + 4], pv[8 * s2 + 5]); w.w = pack2bf(pv[8 * s2 + 6], pv[8 * s2 + 7]);
                zv[s2] = __builtin_bit_cast(bf16x8, w);
            }
            zown[0] = half ? zv[0] : zk[0]; zown[1] = half ? zv[1] : zk[1];        // wave-uniform selects
            *(bf16x8*)(zmine) = half ? zk[0] : zv[0];
            *(bf16x8*)(zmine + 1024) = half ? zk[1] : zv[1];
        }
        // ---- accumulation over all 64 queries: dK^T += Q^T dS (half 0) or dV^T += dO^T P (half 1); the transposed fragments of k-step ks + 1 are
        //      requested before the MFMAs of step ks, those of step 0 ahead of the barrier (they do not depend on the partner) ----
        {
            const unsigned char* Tb = half ? B2 : B1;
            int lo0 = tlo_base, hi0 = thi_base;
            asm volatile("" : "+v"(lo0), "+v"(hi0));
            bf16x8 tf[2][NDB];
            auto rdT = [&](int ks) __attribute__((always_inline)) {
#pragma unroll
                for (int db = 0; db < NDB; ++db) tf[ks & 1][db] = tr_read_pair(Tb + ks * 16 * RB + (lo0 ^ (64 * db)), Tb + ks * 16 * RB + (hi0 ^ (64 * db)));
            };
            rdT(0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this wave's share of tile t + 1 has landed (requested a whole tile ago)
            __builtin_amdgcn_sched_barrier(0);
            __syncthreads();                           // the partner's operand is in LDS; tile t + 1 is complete; nobody reads the stage of tile t - 1 any more
            if (t + 2 < ntiles) stage_load((t + 2) * KVB, stg == 0 ? 2 : stg - 1);
            const bf16x8 zp0 = *(const bf16x8*)(zpart), zp1 = *(const bf16x8*)(zpart + 1024);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const bool mine = (ks >> 1) == half;               // wave-uniform
                const bf16x8 bz = mine ? zown[ks & 1] : ((ks & 1) ? zp1 : zp0);
                if (ks + 1 < 4) rdT(ks + 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int db = 0; db < NDB; ++db) acc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tf[ks & 1][db], bz, acc[db], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        stg = stg == 2 ? 0 : stg + 1;
    }
    if (own_row < Lkv) {
        bf16_t* Orow = (half ? dv + (int64_t)b * dv_bs + (int64_t)own_row * dv_rs : dk + (int64_t)b * dk_bs + (int64_t)own_row * dk_rs) + head * DH;
        const float es = half ? 1.f : scale;
#pragma unroll
        for (int db = 0; db < NDB; ++db)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                u32x2 w;
                w.x = pack2bf(acc[db][4 * g4 + 0] * es, acc[db][4 * g4 + 1] * es);
                w.y = pack2bf(acc[db][4 * g4 + 2] * es, acc[db][4 * g4 + 3] * es);
                *(u32x2*)(Orow + 32 * db + 8 * g4 + 4 * h) = w;
            }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// dQ on the pair scheme (round 3): the DQ mode above owns 256 queries per workgroup - 18 x 24 x B = 864 workgroups at L = 4608, B = 2: 3.4 rounds of
// the 256 CUs, the last one 37 % full. Here a workgroup owns 128 queries; waves p and p + 4 own the same 32 (Q and dO fragments in registers, query on
// the lane), wave half h takes the 32 streamed KEYS kb = h of each 64-key tile: S^T, dP^T -> dS^T -> dQ^T += K^T dS^T over its own keys only - no
// exchange per tile, 24 MFMAs per wave and tile - and the two partial sums of a pair meet once, in LDS, after the last tile. Twice the workgroups
// (6.75 rounds: 4 % idle in the last instead of 16 %), half the accumulator-side registers, the three-stage stream ring and one barrier per tile of
// the fused dK / dV kernel. The sum over keys associates differently from the DQ mode (two partial sums): same value to fp32 rounding, not the same bits.
// ---------------------------------------------------------------------------------------------------------------------
template <int DH>
__global__ __launch_bounds__(512, 2) void attn_bwd_dq_kernel(
    const bf16_t* __restrict__ qq, int64_t q_rs, int64_t q_bs, const bf16_t* __restrict__ dd, int64_t d_rs, int64_t d_bs,
    const bf16_t* __restrict__ kk, int64_t k_rs, int64_t k_bs, const bf16_t* __restrict__ vv, int64_t v_rs, int64_t v_bs,
    const float* __restrict__ lse2, const float* __restrict__ delta, int64_t stat_ld, bf16_t* __restrict__ dq, int64_t dq_rs, int64_t dq_bs,
    int heads, int Lq, int Lkv, int nOwn, float c, float scale) {
    constexpr int RB = 2 * DH, NCH = DH / 8, TILE = KVB * RB, QS = DH / 16, NDB = DH / 32;
    constexpr int BUFSZ = 2 * TILE;                    // [3][K tile | V tile]
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pair = wave & 3, half = wave >> 2;
    const int r = lane & 31, h = lane >> 5;
    const int ot = blockIdx.x % nOwn, bh = blockIdx.x / nOwn;
    const int head = bh % heads, b = bh / heads;
    const bf16_t* Qb = qq + (int64_t)b * q_bs + head * DH;
    const bf16_t* Db = dd + (int64_t)b * d_bs + head * DH;
    const bf16_t* Kb = kk + (int64_t)b * k_bs + head * DH;
    const bf16_t* Vb = vv + (int64_t)b * v_bs + head * DH;
    const int own_row = ot * 128 + pair * 32 + r;
    const int own_ld = own_row < Lq ? own_row : Lq - 1;
    bf16x8 fq[QS], fo[QS];
#pragma unroll
    for (int s = 0; s < QS; ++s) {
        fq[s] = *(const bf16x8*)(Qb + (int64_t)own_ld * q_rs + 16 * s + 8 * h);
        fo[s] = *(const bf16x8*)(Db + (int64_t)own_ld * d_rs + 16 * s + 8 * h);
    }
    const float my_lse = lse2[(int64_t)bh * stat_ld + own_ld], my_delta = delta[(int64_t)bh * stat_ld + own_ld];
    constexpr int RPI = 1024 / RB, NI = TILE / 1024, NIW = NI / 8;
    unsigned d1o[NIW], d2o[NIW];
#pragma unroll
    for (int u = 0; u < NIW; ++u) {
        const int row = (wave * NIW + u) * RPI + lane / NCH;
        const int ch = (lane % NCH) ^ row_swz<DH>(row);
        d1o[u] = (unsigned)(row * (int)k_rs + ch * 8) * 2u;
        d2o[u] = (unsigned)(row * (int)v_rs + ch * 8) * 2u;
    }
    auto dma_stream = [&](const bf16_t* base, int64_t rs, const unsigned (&off)[NIW], int row0, unsigned dst) __attribute__((always_inline)) {
        if (row0 + KVB <= Lkv) {
            const void* tb = uniform_ptr(base + (int64_t)row0 * rs);
#pragma unroll
            for (int u = 0; u < NIW; ++u) glds16_off(tb, off[u], dst + u * 1024);
        } else {                                       // ragged last tile: rows past the end re-read the last key (masked below)
            int lane_r = lane;
            asm volatile("" : "+v"(lane_r));
#pragma unroll
            for (int u = 0; u < NIW; ++u) {
                const int row = (wave * NIW + u) * RPI + lane_r / NCH;
                const int ch = (lane_r % NCH) ^ row_swz<DH>(row);
                int sr = row0 + row; if (sr > Lkv - 1) sr = Lkv - 1;
                glds16_ptr(base + (int64_t)sr * rs + ch * 8, dst + u * 1024);
            }
        }
    };
    auto stage_load = [&](int row0, int stg) __attribute__((always_inline)) {
        const unsigned l0 = __builtin_amdgcn_readfirstlane(lds_addr(smem)) + stg * BUFSZ + wave * NIW * 1024;
        dma_stream(Kb, k_rs, d1o, row0, l0);
        dma_stream(Vb, v_rs, d2o, row0, l0 + TILE);
    };
    const int k_base = RB * r + 16 * (h ^ row_swz<DH>(r));
    const int i16 = lane & 15, g16 = lane >> 4;
    const int t_key = 4 * h + (i16 >> 2), t_lowch = 2 * (g16 & 1) + ((i16 & 3) >> 1), t_b8 = 8 * (i16 & 1);
    const int tlo_base = RB * t_key + 16 * (t_lowch ^ row_swz<DH>(t_key)) + t_b8;
    const int thi_base = RB * (t_key + 8) + 16 * (t_lowch ^ row_swz<DH>(t_key + 8)) + t_b8;
    f32x16 acc[NDB];                                   // partial dQ^T [d][own query] over this wave's key blocks
#pragma unroll
    for (int db = 0; db < NDB; ++db)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[db][i] = 0.f;
    const int ntiles = (Lkv + KVB - 1) / KVB;
    stage_load(0, 0);
    if (ntiles > 1) stage_load(KVB, 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int stg = 0;
    for (int t = 0; t < ntiles; ++t) {
        const unsigned char* B1 = smem + stg * BUFSZ;
        const unsigned char* B2 = B1 + TILE;
        if (t + 2 < ntiles) stage_load((t + 2) * KVB, stg == 0 ? 2 : stg - 1);      // the stage of tile t - 1: every wave left it before the barrier behind us
        f32x16 x1, x2;
#pragma unroll
        for (int i = 0; i < 16; ++i) { x1[i] = 0.f; x2[i] = 0.f; }
        {
            int kb0 = k_base;
            asm volatile("" : "+v"(kb0));
            constexpr int PDS = 3;
            bf16x8 ab[PDS + 1][2];
            auto rd = [&](int s5) __attribute__((always_inline)) {
                ab[s5 % (PDS + 1)][0] = *(const bf16x8*)(B1 + half * 32 * RB + (kb0 ^ (32 * s5)));
                ab[s5 % (PDS + 1)][1] = *(const bf16x8*)(B2 + half * 32 * RB + (kb0 ^ (32 * s5)));
            };
#pragma unroll
            for (int j = 0; j < PDS && j < QS; ++j) rd(j);
#pragma unroll
            for (int s5 = 0; s5 < QS; ++s5) {
                if (s5 + PDS < QS) rd(s5 + PDS);
                __builtin_amdgcn_sched_barrier(0);
                x1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab[s5 % (PDS + 1)][0], fq[s5], x1, 0, 0, 0);
                x2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab[s5 % (PDS + 1)][1], fo[s5], x2, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // the transposed K fragments of this wave's two k-steps (keys 32 half + 16 ks ..): requested now, used after the softmax arithmetic
        bf16x8 tf[2][NDB];
        {
            int lo0 = tlo_base, hi0 = thi_base;
            asm volatile("" : "+v"(lo0), "+v"(hi0));
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int db = 0; db < NDB; ++db)
                    tf[ks][db] = tr_read_pair(B1 + (2 * half + ks) * 16 * RB + (lo0 ^ (64 * db)), B1 + (2 * half + ks) * 16 * RB + (hi0 ^ (64 * db)));
        }
        float z[16];
        if (t * KVB + KVB <= Lkv) {                    // keys past the end exist only in the ragged last tile
#pragma unroll
            for (int i = 0; i < 16; ++i) z[i] = __builtin_amdgcn_exp2f(fmaf(x1[i], c, -my_lse)) * (x2[i] - my_delta);
        } else {
            const int srow0 = t * KVB + half * 32 + 4 * h;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float v = __builtin_amdgcn_exp2f(fmaf(x1[i], c, -my_lse)) * (x2[i] - my_delta);
                z[i] = srow0 + (i & 3) + 8 * (i >> 2) < Lkv ? v : 0.f;
            }
        }
        bf16x8 zf[2];
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            u32x4 w;
            w.x = pack2bf(z[8 * s2 + 0], z[8 * s2 + 1]); w.y = pack2bf(z[8 * s2 + 2], z[8 * s2 + 3]);
            w.z = pack2bf(z[8 * s2 + 4], z[8 * s2 + 5]); w.w = pack2bf(z[8 * s2 + 6], z[8 * s2 + 7]);
            zf[s2] = __builtin_bit_cast(bf16x8, w);
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int db = 0; db < NDB; ++db) acc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tf[ks][db], zf[ks], acc[db], 0, 0, 0);
        // this wave's share of tile t + 1 (requested two tiles ago) has landed; the requests of tile t + 2, issued above, stay in flight
        if (t + 2 < ntiles) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NIW) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
        stg = stg == 2 ? 0 : stg + 1;
    }
    // the pair's two partial sums: half 1 parks its accumulators in LDS (lane-linear, NDB x 4 KiB per wave; the stream stages are free now), half 0 adds them
    float* park = (float*)smem + pair * (NDB * 4 * 256) + lane * 4;     // 4 x NDB x 4 KiB <= the three stream stages at either head width
    if (half == 1) {
#pragma unroll
        for (int db = 0; db < NDB; ++db)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4)
                *(f32x4*)(park + (db * 4 + g4) * 256) = (f32x4){acc[db][4 * g4 + 0], acc[db][4 * g4 + 1], acc[db][4 * g4 + 2], acc[db][4 * g4 + 3]};
    }
    __syncthreads();
    if (half == 0 && own_row < Lq) {
        bf16_t* Orow = dq + (int64_t)b * dq_bs + (int64_t)own_row * dq_rs + head * DH;
#pragma unroll
        for (int db = 0; db < NDB; ++db)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const f32x4 o = *(const f32x4*)(park + (db * 4 + g4) * 256);
                u32x2 w;
                w.x = pack2bf((acc[db][4 * g4 + 0] + o[0]) * scale, (acc[db][4 * g4 + 1] + o[1]) * scale);
                w.y = pack2bf((acc[db][4 * g4 + 2] + o[2]) * scale, (acc[db][4 * g4 + 3] + o[3]) * scale);
                *(u32x2*)(Orow + 32 * db + 8 * g4 + 4 * h) = w;
            }
    }
}

// delta[bh][q] = sum_d dO[b][q][h*DH + d] * O[b][q][h*DH + d]: DH / 8 lanes per (b, q, h), 16-byte loads, the sum over those lanes by xor shuffles
template <int DH>
__global__ __launch_bounds__(256) void attn_delta_kernel(const bf16_t* __restrict__ o, int64_t o_rs, int64_t o_bs, const bf16_t* __restrict__ dout, int64_t d_rs,
                                                         int64_t d_bs, float* __restrict__ delta, int64_t stat_ld, int64_t total, int heads, int Lq) {
    constexpr int LPV = DH / 8, VPW = 64 / LPV;
    const int lane = threadIdx.x & 63, sub = lane % LPV;
    const int64_t id = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * VPW + lane / LPV;
    const int64_t idc = id < total ? id : total - 1;
    const int hd = (int)(idc % heads); const int64_t t = idc / heads; const int q = (int)(t % Lq); const int64_t b = t / Lq;
    const bf16x8 a = *(const bf16x8*)(o + b * o_bs + (int64_t)q * o_rs + hd * DH + 8 * sub);
    const bf16x8 g = *(const bf16x8*)(dout + b * d_bs + (int64_t)q * d_rs + hd * DH + 8 * sub);
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) acc += bf2f((bf16_t)a[k]) * bf2f((bf16_t)g[k]);
#pragma unroll
    for (int m = 1; m < LPV; m <<= 1) acc += __shfl_xor(acc, m, 64);
    if (sub == 0 && id < total) delta[(b * heads + hd) * stat_ld + q] = acc;
}

}  // namespace

static int flash_attn_fwd_impl(const void* q, int64_t q_row_stride, int64_t q_batch_stride, const void* k,
                               int64_t k_row_stride, int64_t k_batch_stride, const void* v, int64_t v_row_stride,
                               int64_t v_batch_stride, void* o, int64_t o_row_stride, int64_t o_batch_stride,
                               int64_t batches, int32_t heads, int64_t Lq, int64_t Lkv, int32_t dh, float softmax_scale,
                               float* lse_out, int64_t lse_ld, ug_stream_t stream) {
    if (batches == 0 || Lq == 0) return UG_OK;
    UG_REQUIRE(q && k && v && o && batches > 0 && heads > 0 && Lq > 0 && Lkv > 0, UG_ERR_BAD_SHAPE, "ug_flash_attn_fwd: bad arguments");
    UG_REQUIRE(dh == 128 || dh == 64, UG_ERR_UNSUPPORTED, "ug_flash_attn_fwd: head dim %d not in {64, 128}", dh);
    UG_REQUIRE(Lq < (1 << 30) && Lkv < (1 << 30), UG_ERR_UNSUPPORTED, "ug_flash_attn_fwd: sequence too long");
    UG_REQUIRE(q_row_stride % 8 == 0 && k_row_stride % 8 == 0 && v_row_stride % 8 == 0 && o_row_stride % 4 == 0 &&
               q_batch_stride % 8 == 0 && k_batch_stride % 8 == 0 && v_batch_stride % 8 == 0 && o_batch_stride % 4 == 0 &&
               ug_aligned(q, 16) && ug_aligned(k, 16) && ug_aligned(v, 16) && ug_aligned(o, 8),
               UG_ERR_BAD_ALIGN, "ug_flash_attn_fwd: strides must be multiples of 8 elements and bases 16-byte aligned");
#ifndef UG_PROBE_BUILD
    // PRODUCT BUILD: one kernel per head width - the X|Y stagger with LDS-DMA staging and 16-byte stores; at head width 128 with s_setprio around the
    // softmax segment (PRIO 3), at head width 64 the <= 128-register form so that two workgroups share a CU (KV 64, OCC 4). Every other form that was
    // built and measured (lock-step loop, 4-wave workgroups, one wave per SIMD, register staging, 128-key tiles, priority variants) is compiled only
    // into the probe library (python -m unigen_amd.build --probe; tools/probe/README.md), where UG_ATTN_* select it.
    constexpr int nw = 8;
    const int qrows = 32 * nw;
    const int nQ = (int)((Lq + qrows - 1) / qrows);
    const int64_t nwg = (int64_t)nQ * heads * batches;
    UG_REQUIRE(nwg < (1ll << 31), UG_ERR_UNSUPPORTED, "ug_flash_attn_fwd: grid too large");
    const float c = softmax_scale * 1.4426950408889634f;
#define UG_ATTN_LAUNCH_KV(KVV, OCCV, DHV, NWV, STG, ...)                                                                                  \
    hipLaunchKernelGGL((flash_attn_kernel<DHV, NWV, STG, __VA_ARGS__, KVV, OCCV>), dim3((unsigned)nwg), dim3(64 * NWV), 2 * 2 * KVV * 2 * DHV + (STG ? 32 * NWV * 2 * DHV : 0), (hipStream_t)stream, \
                       (const bf16_t*)q, q_row_stride, q_batch_stride, (const bf16_t*)k, k_row_stride, k_batch_stride, (const bf16_t*)v, \
                       v_row_stride, v_batch_stride, (bf16_t*)o, o_row_stride, o_batch_stride, (int)heads, (int)Lq, (int)Lkv, nQ, c, lse_out, lse_ld)
    // (Round 4: the same stagger on v_mfma_f32_16x16x32_bf16 - flash_attn_m16_kernel, probe library, UG_ATTN_M16=1 - measured +3.5...+4.2 % in a
    // sustained interleaved A/B at 4608^2 / 4096 x 4608 / 8192 x 8704 and -5.6 % INSIDE the cfg2 forward (1113 vs 1177 TFLOP/s, same box, same
    // library, profiles/r04h_attn_m16_in_app.log): its advantage is the higher clock the chip reaches for that shape after ~10 ms of back-to-back
    // launches; a 0.9 ms launch between GEMMs never gets there, and at equal clock its 64 MFMA issues per tile cost the partner wave's softmax more
    // issue slots than 32 do. Not shipped.)
    // UG_ATTN_LSUM_128 / UG_ATTN_LSUM_64 (build-time, 0 | 1 | 2): the row sums on the matrix pipe (template parameter LSUM: 1 = one accumulation chain, 2 = two), per head width
#ifndef UG_ATTN_LSUM_128
#define UG_ATTN_LSUM_128 0
#endif
#ifndef UG_ATTN_LSUM_64
#define UG_ATTN_LSUM_64 1      // round 6: +1.5...+2.6 % on the cfg5 shapes by itself, +3.9...+6.3 % together with BUFD (profiles/r06_attn_variants.log); head width 128: -4 % (X is its longer segment)
#endif
#ifndef UG_ATTN_AIS_128
#define UG_ATTN_AIS_128 0
#endif
#ifndef UG_ATTN_AIS_64
#define UG_ATTN_AIS_64 0
#endif
#ifndef UG_ATTN_PRIO_64
#define UG_ATTN_PRIO_64 0      // 1: s_setprio around the matrix stream, 3: around the softmax segment (the head-width-128 default)
#endif
#ifndef UG_ATTN_BUFD_64
#define UG_ATTN_BUFD_64 1      // round 6: +2.3...+4.6 %, bit-identical (profiles/r06_attn_variants.log)
#endif
#define UG_ATTN_LAUNCH_LS(KVV, OCCV, LS, AISV, BUFV, DHV, NWV, STG, ...)                                                                                  \
    hipLaunchKernelGGL((flash_attn_kernel<DHV, NWV, STG, __VA_ARGS__, KVV, OCCV, LS, AISV, BUFV>), dim3((unsigned)nwg), dim3(64 * NWV), 2 * 2 * KVV * 2 * DHV + (STG ? 32 * NWV * 2 * DHV : 0), (hipStream_t)stream, \
                       (const bf16_t*)q, q_row_stride, q_batch_stride, (const bf16_t*)k, k_row_stride, k_batch_stride, (const bf16_t*)v, \
                       v_row_stride, v_batch_stride, (bf16_t*)o, o_row_stride, o_batch_stride, (int)heads, (int)Lq, (int)Lkv, nQ, c, lse_out, lse_ld)
#ifndef UG_ATTN_NW16_64
#define UG_ATTN_NW16_64 0
#endif
    // the buffer-form DMAs (BUFD) need one row stride for K and V, a multiple of 16 elements, and byte offsets of a (batch, head)'s keys below 2^31
    const bool bufd_ok = UG_ATTN_BUFD_64 != 0 && k_row_stride == v_row_stride && k_row_stride % 16 == 0 && Lkv * k_row_stride * 2 < (1ll << 31);
#ifndef UG_ATTN_BUFD_128
#define UG_ATTN_BUFD_128 0
#endif
    const bool bufd128_ok = UG_ATTN_BUFD_128 != 0 && Lkv * k_row_stride * 2 < (1ll << 31) && Lkv * v_row_stride * 2 < (1ll << 31);
    // (AIS 5 at head width 128 needs the launch's Q-image area as its third K slot: it is part of the stagger kernels' LDS request either way)
    if (dh == 128 && bufd128_ok) UG_ATTN_LAUNCH_LS(64, 2, UG_ATTN_LSUM_128, UG_ATTN_AIS_128, true, 128, 8, true, 3, true, true);
    else if (dh == 128) UG_ATTN_LAUNCH_LS(64, 2, UG_ATTN_LSUM_128, UG_ATTN_AIS_128, false, 128, 8, true, 3, true, true);
#if UG_ATTN_NW16_64       // measured and not shipped (round 6): -0.8...+1.8 % - shorter segments, but every barrier now waits for the slowest of 16 waves
    else if (bufd_ok) {
        // ONE 16-wave workgroup per CU (512 query rows) instead of two 8-wave ones: the same four waves per SIMD, but one K / V stream for all of them
        // (half the L2 -> LDS traffic and half the DMA instructions per query); 96 KiB of LDS (K | V double buffer 32 + the Q image 64)
        const int nQ16 = (int)((Lq + 511) / 512);
        const int64_t nwg16 = (int64_t)nQ16 * heads * batches;
        static bool attr16 = false;
        if (!attr16) { (void)hipFuncSetAttribute((const void*)flash_attn_kernel<64, 16, true, 0, true, true, 64, 4, UG_ATTN_LSUM_64, (UG_ATTN_AIS_64 == 3 ? 0 : UG_ATTN_AIS_64), true>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 2 * 64 * 2 * 64 + 32 * 16 * 2 * 64); attr16 = true; }
        hipLaunchKernelGGL((flash_attn_kernel<64, 16, true, 0, true, true, 64, 4, UG_ATTN_LSUM_64, (UG_ATTN_AIS_64 == 3 ? 0 : UG_ATTN_AIS_64), true>), dim3((unsigned)nwg16), dim3(1024), 2 * 2 * 64 * 2 * 64 + 32 * 16 * 2 * 64, (hipStream_t)stream,
                           (const bf16_t*)q, q_row_stride, q_batch_stride, (const bf16_t*)k, k_row_stride, k_batch_stride, (const bf16_t*)v,
                           v_row_stride, v_batch_stride, (bf16_t*)o, o_row_stride, o_batch_stride, (int)heads, (int)Lq, (int)Lkv, nQ16, c, lse_out, lse_ld);
    }
#endif
    else if (bufd_ok) UG_ATTN_LAUNCH_LS(64, 4, UG_ATTN_LSUM_64, UG_ATTN_AIS_64, true, 64, 8, true, UG_ATTN_PRIO_64, true, true);
    else UG_ATTN_LAUNCH_LS(64, 4, UG_ATTN_LSUM_64, (UG_ATTN_AIS_64 == 3 ? 1 : UG_ATTN_AIS_64 == 4 ? 2 : UG_ATTN_AIS_64), false, 64, 8, true, 0, true, true);
#undef UG_ATTN_LAUNCH_LS
#undef UG_ATTN_LAUNCH_KV
#else
    static int nw = -1;
    if (nw < 0) { const char* e = getenv("UG_ATTN_WAVES"); nw = (e && atoi(e) == 4) ? 4 : 8; }   // 8 measured faster (841 vs 800 TFLOP/s at L = 4608)
    const int qrows = 32 * nw;
    const int nQ = (int)((Lq + qrows - 1) / qrows);
    const int64_t nwg = (int64_t)nQ * heads * batches;
    UG_REQUIRE(nwg < (1ll << 31), UG_ERR_UNSUPPORTED, "ug_flash_attn_fwd: grid too large");
    const float c = softmax_scale * 1.4426950408889634f;
#define UG_ATTN_LAUNCH_KV(KVV, OCCV, DHV, NWV, STG, ...)                                                                                  \
    hipLaunchKernelGGL((flash_attn_kernel<DHV, NWV, STG, __VA_ARGS__, KVV, OCCV>), dim3((unsigned)nwg), dim3(64 * NWV), 2 * 2 * KVV * 2 * DHV + (STG ? 32 * NWV * 2 * DHV : 0), (hipStream_t)stream, \
                       (const bf16_t*)q, q_row_stride, q_batch_stride, (const bf16_t*)k, k_row_stride, k_batch_stride, (const bf16_t*)v, \
                       v_row_stride, v_batch_stride, (bf16_t*)o, o_row_stride, o_batch_stride, (int)heads, (int)Lq, (int)Lkv, nQ, c, lse_out, lse_ld)
#define UG_ATTN_LAUNCH(DHV, NWV, STG, ...)                                                                                           \
    hipLaunchKernelGGL((flash_attn_kernel<DHV, NWV, STG, ##__VA_ARGS__>), dim3((unsigned)nwg), dim3(64 * NWV), 2 * 2 * KVB * 2 * DHV + (STG ? 32 * NWV * 2 * DHV : 0), (hipStream_t)stream, \
                       (const bf16_t*)q, q_row_stride, q_batch_stride, (const bf16_t*)k, k_row_stride, k_batch_stride, (const bf16_t*)v, \
                       v_row_stride, v_batch_stride, (bf16_t*)o, o_row_stride, o_batch_stride, (int)heads, (int)Lq, (int)Lkv, nQ, c, lse_out, lse_ld)
    const int m16 = ug_env_int("UG_ATTN_M16", 0);                     // 1: always, -1: head width 128 from 2048 keys on (what round 4 tried in the product), 0: never
    if (dh == 128 && nw == 8 && (m16 == 1 || (m16 < 0 && Lkv >= 2048))) {       // round 4: the stagger kernel on v_mfma_f32_16x16x32_bf16
        const int pr16 = ug_env_int("UG_ATTN_PRIO", 0);
        if (pr16 == 1)
            hipLaunchKernelGGL((flash_attn_m16_kernel<128, 1>), dim3((unsigned)nwg), dim3(512), 4 * 64 * 2 * 128, (hipStream_t)stream, (const bf16_t*)q, q_row_stride, q_batch_stride,
                               (const bf16_t*)k, k_row_stride, k_batch_stride, (const bf16_t*)v, v_row_stride, v_batch_stride, (bf16_t*)o, o_row_stride, o_batch_stride,
                               (int)heads, (int)Lq, (int)Lkv, nQ, c, lse_out, lse_ld);
        else if (pr16 == 3)
            hipLaunchKernelGGL((flash_attn_m16_kernel<128, 3>), dim3((unsigned)nwg), dim3(512), 4 * 64 * 2 * 128, (hipStream_t)stream, (const bf16_t*)q, q_row_stride, q_batch_stride,
                               (const bf16_t*)k, k_row_stride, k_batch_stride, (const bf16_t*)v, v_row_stride, v_batch_stride, (bf16_t*)o, o_row_stride, o_batch_stride,
                               (int)heads, (int)Lq, (int)Lkv, nQ, c, lse_out, lse_ld);
        else
            hipLaunchKernelGGL((flash_attn_m16_kernel<128, 0>), dim3((unsigned)nwg), dim3(512), 4 * 64 * 2 * 128, (hipStream_t)stream, (const bf16_t*)q, q_row_stride, q_batch_stride,
                               (const bf16_t*)k, k_row_stride, k_batch_stride, (const bf16_t*)v, v_row_stride, v_batch_stride, (bf16_t*)o, o_row_stride, o_batch_stride,
                               (int)heads, (int)Lq, (int)Lkv, nQ, c, lse_out, lse_ld);
        UG_CHECK_LAUNCH("ug_flash_attn_fwd");
        return UG_OK;
    }
    if (dh == 64 && nw == 8 && m16 == 1) {        // head width 64 on the same kernel (one workgroup per CU): A/B only
        hipLaunchKernelGGL((flash_attn_m16_kernel<64, 0>), dim3((unsigned)nwg), dim3(512), 4 * 64 * 2 * 64, (hipStream_t)stream, (const bf16_t*)q, q_row_stride, q_batch_stride,
                           (const bf16_t*)k, k_row_stride, k_batch_stride, (const bf16_t*)v, v_row_stride, v_batch_stride, (bf16_t*)o, o_row_stride, o_batch_stride,
                           (int)heads, (int)Lq, (int)Lkv, nQ, c, lse_out, lse_ld);
        UG_CHECK_LAUNCH("ug_flash_attn_fwd");
        return UG_OK;
    }
    static int pwg = -1;
    if (pwg < 0) { const char* e = getenv("UG_ATTN_PWG"); pwg = e ? atoi(e) : 0; }
    if (pwg && dh == 128 && !lse_out) {
        const int nQp = (int)((Lq + 255) / 256);
        const int64_t nwgp = (int64_t)nQp * heads * batches;
        static bool attr = false;
        if (!attr) { (void)hipFuncSetAttribute((const void*)flash_attn_pwg_kernel<128>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * KVB * 2 * 128); attr = true; }
        hipLaunchKernelGGL((flash_attn_pwg_kernel<128>), dim3((unsigned)nwgp), dim3(256), 4 * KVB * 2 * 128, (hipStream_t)stream,
                           (const bf16_t*)q, q_row_stride, q_batch_stride, (const bf16_t*)k, k_row_stride, k_batch_stride, (const bf16_t*)v,
                           v_row_stride, v_batch_stride, (bf16_t*)o, o_row_stride, o_batch_stride, (int)heads, (int)Lq, (int)Lkv, nQp, c);
        UG_CHECK_LAUNCH("ug_flash_attn_fwd");
        return UG_OK;
    }
    static int stagger = -1;
    // UG_ATTN_STAGGER=0 selects the lock-step loop; default: the X|Y stagger. Same-box A/B after the branch-free fetch and the softmax pin
    // (before them the stagger variant measured 602 vs 842): dh = 128: 989 vs 955 TFLOP/s at L = 4608, 1027 vs 1007 (4096 x 4608),
    // 1068 vs 1044 (8192), 956 vs 933 (B16, 2048); dh = 64 inside the SD3.5 forward: 795 vs 775.
    if (stagger < 0) { const char* e = getenv("UG_ATTN_STAGGER"); stagger = (e && atoi(e) == 0) ? 0 : 1; }
    const bool stg = stagger == 1;
    // UG_ATTN_PRIO = 0 | 1 | 2 | 3, UG_ATTN_WIDE = 0 | 1: A/B switches of the stagger kernel (re-read per call when UG_ENV_DYNAMIC=1)
    // Interleaved A/B, round 2 (tools/attn_ab.py, same process): prio 0 + wide stores is the fastest form everywhere - dh 128: 1137 / 1147 / 1178
    // vs 1124 / 1133 / 1172 TFLOP/s for the round-1 default (prio 1, narrow) at 4608^2 / 4096x4608 / 8192x8704; dh 64: 880-885 vs 856-871; the
    // static young-half priority (2) loses 1-2 % at dh 128.
    // UG_ATTN_PRIO = 3 (DMA variant): s_setprio 1 around the softmax segment - it is the longer one of each segment pair (loop ablations,
    // profiles/r02f_attn_bwd.log). Interleaved A/B, 12 of 12 pairs: +0.3-0.5 % at dh 128 (1137 -> 1141, 1153 -> 1158, 1185 -> 1191 TFLOP/s);
    // dh 64: -0.5 % (within noise) -> default 3 at dh 128, 0 at dh 64.
    const int prio = ug_env_int("UG_ATTN_PRIO", dh == 128 ? 3 : 0), wide = ug_env_int("UG_ATTN_WIDE", 1), dma = ug_env_int("UG_ATTN_DMA", 1);
#define UG_ATTN_STG(DHV)                                                                          \
    do {                                                                                          \
        if (dma && prio == 3) { UG_ATTN_LAUNCH(DHV, 8, true, 3, true, true); break; }             \
        if (dma) { UG_ATTN_LAUNCH(DHV, 8, true, 0, true, true); break; }                          \
        if (wide) { if (prio == 0 || prio == 3) UG_ATTN_LAUNCH(DHV, 8, true, 0, true); else if (prio == 2) UG_ATTN_LAUNCH(DHV, 8, true, 2, true); else UG_ATTN_LAUNCH(DHV, 8, true, 1, true); } \
        else { if (prio == 0 || prio == 3) UG_ATTN_LAUNCH(DHV, 8, true, 0, false); else if (prio == 2) UG_ATTN_LAUNCH(DHV, 8, true, 2, false); else UG_ATTN_LAUNCH(DHV, 8, true, 1, false); } \
    } while (0)
    // Head dim 64 (round 3, UG_ATTN_KV64): 464 (default) = 64-key tiles at <= 128 registers so that TWO workgroups share a CU; 128 = 128-key
    // tiles, one workgroup per CU; 64 = the round-2 kernel. Interleaved A/B (tools/attn_ab.py, profiles/r03e_attn_ab64.log): alone
    // 872 / 906 / 902 TFLOP/s (64 / 128 / 464) at 4096 x 4429, 886 / 920 / 912 at 4096^2; inside the SD3.5 forward, where other kernels'
    // tails and launches leave more bubbles to fill, 834 / 868 / 918 (0.4214 / 0.4144 / 0.4043 s per forward). 464 is bit-identical to 64.
    const int kv64 = ug_env_int("UG_ATTN_KV64", 464);
    if (dh == 128) { if (nw == 4) UG_ATTN_LAUNCH(128, 4, false); else if (stg) UG_ATTN_STG(128); else UG_ATTN_LAUNCH(128, 8, false); }
    else if (nw == 8 && stg && dma && kv64 == 128) {
        static bool attr = false;
        if (!attr) { (void)hipFuncSetAttribute((const void*)flash_attn_kernel<64, 8, true, 0, true, true, 128, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 2 * 128 * 2 * 64 + 32 * 8 * 2 * 64); attr = true; }
        UG_ATTN_LAUNCH_KV(128, 2, 64, 8, true, 0, true, true);
    }
    else if (nw == 8 && stg && dma && kv64 == 464) {        // UG_ATTN_KV64=464: 64-key tiles, <= 128 registers, two workgroups per CU
        UG_ATTN_LAUNCH_KV(64, 4, 64, 8, true, 0, true, true);
    }
    else           { if (nw == 4) UG_ATTN_LAUNCH(64, 4, false); else if (stg) UG_ATTN_STG(64); else UG_ATTN_LAUNCH(64, 8, false); }
#undef UG_ATTN_LAUNCH_KV
#undef UG_ATTN_STG
#undef UG_ATTN_LAUNCH
#endif   // UG_PROBE_BUILD
    UG_CHECK_LAUNCH("ug_flash_attn_fwd");
    return UG_OK;
}

extern "C" int ug_flash_attn_fwd(const void* q, int64_t q_row_stride, int64_t q_batch_stride, const void* k,
                                 int64_t k_row_stride, int64_t k_batch_stride, const void* v, int64_t v_row_stride,
                                 int64_t v_batch_stride, void* o, int64_t o_row_stride, int64_t o_batch_stride,
                                 int64_t batches, int32_t heads, int64_t Lq, int64_t Lkv, int32_t dh, float softmax_scale,
                                 ug_stream_t stream) {
    return flash_attn_fwd_impl(q, q_row_stride, q_batch_stride, k, k_row_stride, k_batch_stride, v, v_row_stride, v_batch_stride, o, o_row_stride,
                               o_batch_stride, batches, heads, Lq, Lkv, dh, softmax_scale, nullptr, 0, stream);
}

extern "C" int ug_flash_attn_fwd_lse(const void* q, int64_t q_row_stride, int64_t q_batch_stride, const void* k,
                                     int64_t k_row_stride, int64_t k_batch_stride, const void* v, int64_t v_row_stride,
                                     int64_t v_batch_stride, void* o, int64_t o_row_stride, int64_t o_batch_stride,
                                     int64_t batches, int32_t heads, int64_t Lq, int64_t Lkv, int32_t dh, float softmax_scale,
                                     float* lse2, int64_t lse_ld, ug_stream_t stream) {
    UG_REQUIRE(lse2 && lse_ld >= Lq, UG_ERR_BAD_SHAPE, "ug_flash_attn_fwd_lse: lse2 [batches][heads][lse_ld >= Lq] needed");
    return flash_attn_fwd_impl(q, q_row_stride, q_batch_stride, k, k_row_stride, k_batch_stride, v, v_row_stride, v_batch_stride, o, o_row_stride,
                               o_batch_stride, batches, heads, Lq, Lkv, dh, softmax_scale, lse2, lse_ld, stream);
}

extern "C" int64_t ug_flash_attn_bwd_workspace_bytes(int64_t batches, int32_t heads, int64_t Lq) {
    if (batches <= 0 || heads <= 0 || Lq <= 0) return 0;
    return 2 * batches * heads * ((Lq + 63) / 64 * 64) * (int64_t)sizeof(float);
}

extern "C" int ug_flash_attn_bwd(const void* q, int64_t q_rs, int64_t q_bs, const void* k, int64_t k_rs, int64_t k_bs, const void* v, int64_t v_rs,
                                 int64_t v_bs, const void* o, int64_t o_rs, int64_t o_bs, const void* dout, int64_t do_rs, int64_t do_bs, void* dq,
                                 int64_t dq_rs, int64_t dq_bs, void* dk, int64_t dk_rs, int64_t dk_bs, void* dv, int64_t dv_rs, int64_t dv_bs,
                                 int64_t batches, int32_t heads, int64_t Lq, int64_t Lkv, int32_t dh, float softmax_scale, const float* lse_in,
                                 void* workspace, int64_t workspace_bytes, ug_stream_t stream) {
    if (batches == 0 || Lq == 0) return UG_OK;
    UG_REQUIRE(q && k && v && o && dout && dq && dk && dv && batches > 0 && heads > 0 && Lq > 0 && Lkv > 0, UG_ERR_BAD_SHAPE, "ug_flash_attn_bwd: bad arguments");
    UG_REQUIRE(dh == 128 || dh == 64, UG_ERR_UNSUPPORTED, "ug_flash_attn_bwd: head dim %d not in {64, 128}", dh);
    UG_REQUIRE(Lq < (1 << 30) && Lkv < (1 << 30), UG_ERR_UNSUPPORTED, "ug_flash_attn_bwd: sequence too long");
    const int64_t strides[] = {q_rs, q_bs, k_rs, k_bs, v_rs, v_bs, o_rs, o_bs, do_rs, do_bs, dq_rs, dq_bs, dk_rs, dk_bs, dv_rs, dv_bs};
    for (int64_t sgl : strides) UG_REQUIRE(sgl % 8 == 0, UG_ERR_BAD_ALIGN, "ug_flash_attn_bwd: strides must be multiples of 8 elements");
    UG_REQUIRE(ug_aligned(q, 16) && ug_aligned(k, 16) && ug_aligned(v, 16) && ug_aligned(o, 16) && ug_aligned(dout, 16) && ug_aligned(dq, 8) &&
               ug_aligned(dk, 8) && ug_aligned(dv, 8), UG_ERR_BAD_ALIGN, "ug_flash_attn_bwd: bases must be 16-byte aligned");
    const int64_t stat_ld = (Lq + 63) / 64 * 64;
    UG_REQUIRE(workspace && ug_aligned(workspace, 16) && workspace_bytes >= ug_flash_attn_bwd_workspace_bytes(batches, heads, Lq), UG_ERR_BAD_SHAPE,
               "ug_flash_attn_bwd: workspace of ug_flash_attn_bwd_workspace_bytes() needed");
    // lse_in: the statistics ug_flash_attn_fwd_lse wrote, [batches][heads][stat_ld] with the padding zero (saves the LSE launch); else computed here
    float* lse2 = lse_in ? const_cast<float*>(lse_in) : (float*)workspace;
    float* delta = (float*)workspace + batches * heads * stat_ld;
    hipStream_t s = (hipStream_t)stream;
    const float c = softmax_scale * 1.4426950408889634f;
    const int bwd_dma = UG_TUNE("UG_ATTN_BWD_DMA", 1);     // LDS-DMA staging of the streamed tiles (0: through registers)
    const int nQ = (int)((Lq + 255) / 256), nK = (int)((Lkv + 255) / 256);
    const int64_t gq = (int64_t)nQ * heads * batches, gk = (int64_t)nK * heads * batches;
    UG_REQUIRE(gq < (1ll << 31) && gk < (1ll << 31), UG_ERR_UNSUPPORTED, "ug_flash_attn_bwd: grid too large");
    // fused dK / dV kernel (128 keys per workgroup; needs the LDS-DMA staging and equal row strides are NOT required); UG_ATTN_BWD_FUSE_DKV=0: the two modes
    const int nK2 = (int)((Lkv + 127) / 128);
    const int64_t gk2 = (int64_t)nK2 * heads * batches;
    UG_REQUIRE(gk2 < (1ll << 31) && (int64_t)((Lq + 127) / 128) * heads * batches < (1ll << 31), UG_ERR_UNSUPPORTED, "ug_flash_attn_bwd: grid too large");
    const bool fuse_dkv = bwd_dma && UG_TUNE("UG_ATTN_BWD_FUSE_DKV", 1);
    // dQ on 128-query workgroups (attn_bwd_dq_kernel); UG_ATTN_BWD_PAIR_DQ=0: the 256-query DQ mode
    const int nQ2 = (int)((Lq + 127) / 128);
    const int64_t gq2 = (int64_t)nQ2 * heads * batches;
    // Measured (tools/attn_bwd_ab.py, profiles/r03y_attn_bwd_pair_dq.log): dh 128 at 4608^2 / 8704^2 +2.6 % / +2.5 % of the whole backward; dh 128 at
    // 1000^2 -7 %, dh 64 -3 % (half the MFMAs per wave and barrier) -> on by default only for head width 128 and >= 2048 queries (2 forces it everywhere)
    const int pdq = UG_TUNE("UG_ATTN_BWD_PAIR_DQ", 1);
    const bool pair_dq = bwd_dma && gq2 < (1ll << 31) && (pdq == 2 || (pdq == 1 && dh == 128 && Lq >= 2048));
    (void)hipMemsetAsync(workspace, 0, (size_t)(2 * batches * heads * stat_ld) * sizeof(float), s);     // padded statistics rows read as 0
    const int64_t total = batches * Lq * heads;
    if (dh == 128)
        hipLaunchKernelGGL(attn_delta_kernel<128>, dim3((unsigned)((total + 15) / 16)), dim3(256), 0, s, (const bf16_t*)o, o_rs, o_bs, (const bf16_t*)dout, do_rs, do_bs,
                           delta, stat_ld, total, (int)heads, (int)Lq);
    else
        hipLaunchKernelGGL(attn_delta_kernel<64>, dim3((unsigned)((total + 31) / 32)), dim3(256), 0, s, (const bf16_t*)o, o_rs, o_bs, (const bf16_t*)dout, do_rs, do_bs,
                           delta, stat_ld, total, (int)heads, (int)Lq);
#ifdef UG_PROBE_BUILD
#define UG_BWD(DHV, MODEV, GRID, ...)                                                                                                                  \
    do { if (bwd_dma) UG_BWD_(DHV, MODEV, true, GRID, __VA_ARGS__); else UG_BWD_(DHV, MODEV, false, GRID, __VA_ARGS__); } while (0)
#define UG_BWD_SPLIT_DKV(DHV)                                                                                                                       \
    do {                                                                                                                                            \
        UG_BWD(DHV, BWD_DK, gk, k, k_rs, k_bs, v, v_rs, v_bs, q, q_rs, q_bs, dout, do_rs, do_bs, dk, dk_rs, dk_bs, Lkv, Lq, nK);                   \
        UG_BWD(DHV, BWD_DV, gk, k, k_rs, k_bs, nullptr, 0, 0, q, q_rs, q_bs, dout, do_rs, do_bs, dv, dv_rs, dv_bs, Lkv, Lq, nK);                   \
    } while (0)
#else      /* product: LDS-DMA staging only; dK and dV always by the fused kernel (the two separate modes are probe-build variants) */
#define UG_BWD(DHV, MODEV, GRID, ...) UG_BWD_(DHV, MODEV, true, GRID, __VA_ARGS__)
#define UG_BWD_SPLIT_DKV(DHV) do { } while (0)
#endif
#define UG_BWD_(DHV, MODEV, DMAV, GRID, O1, O1R, O1B, O2, O2R, O2B, S1, S1R, S1B, S2, S2R, S2B, OUT, OR, OB, LOWN, LST, NOWN)                          \
    do {                                                                                                                                                \
        const int lds_ = 2 * (2 * KVB * 2 * DHV + (DMAV ? 512 : 0));                                                                                   \
        static bool attr_ = false;        /* one per expansion site = per instantiation */                                                              \
        if (!attr_) { (void)hipFuncSetAttribute((const void*)attn_bwd_kernel<DHV, MODEV, DMAV>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_); attr_ = true; } \
        hipLaunchKernelGGL((attn_bwd_kernel<DHV, MODEV, DMAV>), dim3((unsigned)(GRID)), dim3(512), lds_, s, (const bf16_t*)(O1), O1R, O1B, (const bf16_t*)(O2), O2R, O2B, \
                           (const bf16_t*)(S1), S1R, S1B, (const bf16_t*)(S2), S2R, S2B, lse2, delta, stat_ld, (bf16_t*)(OUT), OR, OB, (int)heads, (int)(LOWN),   \
                           (int)(LST), (int)(NOWN), c, softmax_scale);                                                                                 \
    } while (0)
#define UG_BWD_PAIR_DQ(DHV)                                                                                                                         \
    do {                                                                                                                                            \
        constexpr int ldsq_ = 3 * (2 * KVB * 2 * DHV);                                                                                              \
        static bool attrq_ = false;                                                                                                                 \
        if (!attrq_) { (void)hipFuncSetAttribute((const void*)attn_bwd_dq_kernel<DHV>, hipFuncAttributeMaxDynamicSharedMemorySize, ldsq_); attrq_ = true; } \
        hipLaunchKernelGGL((attn_bwd_dq_kernel<DHV>), dim3((unsigned)gq2), dim3(512), ldsq_, s, (const bf16_t*)q, q_rs, q_bs, (const bf16_t*)dout, do_rs, do_bs, \
                           (const bf16_t*)k, k_rs, k_bs, (const bf16_t*)v, v_rs, v_bs, lse2, delta, stat_ld, (bf16_t*)dq, dq_rs, dq_bs, (int)heads,        \
                           (int)Lq, (int)Lkv, nQ2, c, softmax_scale);                                                                               \
    } while (0)
#define UG_BWD_PAIR_DQ_128() UG_BWD_PAIR_DQ(128)
#ifdef UG_PROBE_BUILD
#define UG_BWD_PAIR_DQ_64() UG_BWD_PAIR_DQ(64)
#else      /* product: the pair-scheme dQ kernel only runs at head width 128 (pair_dq above); no head-width-64 instantiation */
#define UG_BWD_PAIR_DQ_64() do { } while (0)
#endif
#define UG_BWD_ALL(DHV)                                                                                                                              \
    do {                                                                                                                                              \
        if (!lse_in) UG_BWD(DHV, BWD_LSE, gq, q, q_rs, q_bs, nullptr, 0, 0, k, k_rs, k_bs, nullptr, 0, 0, nullptr, 0, 0, Lq, Lkv, nQ);             \
        if (pair_dq) {                                                                                                                             \
            UG_BWD_PAIR_DQ_##DHV();                                                                                                                 \
        } else {                                                                                                                                    \
            UG_BWD(DHV, BWD_DQ, gq, q, q_rs, q_bs, dout, do_rs, do_bs, k, k_rs, k_bs, v, v_rs, v_bs, dq, dq_rs, dq_bs, Lq, Lkv, nQ);               \
        }                                                                                                                                           \
        if (fuse_dkv) {                                                                                                                            \
            constexpr int lds_ = 3 * (2 * KVB * 2 * DHV + 512) + 2 * 8 * 2048;                                                                          \
            static bool attr_ = false;                                                                                                              \
            if (!attr_) { (void)hipFuncSetAttribute((const void*)attn_bwd_dkv_kernel<DHV>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_); attr_ = true; } \
            hipLaunchKernelGGL((attn_bwd_dkv_kernel<DHV>), dim3((unsigned)gk2), dim3(512), lds_, s, (const bf16_t*)k, k_rs, k_bs, (const bf16_t*)v, v_rs, v_bs,  \
                               (const bf16_t*)q, q_rs, q_bs, (const bf16_t*)dout, do_rs, do_bs, lse2, delta, stat_ld, (bf16_t*)dk, dk_rs, dk_bs, (bf16_t*)dv,   \
                               dv_rs, dv_bs, (int)heads, (int)Lkv, (int)Lq, nK2, c, softmax_scale);                                                  \
        } else {                                                                                                                                    \
            UG_BWD_SPLIT_DKV(DHV);                                                                                                                  \
        }                                                                                                                                           \
    } while (0)
    if (dh == 128) UG_BWD_ALL(128); else UG_BWD_ALL(64);
#undef UG_BWD_ALL
#undef UG_BWD_PAIR_DQ
#undef UG_BWD_PAIR_DQ_128
#undef UG_BWD_PAIR_DQ_64
#undef UG_BWD_SPLIT_DKV
#undef UG_BWD
#undef UG_BWD_
    UG_CHECK_LAUNCH("ug_flash_attn_bwd");
    return UG_OK;
}
