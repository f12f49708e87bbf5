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

typedef __attribute__((address_space(3))) bf16x4* lds_b64_ptr;

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

// STAGGER (8 waves): waves 0-3 and 4-7 - the two waves of every SIMD - run half a tile apart: in each segment one group does
// S^T + softmax of a tile while the other does P.V of the previous one, so a wave's softmax VALU work runs under its partner's
// MFMAs instead of both waves hitting the matrix pipe, then the VALU, together. Two barriers per tile; K(t+1) is written at the
// end of odd segments and V(t) at the end of even ones, each into the buffer nobody reads in that segment.
template <int DH, int NW, bool STAGGER>   // head dim 128 | 64; waves per workgroup: 8 (256 query rows, 1 / CU) or 4 (128 rows, 2 / CU)
__global__ __launch_bounds__(64 * NW, 2) void flash_attn_kernel(
    const bf16_t* __restrict__ q, int64_t q_rs, int64_t q_bs, const bf16_t* __restrict__ k, int64_t k_rs, int64_t k_bs,
    const bf16_t* __restrict__ v, int64_t v_rs, int64_t v_bs, bf16_t* __restrict__ o, int64_t o_rs, int64_t o_bs,
    int heads, int Lq, int Lkv, int nQ, float c /* softmax_scale * log2(e) */) {
    constexpr int RB = 2 * DH;                       // row bytes
    constexpr int NCH = DH / 8;                      // 16-byte chunks per row
    constexpr int TILE = KVB * RB;                   // bytes of one K (or V) tile image
    constexpr int QS = DH / 16;                      // k-steps of S^T = K Q^T
    constexpr int NDB = DH / 32;                     // 32-wide d blocks of O^T
    constexpr int QROWS = 32 * NW, NT = 64 * NW, NST = (KVB * NCH) / NT;   // staging chunks of K (and of V) per thread and tile
    static_assert(NST >= 1, "tile smaller than the workgroup");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // [2][K tile | V tile]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;

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
    bf16x8 qf[STAGGER ? 1 : QS];
    const int q_lds = QBASE + RB * (wave * 32 + r);
    const int qx = h ^ row_swz<DH>(r);                  // wave * 32 keeps row_swz unchanged (multiple of 16)
    if constexpr (!STAGGER) {
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
    auto stage_load = [&](int kv0) {
#pragma unroll
        for (int u = 0; u < NST; ++u) {
            int key = kv0 + st_row[u]; if (key > Lkv - 1) key = Lkv - 1;
            kreg[u] = *(const u32x4*)(Kb + (int64_t)key * k_rs + st_ch[u] * 8);
            vreg[u] = *(const u32x4*)(Vb + (int64_t)key * v_rs + st_ch[u] * 8);
        }
    };
    auto stage_write = [&](int buf) {
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

    const int ntiles = (Lkv + KVB - 1) / KVB;
    bf16x8 pf[2][2];                                   // P^T fragments of the tile between its S and P stages
    // CUR = buffer parity as a compile-time constant: every LDS address below is then a loop-invariant VGPR + an immediate offset
    // (with a runtime parity hipcc re-materialised ~50 address adds per tile, a quarter of the VALU work of the loop).
    f32x16 sacc[2];                                    // S^T of the tile between its QK^T and its softmax
    auto do_QK = [&](int t, auto cur_c) {
        constexpr int CUR = decltype(cur_c)::value;
        const int kv0 = t * KVB;
        const unsigned char* Kbuf = smem + CUR * 2 * TILE;
        // ---- S^T[key][q]: all 8 K fragments of key block 0 first, then block-0 MFMAs with the block-1 reads between them ----
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int i = 0; i < 16; ++i) sacc[kb][i] = 0.f;
        if constexpr (!STAGGER) {
            bf16x8 kf0[QS], kf1[QS];
#pragma unroll
            for (int s = 0; s < QS; ++s) kf0[s] = *(const bf16x8*)(Kbuf + k_rowoff + 16 * ((2 * s) ^ kx));
#pragma unroll
            for (int s = 0; s < QS; ++s) kf1[s] = *(const bf16x8*)(Kbuf + 32 * RB + k_rowoff + 16 * ((2 * s) ^ kx));
#pragma unroll
            for (int s = 0; s < QS; ++s) sacc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf0[s], qf[s], sacc[0], 0, 0, 0);
#pragma unroll
            for (int s = 0; s < QS; ++s) sacc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf1[s], qf[s], sacc[1], 0, 0, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, QS, 0);       // ds_reads of key block 0
#pragma unroll
            for (int s = 0; s < QS; ++s) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);    // 1 MFMA (block 0)
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);    // 1 ds_read (block 1)
            }
            __builtin_amdgcn_sched_group_barrier(0x008, QS, 0);       // MFMAs of block 1
        } else {
            // Q comes from LDS too: per k-step one Q fragment and the two key blocks' K fragments, read two steps ahead of their
            // MFMAs (9 fragments = 36 VGPRs live instead of 24 fragments if hipcc hoisted every read).
            bf16x8 ql[QS], kf0[QS], kf1[QS];
#pragma unroll
            for (int s = 0; s < QS; ++s) {
                ql[s] = *(const bf16x8*)(smem + q_lds + 16 * ((2 * s) ^ qx));
                kf0[s] = *(const bf16x8*)(Kbuf + k_rowoff + 16 * ((2 * s) ^ kx));
                kf1[s] = *(const bf16x8*)(Kbuf + 32 * RB + k_rowoff + 16 * ((2 * s) ^ kx));
            }
#pragma unroll
            for (int s = 0; s < QS; ++s) {
                sacc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf0[s], ql[s], sacc[0], 0, 0, 0);
                sacc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf1[s], ql[s], sacc[1], 0, 0, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);        // fragments of k-steps 0, 1
#pragma unroll
            for (int s = 0; s < QS - 2; ++s) {
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);    // MFMAs of step s
                __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);    // fragments of step s + 2
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
        }
        if (kv0 + KVB > Lkv) {   // ragged last tile: keys >= Lkv do not exist
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int key = kv0 + kb * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
                    if (key >= Lkv) sacc[kb][i] = -INFINITY;
                }
        }
    };
    auto do_SM = [&]() {
        // ---- online softmax, all lane-local (this lane: query r, 32 of the tile's 64 keys; lane^32 has the rest) ----
        float tmax = sacc[0][0];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int i = 0; i < 16; ++i) tmax = fmaxf(tmax, sacc[kb][i]);
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
        const float m_new = fmaxf(m_run, tmax);
        if (!__all(m_new == m_run)) {
            const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c);
            l_run *= alpha;
#pragma unroll
            for (int db = 0; db < NDB; ++db)
#pragma unroll
                for (int i = 0; i < 16; ++i) oacc[db][i] *= alpha;
            m_run = m_new;
        }
        const float mc = m_run * c;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            float p[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                p[i] = __builtin_amdgcn_exp2f(fmaf(sacc[kb][i], c, -mc));
                l_run += p[i];
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
    auto do_P = [&](int t, auto cur_c) {
        constexpr int CUR = decltype(cur_c)::value;
        const unsigned char* Vbuf = smem + CUR * 2 * TILE + TILE;
        // ---- O^T[d][q] += V^T[d][key] P^T[key][q]: the V fragments of d-block db+1 are read between the MFMAs of block db ----
        {
            bf16x8 vf[NDB][4];
#pragma unroll
            for (int db = 0; db < NDB; ++db)
#pragma unroll
                for (int ks = 0; ks < 4; ++ks)
                    vf[db][ks] = tr_read_pair(Vbuf + ks * 16 * RB + voff_lo[db], Vbuf + ks * 16 * RB + voff_hi[db]);
#pragma unroll
            for (int db = 0; db < NDB; ++db)
#pragma unroll
                for (int ks = 0; ks < 4; ++ks)
                    oacc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[db][ks], pf[ks >> 1][ks & 1], oacc[db], 0, 0, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 8, 1);            // 8 tr reads (d-block 0)
#pragma unroll
            for (int i = 0; i < 4 * (NDB - 1); ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 1);        // 1 MFMA
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 1);        // 2 tr reads of the next d-block
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 1);            // last d-block
        }
    };
    if constexpr (!STAGGER) {
        stage_load(0);
        stage_write(0);
        __syncthreads();
        auto tile = [&](int t, auto cur_c) {
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
        static_assert(!STAGGER || NW == 8, "the stagger pairs waves w and w+4 of one SIMD");
        // X / Y stagger. A wave alternates a MATRIX-only segment X(t) = P.V of tile t followed by S^T = K.Q^T of tile t+1, and a
        // VALU-only segment Y(t+1) = online softmax of tile t+1. Waves 0-3 (group A) and 4-7 (group B) - the two waves of every
        // SIMD - run one segment apart, so in every segment a SIMD has one wave feeding the matrix pipe and one feeding the VALU
        // (PMC on the lock-step loop: matrix pipe busy 42 %, VALU 45 %, hardly overlapping).
        //   global segment:   0       1       2       3       4
        //   group A:        QK(0)    Y(0)    X(0)    Y(1)    X(1) ...
        //   group B:          -     QK(0)    Y(0)    X(0)    Y(1) ...
        // K(t+1) and V(t) are first needed in segment 2t+2: every thread fetches its share at the START of even segment 2t and
        // publishes it at the END of odd segment 2t+1 (into buffers nobody reads in 2t / 2t+1). Loads cross barriers: raw s_barrier.
        auto seg_barrier = [&]() {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        };
        auto fetch = [&](int kt, int vt) {              // K(kt), V(vt) -> registers (tiles beyond the end: nothing)
#pragma unroll
            for (int u = 0; u < NST; ++u) {
                if (kt < ntiles) { int key = kt * KVB + st_row[u]; if (key > Lkv - 1) key = Lkv - 1; kreg[u] = *(const u32x4*)(Kb + (int64_t)key * k_rs + st_ch[u] * 8); }
                if (vt < ntiles) { int key = vt * KVB + st_row[u]; if (key > Lkv - 1) key = Lkv - 1; vreg[u] = *(const u32x4*)(Vb + (int64_t)key * v_rs + st_ch[u] * 8); }
            }
        };
        auto publish = [&](int kt, int vt) {
#pragma unroll
            for (int u = 0; u < NST; ++u) {
                if (kt < ntiles) *(u32x4*)(smem + (kt & 1) * 2 * TILE + st_off[u]) = kreg[u];
                if (vt < ntiles) *(u32x4*)(smem + (vt & 1) * 2 * TILE + TILE + st_off[u]) = vreg[u];
            }
        };
        const bool groupA = __builtin_amdgcn_readfirstlane(wave) < 4;
        fetch(0, ntiles);                              // K(0) only
        publish(0, ntiles);
        seg_barrier();
        fetch(1, 0);                                   // segment 0 (even): K(1), V(0) in flight
        if (!groupA) seg_barrier();                    // B idles through segment 0
        do_QK(0, std::integral_constant<int, 0>{});    // A: segment 0 | B: segment 1
        if (!groupA) publish(1, 0);                    // end of segment 1 (B)
        seg_barrier();
        // one tile = Y(t) | X(t); buffer parity is a compile-time constant (two tiles per trip)
        auto tile = [&](int t, auto cur_c) {
            constexpr int CUR = decltype(cur_c)::value;
            // Y(t): A in odd segment 2t+1 (publishes K(t+1), V(t) at its end) | B in even segment 2t+2 (fetches K(t+2), V(t+1) at its start)
            if (!groupA) fetch(t + 2, t + 1);
            do_SM();
            if (groupA) publish(t + 1, t);
            seg_barrier();
            // X(t) = P.V(t) then K.Q^T(t+1): A in even segment 2t+2 (fetch) | B in odd segment 2t+3 (publish)
            if (groupA) fetch(t + 2, t + 1);
            do_P(t, cur_c);
            if (t + 1 < ntiles) do_QK(t + 1, std::integral_constant<int, CUR ^ 1>{});
            if (!groupA) publish(t + 2, t + 1);
            seg_barrier();
        };
        for (int t = 0; t < ntiles; t += 2) {
            tile(t, std::integral_constant<int, 0>{});
            if (t + 1 < ntiles) tile(t + 1, std::integral_constant<int, 1>{});
        }
        if (groupA) seg_barrier();                     // A's trailing (empty) segment pairs with B's last one
    }

    // ---- epilogue: O[q][d] = O^T / l ----
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_tot;
    if (q_row < Lq) {
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
}

}  // namespace

extern "C" int ug_flash_attn_fwd(const void* q, int64_t q_row_stride, int64_t q_batch_stride, const void* k,
                                 int64_t k_row_stride, int64_t k_batch_stride, const void* v, int64_t v_row_stride,
                                 int64_t v_batch_stride, void* o, int64_t o_row_stride, int64_t o_batch_stride,
                                 int64_t batches, int32_t heads, int64_t Lq, int64_t Lkv, int32_t dh, float softmax_scale,
                                 ug_stream_t stream) {
    if (batches == 0 || Lq == 0) return UG_OK;
    UG_REQUIRE(q && k && v && o && batches > 0 && heads > 0 && Lq > 0 && Lkv > 0, UG_ERR_BAD_SHAPE, "ug_flash_attn_fwd: bad arguments");
    UG_REQUIRE(dh == 128 || dh == 64, UG_ERR_UNSUPPORTED, "ug_flash_attn_fwd: head dim %d not in {64, 128}", dh);
    UG_REQUIRE(Lq < (1 << 30) && Lkv < (1 << 30), UG_ERR_UNSUPPORTED, "ug_flash_attn_fwd: sequence too long");
    UG_REQUIRE(q_row_stride % 8 == 0 && k_row_stride % 8 == 0 && v_row_stride % 8 == 0 && o_row_stride % 4 == 0 &&
               q_batch_stride % 8 == 0 && k_batch_stride % 8 == 0 && v_batch_stride % 8 == 0 && o_batch_stride % 4 == 0 &&
               ug_aligned(q, 16) && ug_aligned(k, 16) && ug_aligned(v, 16) && ug_aligned(o, 8),
               UG_ERR_BAD_ALIGN, "ug_flash_attn_fwd: strides must be multiples of 8 elements and bases 16-byte aligned");
    static int nw = -1;
    if (nw < 0) { const char* e = getenv("UG_ATTN_WAVES"); nw = (e && atoi(e) == 4) ? 4 : 8; }   // 8 measured faster (841 vs 800 TFLOP/s at L = 4608)
    const int qrows = 32 * nw;
    const int nQ = (int)((Lq + qrows - 1) / qrows);
    const int64_t nwg = (int64_t)nQ * heads * batches;
    UG_REQUIRE(nwg < (1ll << 31), UG_ERR_UNSUPPORTED, "ug_flash_attn_fwd: grid too large");
    const float c = softmax_scale * 1.4426950408889634f;
#define UG_ATTN_LAUNCH(DHV, NWV, STG)                                                                                                \
    hipLaunchKernelGGL((flash_attn_kernel<DHV, NWV, STG>), dim3((unsigned)nwg), dim3(64 * NWV), 2 * 2 * KVB * 2 * DHV + (STG ? 32 * NWV * 2 * DHV : 0), (hipStream_t)stream, \
                       (const bf16_t*)q, q_row_stride, q_batch_stride, (const bf16_t*)k, k_row_stride, k_batch_stride, (const bf16_t*)v, \
                       v_row_stride, v_batch_stride, (bf16_t*)o, o_row_stride, o_batch_stride, (int)heads, (int)Lq, (int)Lkv, nQ, c)
    static int stagger = -1;
    if (stagger < 0) { const char* e = getenv("UG_ATTN_STAGGER"); stagger = (e && atoi(e) == 1) ? 1 : 0; }   // X/Y stagger measured 602 vs 842 TFLOP/s at L = 4608, dh = 128 (a lone hipcc-scheduled MFMA stream does not keep the pipe busy) -> off
    if (dh == 128) { if (nw == 4) UG_ATTN_LAUNCH(128, 4, false); else if (stagger) UG_ATTN_LAUNCH(128, 8, true); else UG_ATTN_LAUNCH(128, 8, false); }
    else           { if (nw == 4) UG_ATTN_LAUNCH(64, 4, false); else if (stagger) UG_ATTN_LAUNCH(64, 8, true); else UG_ATTN_LAUNCH(64, 8, false); }
#undef UG_ATTN_LAUNCH
    UG_CHECK_LAUNCH("ug_flash_attn_fwd");
    return UG_OK;
}
