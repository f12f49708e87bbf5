// bf16 MFMA GEMM, ONE WAVE PER SIMD: 256 x 256 output tile per workgroup, 4 waves x (128 x 128) sub-tiles, up to 512 registers per lane.
//
// Why a second structure next to gemm256_kernel (gemm.hip): the 8-wave kernel hands the matrix pipe back and forth between the two waves
// of a SIMD with 8 barrier pairs per 64-deep K-tile and keeps it busy ~64 % of the time (PMC, round 1); its fragments are re-read from LDS
// at 0.375 ds_read_b128 per MFMA. Here a single in-order wave per SIMD overlaps its own LDS reads and LDS-DMA issue with its MFMAs:
//   * per 32-deep K-step a wave issues 64 x v_mfma_f32_16x16x32_bf16 (8 x 8 accumulators of 16 x 16, 256 registers, AGPR side) against
//     16 ds_read_b128 (0.25 per MFMA) for the NEXT step's fragments (two register sets, 128 VGPRs) and 8 LDS-DMA pieces;
//   * ONE barrier per K-step (1024 MFMA cycles) instead of 4 per 1024;
//   * a 4-slot LDS ring of 32 KiB steps [A 256 x 32 | W 256 x 32], refilled 4 steps ahead (global_load_lds_dwordx4, counted vmcnt(16)):
//     step s + 4 is requested during step s into the slot whose fragments were just moved to registers.
// LDS image of a step: 64-byte rows (32 bf16); 16-byte chunk c of row r is stored at position c ^ f((r >> 2) & 3), f = {0, 3, 2, 1}: every
// 16-lane group of a ds_read_b128 (cdna guide, LDS table) then touches 16 distinct 16-byte slots of the 256-byte bank row - conflict-free -
// and the DMA writes stay lane-linear (the permutation is applied to the per-lane SOURCE chunk, guide rule 21).
// Epilogues, row maps, grouped launches, the column split and the XCD-aware tile order are the shared ones of gemm_epilogue.h.
#include "ug_common.h"
#include "gemm_epilogue.h"
#include <type_traits>

namespace {

constexpr int PK = 32;                         // K per step
constexpr int PHALF = 256 * PK * 2;            // bytes of one operand of a step (16 KiB)
constexpr int PSLOT = 2 * PHALF;               // 32 KiB

__device__ __forceinline__ int swz32(int row) { return (4 - ((row >> 2) & 3)) & 3; }

// NW = 4: one wave per SIMD, 128 x 128 per wave (512 registers). NW = 8: two waves per SIMD, 128 x 64 per wave (256 registers): the same
// self-contained step body per wave, but a wave's LDS-DMA issue (which holds its own instruction stream ~35 cycles per 1 KiB piece: measured
// by building the NW = 4 loop without its DMAs, 1340 -> 1620 TFLOP/s at 8192^3) now runs under the SIMD partner's MFMAs.
// VAR (diagnostic builds, UG_PWG_VAR): bit 0 = no DMA in the loop (WRONG results, timing only), bit 1 = waves 4-7 issue their DMAs in the
// second half of the step (stagger against waves 0-3); bit 2 = every wave issues all its LDS reads in the first half of the step.
template <int EPI, int NW, int VAR, int PRING = 4>      // PRING: LDS ring slots (32 KiB each); 5 = all 160 KiB
__global__ __launch_bounds__(64 * NW, NW / 4) void gemm_pwg_kernel(const ug_gemm_desc p, const int tiles_per_group, const int total_tiles) {
    constexpr int NT = 32 / NW;                    // 16-column blocks per wave: 8 (128 columns) or 4 (64 columns)
    constexpr int RPW = 256 / NW;                  // rows of A (and of W) a wave stages per step
    constexpr int NPC = RPW / 16;                  // DMA pieces per operand per wave per step: 4 or 2
    constexpr int NG = 2 * NT;                     // MFMA groups of 4 per step: 16 or 8
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave / (NW / 2), wc = wave % (NW / 2);
    const int64_t M = p.M, N = p.N;
    const int nM = (int)((M + 255) / 256), nN = (int)((N + 255) / 256);
    const int nsteps = (int)(p.K / PK);
    const int frow = lane & 15, fch = lane >> 4;
    const int a_off = (wr * 128 + frow) * 64 + ((fch ^ swz32(frow)) << 4);
    const int b_off = PHALF + (wc * 16 * NT + frow) * 64 + ((fch ^ swz32(frow)) << 4);
    const bool late = (VAR & 2) && wave >= NW / 2;

    for (int tile = blockIdx.x; tile < total_tiles; tile += gridDim.x) {
        const int g = tile / tiles_per_group;
        const TileCoord tc = tile_of_block(tile - g * tiles_per_group, nM, nN);
        const int64_t m0 = (int64_t)tc.tm * 256, n0 = (int64_t)tc.tn * 256;
        // ---- staging sources: wave w stages rows [RPW w, RPW (w + 1)) of A and of W, 16 rows (1 KiB) per DMA ----
        const bf16_t* asrc[NPC]; const bf16_t* bsrc[NPC];
        {
            const bf16_t* Ab = (const bf16_t*)p.A + (int64_t)g * p.a_gstride;
            const bf16_t* Wb = (const bf16_t*)p.W + (int64_t)g * p.w_gstride;
#pragma unroll
            for (int i = 0; i < NPC; ++i) {
                const int row = wave * RPW + i * 16 + (lane >> 2);
                const int c = (lane & 3) ^ swz32(row);
                int64_t am = m0 + row; if (am > M - 1) am = M - 1;
                int64_t wn = n0 + row; if (wn > N - 1) wn = N - 1;
                asrc[i] = Ab + (int64_t)rowmap32((unsigned)am, (unsigned)p.a_rpb, (unsigned)p.a_bstride) * p.lda + c * 8;
                bsrc[i] = Wb + wn * p.ldw + c * 8;
            }
        }
        // one LDS-DMA piece (1 KiB) of K-step `src_step`, into ring slot `slot_step` % 4: pieces [0, NPC) = this wave's A rows, then its W rows
        auto stage_piece = [&](int piece, int slot_step, int src_step) {
            unsigned char* slot = smem + (slot_step % PRING) * PSLOT + wave * RPW * 64;
            const int64_t ko = (int64_t)src_step * PK;
            if (piece < NPC) glds16(asrc[piece] + ko, slot + piece * 1024);
            else glds16(bsrc[piece - NPC] + ko, slot + PHALF + (piece - NPC) * 1024);
        };
        // Accumulators live in AGPRs and are touched only by the (inline-asm) MFMAs until the epilogue; the two fragment sets live in VGPRs.
        // As builtins in a branchy loop hipcc kept part of the accumulators in VGPRs, copied them through v_accvgpr_* around every step and
        // spilled 360 registers (first build); with explicit operand classes and a branch-free step body there is nothing left to decide.
        f32x4 acc[8][NT];
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if constexpr (VAR & 32) asm volatile("" : "+v"(acc[i][j])); else asm volatile("" : "+a"(acc[i][j]));
            }
        bf16x8 af[2][8], bf[2][NT];
        // The previous tile's last step left every LDS read retired (its MFMAs consumed them); its epilogue stores may still be in
        // flight, which only makes the counted waits below wait a little longer (stores retire in issue order ahead of these DMAs).
        __builtin_amdgcn_s_barrier();
        const int last = nsteps - 1;
#pragma unroll
        for (int st = 0; st < PRING; ++st)
#pragma unroll
            for (int pc = 0; pc < 2 * NPC; ++pc) stage_piece(pc, st, st < last ? st : last);      // steps past the end: clamped re-reads nobody consumes
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((PRING - 1) * 2 * NPC) : "memory");             // step 0 landed (the younger steps may fly)
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int t = 0; t < 8; ++t) af[0][t] = *(const bf16x8*)(smem + a_off + t * 1024);
#pragma unroll
        for (int t = 0; t < NT; ++t) bf[0][t] = *(const bf16x8*)(smem + b_off + t * 1024);

        auto step = [&](auto set_c, int s) __attribute__((always_inline)) {
            constexpr int SET = decltype(set_c)::value;
            // top of step s: this step's fragments (set SET) were requested during step s - 1; step s + 1's slot must have landed before
            // anybody reads it below: issued so far are steps <= s + PRING - 1, so all but the PRING - 2 youngest steps must be complete.
            if constexpr (VAR & 8) {
                // paired refill: DMAs only in odd steps (steps s + 3 and s + 4 = the two 64-byte halves of the same 128-byte lines, back to back)
                if constexpr (SET == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * NPC) : "memory");
                else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NPC) : "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"((PRING - 2) * 2 * NPC) : "memory");
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();           // every wave has its set-SET fragments in registers: slot s % 4 is free, slot (s+1) % 4 visible
            __builtin_amdgcn_sched_barrier(0);
            const unsigned char* nslot = smem + ((s + 1) % PRING) * PSLOT;       // past the end: a stale slot, read and never used
            const int src4 = s + PRING < nsteps ? s + PRING : last;     // the step refilled now: s + PRING (named src4 for the 4-slot ring)
            // per group of 4 MFMAs: fragment reads of the next step (8 A + NT W over the groups), and the 2 NPC DMA pieces of step s + 4
            auto reads = [&](int i) __attribute__((always_inline)) {          // read slot i of 0 .. NG - 1
                constexpr int NR = 8 + NT;
                constexpr int RG = (VAR & 4) ? NG / 2 : NG;             // VAR bit 2: all reads in the first half of the step
                if (i >= RG) return;
                for (int r = (i * NR) / RG; r < ((i + 1) * NR) / RG; ++r) {
                    if (r < 8) af[SET ^ 1][r] = *(const bf16x8*)(nslot + a_off + r * 1024);
                    else bf[SET ^ 1][r - 8] = *(const bf16x8*)(nslot + b_off + (r - 8) * 1024);
                }
            };
            const int src3 = s + 3 < nsteps ? s + 3 : last;
            auto dmas = [&](int i) __attribute__((always_inline)) {           // DMA slot i of 0 .. NG - 1
                if constexpr (VAR & 8) {
                    if constexpr (SET == 1) {
                        constexpr int ND = 4 * NPC;
                        for (int d = (i * ND) / NG; d < ((i + 1) * ND) / NG; ++d) {
                            if (d < 2 * NPC) stage_piece(d, s + 3, src3); else stage_piece(d - 2 * NPC, s + 4, src4);
                        }
                    }
                } else if constexpr (!(VAR & 1)) {
                    constexpr int ND = 2 * NPC;
                    for (int d = (i * ND) / NG; d < ((i + 1) * ND) / NG; ++d) stage_piece(d, s + PRING, src4);
                }
            };
#pragma unroll
            for (int i = 0; i < NG; ++i) {
#pragma unroll
                for (int q = 4 * i; q < 4 * i + 4; ++q) {
                    const int mt = q / NT, nt = q % NT;
                    if constexpr (VAR & 16) asm volatile("" :: "v"(bf[SET][nt]), "v"(af[SET][mt]));      // diagnostic: DMA + LDS reads only, no MFMA
                    else if constexpr (VAR & 32) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[mt][nt]) : "v"(bf[SET][nt]), "v"(af[SET][mt]));
                    else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[mt][nt]) : "v"(bf[SET][nt]), "v"(af[SET][mt]));
                }
                reads(i);
                if constexpr (VAR & 2) { if (late) dmas((i + NG / 2) % NG); else dmas(i); }
                else dmas(i);
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        for (int s = 0; s < nsteps; s += 2) {
            step(std::integral_constant<int, 0>{}, s);
            if (s + 1 < nsteps) step(std::integral_constant<int, 1>{}, s + 1);
        }
        // every DMA still in flight targets slots nobody reads again; drain them before the next tile re-stages the ring, and give the last
        // MFMAs' results their cycles before VALU reads them (hipcc does not see an MFMA in the asm statements: 4 passes, s_nop 15 covers it)
        asm volatile("s_waitcnt vmcnt(0)\n\ts_nop 15" ::: "memory");

        // ---- epilogue: lane holds, for row m = .. + mt * 16 + (lane & 15), columns nt * 16 + 4 (lane >> 4) .. + 3 ----
        const bf16_t* bias = p.bias ? (const bf16_t*)p.bias + (int64_t)g * p.bias_gstride : nullptr;
        const TileSplit ts = tile_split<EPI>(p, n0);
        float bv[NT][4];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int64_t n = n0 + wc * 16 * NT + nt * 16 + (lane >> 4) * 4;
            load_bias4(n < N ? bias : nullptr, n, bv[nt]);
        }
#pragma unroll
        for (int mt = 0; mt < 8; ++mt) {
            const int64_t m = m0 + wr * 128 + mt * 16 + (lane & 15);
            const bool row_ok = m < M;                         // lanes l and l ^ 16 share the row: the swaps stay paired
            RowCtx rc = row_ctx<EPI>(p, g, (unsigned)(row_ok ? m : M - 1));
            rc.coff += ts.cshift;
#pragma unroll
            for (int np = 0; np < NT / 2; ++np) {
                if (EPI == UG_EPI_BIAS_GELU && !ts.gelu)
                    epi_store_pair16<UG_EPI_BIAS>(p, rc, row_ok, n0 + wc * 16 * NT + np * 32, N, lane, acc[mt][2 * np], acc[mt][2 * np + 1], bv[2 * np], bv[2 * np + 1]);
                else
                    epi_store_pair16<EPI>(p, rc, row_ok, n0 + wc * 16 * NT + np * 32, N, lane, acc[mt][2 * np], acc[mt][2 * np + 1], bv[2 * np], bv[2 * np + 1]);
            }
        }
    }
}

template <int EPI, int NW, int VAR, int PRING = 4>
int launch_pwg_t(const ug_gemm_desc& d, hipStream_t s) {
    constexpr int PLDS = PRING * PSLOT;
    const int groups = d.groups > 0 ? d.groups : 1;
    const int64_t t256 = ((d.M + 255) / 256) * ((d.N + 255) / 256) * groups;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)gemm_pwg_kernel<EPI, NW, VAR, PRING>, hipFuncAttributeMaxDynamicSharedMemorySize, PLDS);
        attr_set = true;
    }
    static int ncu = 0;
    if (ncu == 0) {
        int dev = 0; hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ncu = prop.multiProcessorCount;
        if (ncu <= 0) ncu = 256;
    }
    const int total = (int)t256;
    dim3 grid((unsigned)(total < ncu ? total : ncu), 1, 1);
    hipLaunchKernelGGL((gemm_pwg_kernel<EPI, NW, VAR, PRING>), grid, dim3(64 * NW), PLDS, s, d, (int)(t256 / groups), total);
    UG_CHECK_LAUNCH("ug_gemm_bf16(pwg)");
    return UG_OK;
}

}  // namespace

// Called by gemm.hip's dispatcher for shapes the 256^2 tiles fill (no LoRA segment, 16-byte epilogue granularity, not UG_EPI_F32).
// mode (UG_GEMM_PWG): 1 = one wave per SIMD, 2 = two waves per SIMD.
int ug_gemm_launch_pwg(const ug_gemm_desc& d, hipStream_t s) {
    const int mode = ug_env_int("UG_GEMM_PWG", 0);
    const int var = ug_env_int("UG_PWG_VAR", 0);
#define UG_PWG_CASE(E)                                                                            \
    case E:                                                                                       \
        if (mode == 2) return var == 2 ? launch_pwg_t<E, 8, 2>(d, s) : launch_pwg_t<E, 8, 0>(d, s);   \
        return launch_pwg_t<E, 4, 0>(d, s);
    switch (d.epilogue) {
        case UG_EPI_BIAS:
            if (mode == 1 && var >= 100) {        // ring-depth probes: var = 100 * slots + (16: DMA only | 0: full kernel)
                if (var == 216) return launch_pwg_t<UG_EPI_BIAS, 4, 16, 2>(d, s);
                if (var == 316) return launch_pwg_t<UG_EPI_BIAS, 4, 16, 3>(d, s);
                if (var == 516) return launch_pwg_t<UG_EPI_BIAS, 4, 16, 5>(d, s);
                if (var == 300) return launch_pwg_t<UG_EPI_BIAS, 4, 0, 3>(d, s);
                if (var == 500) return launch_pwg_t<UG_EPI_BIAS, 4, 0, 5>(d, s);
            }
            if (mode == 2 && var == 32) return launch_pwg_t<UG_EPI_BIAS, 8, 32>(d, s);
            if (mode == 2 && var == 33) return launch_pwg_t<UG_EPI_BIAS, 8, 33>(d, s);
            if (mode == 2) return var == 1 ? launch_pwg_t<UG_EPI_BIAS, 8, 1>(d, s) : var == 2 ? launch_pwg_t<UG_EPI_BIAS, 8, 2>(d, s) :
                                  var == 4 ? launch_pwg_t<UG_EPI_BIAS, 8, 4>(d, s) : var == 6 ? launch_pwg_t<UG_EPI_BIAS, 8, 6>(d, s) : launch_pwg_t<UG_EPI_BIAS, 8, 0>(d, s);
            return var == 1 ? launch_pwg_t<UG_EPI_BIAS, 4, 1>(d, s) : var == 4 ? launch_pwg_t<UG_EPI_BIAS, 4, 4>(d, s) : var == 5 ? launch_pwg_t<UG_EPI_BIAS, 4, 5>(d, s) :
                   var == 8 ? launch_pwg_t<UG_EPI_BIAS, 4, 8>(d, s) : var == 16 ? launch_pwg_t<UG_EPI_BIAS, 4, 16>(d, s) : launch_pwg_t<UG_EPI_BIAS, 4, 0>(d, s);
        UG_PWG_CASE(UG_EPI_BIAS_GELU)
        UG_PWG_CASE(UG_EPI_RES_GATE)
        UG_PWG_CASE(UG_EPI_RES_SCALE)
        default: UG_FAIL(UG_ERR_UNSUPPORTED, "ug_gemm_bf16(pwg): epilogue %d", d.epilogue);
    }
#undef UG_PWG_CASE
}
