// CoMoE routing kernels: top-1 gate, capacity selection with Random Token Selection, index-based dispatch fused with
// the expert-modulation prologue, and the probability-weighted combine fused with the CoMoE residual sums.
//
// The reference (src/UniGenUtils.py:74-191 on top of deepspeed 0.16.5 sharded_moe.top1gating) materialises S x E x C
// one-hot tensors and dispatches/combines with dense einsums (:140, :183). Here routing is a per-token (expert, slot)
// pair and a per-slot token index: dispatch is a row gather, combine a row scatter, zero FLOPs.
#include "ug_common.h"

namespace {

constexpr int GATE_MAXE = 16;

// one wave per token: logits[e] = sum_d bf16(x+c)[d] * wg[e][d] in fp32 (TopKGate: F.linear(input.float(), wg.float()))
// TOP2 (deepspeed top2gating): idx[S + s] = the arg-max of logits + noise over the experts other than the first choice (the Gumbel-max draw of
// `top2_2nd_expert_sampling`; noise == nullptr: the plain second-largest logit).
template <typename T, bool TOP2>
__global__ __launch_bounds__(256) void moe_gate_kernel(const T* __restrict__ x, const T* __restrict__ c, int64_t ld,
                                                       const T* __restrict__ wg, int64_t S, int D, int E,
                                                       const float* __restrict__ noise, float* __restrict__ gates, int32_t* __restrict__ idx) {
    using EL = ElemT<T>;
    const int lane = threadIdx.x & 63;
    const int64_t s = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (s >= S) return;
    float acc[GATE_MAXE];
#pragma unroll
    for (int e = 0; e < GATE_MAXE; ++e) acc[e] = 0.f;
    const int nchunk = D >> 3;
    for (int ch = lane; ch < nchunk; ch += 64) {
        float a[8], b[8];
        EL::load8(x + s * ld + ch * 8, a);
        EL::load8(c + s * ld + ch * 8, b);
#pragma unroll
        for (int i = 0; i < 8; ++i) a[i] = EL::rnd(a[i] + b[i]);     // choice_expert_input = hidden + condition (bf16 add)
#pragma unroll
        for (int e = 0; e < GATE_MAXE; ++e) {
            if (e < E) {
                float w[8];
                EL::load8(wg + (int64_t)e * D + ch * 8, w);
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[e] += a[i] * w[i];
            }
        }
    }
    float mx = -INFINITY;
    int best = 0;
#pragma unroll
    for (int e = 0; e < GATE_MAXE; ++e) {
        if (e < E) {
            acc[e] = wave_sum(acc[e]);
            if (acc[e] > mx) { mx = acc[e]; best = e; }     // first maximum wins, as torch.argmax
        }
    }
    int second = 0;
    if constexpr (TOP2) {
        float m2 = -INFINITY;
        second = best == 0 ? 1 : 0;
#pragma unroll
        for (int e = 0; e < GATE_MAXE; ++e) {
            if (e < E && e != best) {                       // logits.masked_fill(mask1, -inf) after logits += gumbel
                const float v = acc[e] + (noise ? noise[s * E + e] : 0.f);
                if (v > m2) { m2 = v; second = e; }
            }
        }
    }
    float den = 0.f;
#pragma unroll
    for (int e = 0; e < GATE_MAXE; ++e)
        if (e < E) { acc[e] = expf(acc[e] - mx); den += acc[e]; }
    if (lane == 0) {
        for (int e = 0; e < E; ++e) gates[s * E + e] = acc[e] / den;
        idx[s] = best;
        if constexpr (TOP2) idx[S + s] = second;
    }
}

// TOPK (deepspeed topkgating, k > 2): the K largest LOGITS per token (descending; the lower expert index first among equals), the softmax
// probabilities, and the logits themselves - topkgating's capacity rule ranks logits, not probabilities.
template <typename T>
__global__ __launch_bounds__(256) void moe_gate_topk_kernel(const T* __restrict__ x, const T* __restrict__ c, int64_t ld,
                                                            const T* __restrict__ wg, int64_t S, int D, int E, int K,
                                                            float* __restrict__ gates, float* __restrict__ logits, int32_t* __restrict__ idx) {
    using EL = ElemT<T>;
    const int lane = threadIdx.x & 63;
    const int64_t s = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (s >= S) return;
    float acc[GATE_MAXE];
#pragma unroll
    for (int e = 0; e < GATE_MAXE; ++e) acc[e] = 0.f;
    const int nchunk = D >> 3;
    for (int ch = lane; ch < nchunk; ch += 64) {
        float a[8], b[8];
        EL::load8(x + s * ld + ch * 8, a);
        EL::load8(c + s * ld + ch * 8, b);
#pragma unroll
        for (int i = 0; i < 8; ++i) a[i] = EL::rnd(a[i] + b[i]);
#pragma unroll
        for (int e = 0; e < GATE_MAXE; ++e) {
            if (e < E) {
                float w[8];
                EL::load8(wg + (int64_t)e * D + ch * 8, w);
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[e] += a[i] * w[i];
            }
        }
    }
    float mx = -INFINITY;
#pragma unroll
    for (int e = 0; e < GATE_MAXE; ++e)
        if (e < E) { acc[e] = wave_sum(acc[e]); mx = fmaxf(mx, acc[e]); }
    if (lane == 0) {
        unsigned taken = 0;
        for (int k = 0; k < K; ++k) {                       // K rounds of "largest not yet taken"
            float m = -INFINITY;
            int best = 0;
            bool any = false;
#pragma unroll
            for (int e = 0; e < GATE_MAXE; ++e)
                if (e < E && !((taken >> e) & 1u) && (!any || acc[e] > m)) { m = acc[e]; best = e; any = true; }
            taken |= 1u << best;
            idx[(int64_t)k * S + s] = best;
        }
        float den = 0.f, p[GATE_MAXE];
#pragma unroll
        for (int e = 0; e < GATE_MAXE; ++e)
            if (e < E) { p[e] = expf(acc[e] - mx); den += p[e]; }
#pragma unroll
        for (int e = 0; e < GATE_MAXE; ++e)
            if (e < E) { gates[s * E + e] = p[e] / den; logits[s * E + e] = acc[e]; }
    }
}

// block-wide exclusive scan of one flag per thread (1024 threads); returns exclusive prefix, *total = block total
__device__ __forceinline__ int block_excl_scan(int flag, int* wsum /*[17]*/, int* total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long bal = __ballot(flag);
    const int within = __popcll(bal & ((1ull << lane) - 1ull));
    __syncthreads();                       // protect wsum reuse
    if (lane == 0) wsum[wave] = __popcll(bal);
    __syncthreads();
    int base = 0, tot = 0;
    for (int w = 0; w < 16; ++w) { const int v = wsum[w]; if (w < wave) base += v; tot += v; }
    *total = tot;
    return base + within;
}

// one block (1024 threads) per expert
__global__ __launch_bounds__(1024) void moe_capacity_kernel(const int32_t* __restrict__ idx, const float* __restrict__ uniform,
                                                            int S, int E, int capacity, int32_t* __restrict__ slot,
                                                            int32_t* __restrict__ token_of_slot, int64_t* __restrict__ exp_counts) {
    __shared__ int hist[256];
    __shared__ int wsum[17];
    __shared__ unsigned sh_prefix;
    __shared__ int sh_k;
    const int e = blockIdx.x;
    const int tid = threadIdx.x;
    // 1. count tokens routed to e
    int cnt = 0;
    for (int s = tid; s < S; s += 1024) cnt += (idx[s] == e);
    {
        const int lane = tid & 63, wave = tid >> 6;
        for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
        if (lane == 0) wsum[wave] = cnt;
        __syncthreads();
        cnt = 0;
        for (int w = 0; w < 16; ++w) cnt += wsum[w];
        __syncthreads();
    }
    const int n_e = cnt;
    if (tid == 0) exp_counts[e] = (int64_t)n_e;
    // 2. threshold = capacity-th largest uniform among this expert's tokens (radix select, MSB first)
    unsigned T = 0; int k_eq = 0;
    const bool drop = n_e > capacity;
    if (drop) {
        unsigned prefix = 0, mask = 0; int k = capacity;
        for (int shift = 24; shift >= 0; shift -= 8) {
            for (int i = tid; i < 256; i += 1024) hist[i] = 0;
            __syncthreads();
            for (int s = tid; s < S; s += 1024) {
                if (idx[s] == e) {
                    const unsigned key = __float_as_uint(uniform[(int64_t)s * E + e]);
                    if ((key & mask) == prefix) atomicAdd(&hist[(key >> shift) & 255], 1);
                }
            }
            __syncthreads();
            if (tid == 0) {
                int kk = k, d = 255;
                for (; d > 0; --d) { if (hist[d] >= kk) break; kk -= hist[d]; }
                sh_prefix = prefix | ((unsigned)d << shift);
                sh_k = kk;
            }
            __syncthreads();
            prefix = sh_prefix; k = sh_k; mask |= (255u << shift);
            __syncthreads();
        }
        T = prefix; k_eq = k;   // keep every key > T and the first k_eq tokens with key == T
    }
    // 3. slots = rank among kept tokens in token order (cumsum(mask1) - 1 in top1gating)
    int base_eq = 0, base_kept = 0;
    for (int s0 = 0; s0 < S; s0 += 1024) {
        const int s = s0 + tid;
        const bool mine = s < S && idx[s] == e;
        bool kept = mine;
        if (drop) {
            unsigned key = 0;
            if (mine) key = __float_as_uint(uniform[(int64_t)s * E + e]);
            const int feq = mine && key == T;
            int tot_eq;
            const int r_eq = block_excl_scan(feq, wsum, &tot_eq);
            kept = mine && (key > T || (feq && base_eq + r_eq < k_eq));
            base_eq += tot_eq;
        }
        int tot_k;
        const int rk = block_excl_scan(kept ? 1 : 0, wsum, &tot_k);
        if (mine) {
            const int sl = kept ? base_kept + rk : -1;
            slot[s] = sl;
            if (kept) token_of_slot[(int64_t)e * capacity + sl] = s;
        }
        base_kept += tot_k;
    }
    for (int c = base_kept + tid; c < capacity; c += 1024) token_of_slot[(int64_t)e * capacity + c] = -1;
}

// deepspeed top2gating's capacity rule (no random token selection): first choices take an expert's slots in token order, second choices follow
// behind ALL its first choices (locations2 += sum(mask1)); whatever lands at or beyond `capacity` is dropped. One block (1024 threads) per
// expert. idx / slot are [2][S] (choice-major); exp_counts = first + second choices before the drop (torch.sum(mask1 + mask2, dim=0)).
__global__ __launch_bounds__(1024) void moe_capacity_top2_kernel(const int32_t* __restrict__ idx, int S, int capacity, int32_t* __restrict__ slot,
                                                                 int32_t* __restrict__ token_of_slot, int64_t* __restrict__ exp_counts) {
    __shared__ int wsum[17];
    const int e = blockIdx.x;
    const int tid = threadIdx.x;
    int base = 0;
    for (int k = 0; k < 2; ++k) {
        const int32_t* ik = idx + (int64_t)k * S;
        int32_t* sk = slot + (int64_t)k * S;
        for (int s0 = 0; s0 < S; s0 += 1024) {
            const int s = s0 + tid;
            const bool mine = s < S && ik[s] == e;
            int tot;
            const int r = block_excl_scan(mine ? 1 : 0, wsum, &tot);
            if (mine) {
                const int loc = base + r;
                const bool kept = loc < capacity;
                sk[s] = kept ? loc : -1;
                if (kept) token_of_slot[(int64_t)e * capacity + loc] = s;
            }
            base += tot;
        }
    }
    if (tid == 0) exp_counts[e] = (int64_t)base;
    for (int c = base + tid; c < capacity; c += 1024) token_of_slot[(int64_t)e * capacity + c] = -1;
}

// combine weights of top2gating: the two gate probabilities of a token (zero for a dropped choice) normalised by their sum clamped at
// finfo(float32).eps. weights [2][S].
__global__ __launch_bounds__(256) void moe_weights_top2_kernel(const float* __restrict__ gates, const int32_t* __restrict__ idx,
                                                               const int32_t* __restrict__ slot, int S, int E, float* __restrict__ weights) {
    const int s = blockIdx.x * 256 + threadIdx.x;
    if (s >= S) return;
    const float g1 = slot[s] >= 0 ? gates[(int64_t)s * E + idx[s]] : 0.f;
    const float g2 = slot[S + s] >= 0 ? gates[(int64_t)s * E + idx[S + s]] : 0.f;
    const float den = fmaxf(g1 + g2, 1.1920928955078125e-07f);
    weights[s] = g1 / den;
    weights[S + s] = g2 / den;
}

// order-preserving map of a float onto unsigned (larger float <-> larger key), for the radix select below
__device__ __forceinline__ unsigned ug_fkey(float v) {
    const unsigned u = __float_as_uint(v);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// deepspeed topkgating's capacity rule (drop_policy "probs"): an expert keeps the `capacity` largest entries of its column of
// topk_masked_gates = (the token's LOGIT if the expert is one of its K choices, else 0), i.e. torch.topk(., k = capacity, dim = 0) over ALL S
// tokens - non-choosers compete with their zeros - and a choice survives if it is among them. Ties at the threshold: token order (torch's
// choice among exact ties is unspecified). Slots = rank among the kept tokens in token order (cumsum(mask) - 1). One block (1024 threads) per
// expert; idx / slot are [K][S]; exp_counts = choosers before the drop.
__global__ __launch_bounds__(1024) void moe_capacity_topk_kernel(const int32_t* __restrict__ idx, const float* __restrict__ logits, int S, int E, int K,
                                                                 int capacity, int32_t* __restrict__ slot, int32_t* __restrict__ token_of_slot,
                                                                 int64_t* __restrict__ exp_counts) {
    __shared__ int hist[256];
    __shared__ int wsum[17];
    __shared__ unsigned sh_prefix;
    __shared__ int sh_k;
    const int e = blockIdx.x;
    const int tid = threadIdx.x;
    auto choice_of = [&](int s) {                   // which of the token's K choices is expert e (-1: none)
        int kk = -1;
        for (int k = 0; k < K; ++k) if (idx[(int64_t)k * S + s] == e) kk = k;
        return kk;
    };
    auto key_of = [&](int s, int kk) { return ug_fkey(kk >= 0 ? logits[(int64_t)s * E + e] : 0.f); };
    int cnt = 0;
    for (int s = tid; s < S; s += 1024) cnt += (choice_of(s) >= 0);
    {
        const int lane = tid & 63, wave = tid >> 6;
        for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
        if (lane == 0) wsum[wave] = cnt;
        __syncthreads();
        cnt = 0;
        for (int w = 0; w < 16; ++w) cnt += wsum[w];
        __syncthreads();
    }
    if (tid == 0) exp_counts[e] = (int64_t)cnt;
    // threshold = capacity-th largest key of the whole column (radix select, MSB first); capacity >= S keeps every chooser
    unsigned T = 0; int k_eq = 0;
    const bool drop = capacity < S;
    if (drop) {
        unsigned prefix = 0, mask = 0; int k = capacity;
        for (int shift = 24; shift >= 0; shift -= 8) {
            for (int i = tid; i < 256; i += 1024) hist[i] = 0;
            __syncthreads();
            for (int s = tid; s < S; s += 1024) {
                const unsigned key = key_of(s, choice_of(s));
                if ((key & mask) == prefix) atomicAdd(&hist[(key >> shift) & 255], 1);
            }
            __syncthreads();
            if (tid == 0) {
                int kk = k, d = 255;
                for (; d > 0; --d) { if (hist[d] >= kk) break; kk -= hist[d]; }
                sh_prefix = prefix | ((unsigned)d << shift);
                sh_k = kk;
            }
            __syncthreads();
            prefix = sh_prefix; k = sh_k; mask |= (255u << shift);
            __syncthreads();
        }
        T = prefix; k_eq = k;       // keep every key > T and the first k_eq column entries (choosers or not) with key == T, in token order
    }
    int base_eq = 0, base_kept = 0;
    for (int s0 = 0; s0 < S; s0 += 1024) {
        const int s = s0 + tid;
        const int kk = s < S ? choice_of(s) : -1;
        const bool mine = kk >= 0;
        bool kept = mine;
        if (drop) {
            const unsigned key = s < S ? key_of(s, kk) : 0u;
            const int feq = s < S && key == T;
            int tot_eq;
            const int r_eq = block_excl_scan(feq, wsum, &tot_eq);
            kept = mine && (key > T || (feq && base_eq + r_eq < k_eq));
            base_eq += tot_eq;
        }
        int tot_k;
        const int rk = block_excl_scan(kept ? 1 : 0, wsum, &tot_k);
        if (mine) {
            const int sl = kept ? base_kept + rk : -1;
            slot[(int64_t)kk * S + s] = sl;
            if (kept) token_of_slot[(int64_t)e * capacity + sl] = s;
        }
        base_kept += tot_k;
    }
    for (int c = base_kept + tid; c < capacity; c += 1024) token_of_slot[(int64_t)e * capacity + c] = -1;
}

// combine weights of topkgating: a token's kept gate probabilities over their sum clamped at finfo(float32).eps. weights [K][S].
__global__ __launch_bounds__(256) void moe_weights_topk_kernel(const float* __restrict__ gates, const int32_t* __restrict__ idx,
                                                               const int32_t* __restrict__ slot, int S, int E, int K, float* __restrict__ weights) {
    const int s = blockIdx.x * 256 + threadIdx.x;
    if (s >= S) return;
    float g[GATE_MAXE], den = 0.f;
    for (int k = 0; k < K; ++k) {
        g[k] = slot[(int64_t)k * S + s] >= 0 ? gates[(int64_t)s * E + idx[(int64_t)k * S + s]] : 0.f;
        den += g[k];
    }
    den = fmaxf(den, 1.1920928955078125e-07f);
    for (int k = 0; k < K; ++k) weights[(int64_t)k * S + s] = g[k] / den;
}

// l_aux = E * sum_e mean_s(gates[s][e]) * (n_e / S), n_e = tokens whose FIRST choice is e: exp_counts[e] (top-1: the same thing) or, when
// idx1 is given (top-2: exp_counts holds both choices), counted from idx1; single block, fixed summation order
__global__ __launch_bounds__(1024) void moe_laux_kernel(const float* __restrict__ gates, const int64_t* __restrict__ exp_counts, const int32_t* __restrict__ idx1,
                                                        int S, int E, float* __restrict__ l_aux, float scale) {
    __shared__ float part[16];
    __shared__ int ipart[16];
    __shared__ float terms[GATE_MAXE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int e = 0; e < E; ++e) {
        float a = 0.f;
        int n = 0;
        for (int s = tid; s < S; s += 1024) { a += gates[(int64_t)s * E + e]; if (idx1) n += (idx1[s] == e); }
        a = wave_sum(a);
        for (int o = 32; o > 0; o >>= 1) n += __shfl_xor(n, o, 64);
        if (lane == 0) { part[wave] = a; ipart[wave] = n; }
        __syncthreads();
        if (tid == 0) {
            float t = 0.f;
            int nt = 0;
            for (int w = 0; w < 16; ++w) { t += part[w]; nt += ipart[w]; }
            terms[e] = (t / (float)S) * ((idx1 ? (float)nt : (float)exp_counts[e]) / (float)S);
        }
        __syncthreads();
    }
    if (tid == 0) {
        float t = 0.f;
        for (int e = 0; e < E; ++e) t += terms[e];
        *l_aux = t * scale;          // top-1 / top-2: E; topkgating: mean(me * ce) * E * E / k = sum * E / k with ce over all k choices
    }
}

// one wave per (expert, slot) row
template <typename T>
__global__ __launch_bounds__(256) void moe_dispatch_kernel(const T* __restrict__ x, int64_t ldx, const T* __restrict__ add,
                                                           const T* __restrict__ mod, int64_t mod_estride, int64_t mod_bstride,
                                                           const int32_t* __restrict__ token_of_slot,
                                                           int64_t nslots, int capacity, int tokens_per_sample, int D,
                                                           T* __restrict__ out) {
    using EL = ElemT<T>;
    const int lane = threadIdx.x & 63;
    const int64_t sl = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (sl >= nslots) return;
    const int tok = token_of_slot[sl];
    T* orow = out + sl * D;
    const int nchunk = D >> 3;
    if (tok < 0) {
        for (int ch = lane; ch < nchunk; ch += 64) EL::zero8(orow + ch * 8);
        return;
    }
    const int e = (int)(sl / capacity);
    const int b = tok / tokens_per_sample;
    const T* xr = x + (int64_t)tok * ldx;
    const T* mr = mod ? mod + (int64_t)e * mod_estride + (int64_t)b * mod_bstride : nullptr;
    const T* ar = add ? add + sl * D : nullptr;
    for (int ch = lane; ch < nchunk; ch += 64) {
        float a[8], m[8];
        EL::load8(xr + ch * 8, a);
        if (mr) EL::load8(mr + ch * 8, m);
        if (ar) {
            float t[8];
            EL::load8(ar + ch * 8, t);
#pragma unroll
            for (int i = 0; i < 8; ++i) a[i] = EL::rnd(a[i] + t[i]);
        }
        if (mr) {
#pragma unroll
            for (int i = 0; i < 8; ++i) a[i] *= m[i];
        }
        EL::store8(orow + ch * 8, a);
    }
}

// one wave per token
template <typename T>
__global__ __launch_bounds__(256) void moe_combine_kernel(const T* __restrict__ yh, const T* __restrict__ yc,
                                                          const float* __restrict__ gates, const int32_t* __restrict__ idx,
                                                          const int32_t* __restrict__ slot, int E, int capacity,
                                                          const T* __restrict__ xs, const T* __restrict__ cs, int64_t ld_s,
                                                          int64_t s_rpb, int64_t s_bstride,
                                                          T* __restrict__ out, int64_t ldo, int64_t S, int D, int accumulate) {
    using EL = ElemT<T>;
    const int lane = threadIdx.x & 63;
    const int64_t s = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (s >= S) return;
    const int e = idx[s];
    const int sl = slot[s];
    const float p = EL::rnd(gates[s * E + e]);             // combine_weights.type_as(input)
    const int64_t srow = ug_rowmap(s, s_rpb, s_bstride);   // token s of the shared-expert streams (they live in a [B][2N] buffer)
    const int64_t yrow = ((int64_t)e * capacity + (sl < 0 ? 0 : sl)) * D;
    const int nchunk = D >> 3;
    for (int ch = lane; ch < nchunk; ch += 64) {
        float h[8], c[8], o[8];
        if (sl >= 0) {
            EL::load8(yh + yrow + ch * 8, h);
            EL::load8(yc + yrow + ch * 8, c);
#pragma unroll
            for (int i = 0; i < 8; ++i) { h[i] = EL::rnd(p * h[i]); c[i] = EL::rnd(p * c[i]); }
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) { h[i] = 0.f; c[i] = 0.f; }
        }
        if (xs) {
            float a[8], b[8];
            EL::load8(xs + srow * ld_s + ch * 8, a);
            EL::load8(cs + srow * ld_s + ch * 8, b);
#pragma unroll
            for (int i = 0; i < 8; ++i) o[i] = EL::rnd(a[i] + h[i]) + EL::rnd(b[i] + c[i]);
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) o[i] = h[i] + c[i];
        }
        if (accumulate) {
            float prev[8];
            EL::load8(out + s * ldo + ch * 8, prev);
#pragma unroll
            for (int i = 0; i < 8; ++i) o[i] = prev[i] + EL::rnd(o[i]);
        }
        EL::store8(out + s * ldo + ch * 8, o);
    }
}

// top-k combine (k = 2: deepspeed top2gating): einsum("sec,ecm->sm") over a token's K (expert, slot) pairs in the activation dtype =
// ONE rounding of the fp32 sum of the products  weight_k.type_as(y) * y[e_k][slot_k]; the residual sums around it as in moe_combine_kernel.
// weights fp32 / idx / slot: [K][kstride].
template <typename T>
__global__ __launch_bounds__(256) void moe_combine_topk_kernel(const T* __restrict__ yh, const T* __restrict__ yc,
                                                               const float* __restrict__ weights, const int32_t* __restrict__ idx,
                                                               const int32_t* __restrict__ slot, int K, int64_t kstride, int capacity,
                                                               const T* __restrict__ xs, const T* __restrict__ cs, int64_t ld_s,
                                                               int64_t s_rpb, int64_t s_bstride,
                                                               T* __restrict__ out, int64_t ldo, int64_t S, int D, int accumulate) {
    using EL = ElemT<T>;
    const int lane = threadIdx.x & 63;
    const int64_t s = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (s >= S) return;
    const int64_t srow = ug_rowmap(s, s_rpb, s_bstride);
    const int nchunk = D >> 3;
    for (int ch = lane; ch < nchunk; ch += 64) {
        float h[8], c[8], o[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) { h[i] = 0.f; c[i] = 0.f; }
        for (int k = 0; k < K; ++k) {
            const int sl = slot[k * kstride + s];
            if (sl < 0) continue;
            const float p = EL::rnd(weights[k * kstride + s]);
            const int64_t yrow = ((int64_t)idx[k * kstride + s] * capacity + sl) * D;
            float a[8], b[8];
            EL::load8(yh + yrow + ch * 8, a);
            EL::load8(yc + yrow + ch * 8, b);
#pragma unroll
            for (int i = 0; i < 8; ++i) { h[i] = fmaf(p, a[i], h[i]); c[i] = fmaf(p, b[i], c[i]); }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) { h[i] = EL::rnd(h[i]); c[i] = EL::rnd(c[i]); }
        if (xs) {
            float a[8], b[8];
            EL::load8(xs + srow * ld_s + ch * 8, a);
            EL::load8(cs + srow * ld_s + ch * 8, b);
#pragma unroll
            for (int i = 0; i < 8; ++i) o[i] = EL::rnd(a[i] + h[i]) + EL::rnd(b[i] + c[i]);
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) o[i] = h[i] + c[i];
        }
        if (accumulate) {
            float prev[8];
            EL::load8(out + s * ldo + ch * 8, prev);
#pragma unroll
            for (int i = 0; i < 8; ++i) o[i] = prev[i] + EL::rnd(o[i]);
        }
        EL::store8(out + s * ldo + ch * 8, o);
    }
}

}  // namespace

namespace {

template <typename T>
int moe_gate_impl(const void* x, const void* c, int64_t ld, const void* wg, int64_t S, int64_t D, int32_t E, const float* noise, bool top2, float* gates,
                  int32_t* idx, ug_stream_t stream) {
    if (S == 0) return UG_OK;
    UG_REQUIRE(x && c && wg && gates && idx && S > 0 && D > 0, UG_ERR_BAD_SHAPE, "ug_moe_gate_top1: bad arguments");
    UG_REQUIRE(E >= (top2 ? 2 : 1) && E <= GATE_MAXE, UG_ERR_UNSUPPORTED, "ug_moe_gate_top%d: E=%d not in [%d,%d]", top2 ? 2 : 1, E, top2 ? 2 : 1, GATE_MAXE);
    UG_REQUIRE(D % 8 == 0 && ld % 8 == 0 && ug_aligned(x, 16) && ug_aligned(c, 16) && ug_aligned(wg, 16), UG_ERR_BAD_ALIGN,
               "ug_moe_gate_top1: 16-byte alignment required");
    if (top2)
        hipLaunchKernelGGL((moe_gate_kernel<T, true>), dim3((unsigned)((S + 3) / 4)), dim3(256), 0, (hipStream_t)stream, (const T*)x,
                           (const T*)c, ld, (const T*)wg, S, (int)D, (int)E, noise, gates, idx);
    else
        hipLaunchKernelGGL((moe_gate_kernel<T, false>), dim3((unsigned)((S + 3) / 4)), dim3(256), 0, (hipStream_t)stream, (const T*)x,
                           (const T*)c, ld, (const T*)wg, S, (int)D, (int)E, (const float*)nullptr, gates, idx);
    UG_CHECK_LAUNCH("ug_moe_gate");
    return UG_OK;
}

template <typename T>
int moe_dispatch_impl(const void* x, int64_t ldx, const void* add, const void* mod, int64_t mod_estride, int64_t mod_bstride,
                      const int32_t* token_of_slot, int32_t E, int64_t capacity, int64_t tokens_per_sample, int64_t D, void* out, ug_stream_t stream) {
    UG_REQUIRE(x && token_of_slot && out && E > 0 && capacity > 0 && tokens_per_sample > 0, UG_ERR_BAD_SHAPE,
               "ug_moe_dispatch_modulate: bad arguments");
    UG_REQUIRE(D % 8 == 0 && ldx % 8 == 0 && ug_aligned(x, 16) && (!mod || (ug_aligned(mod, 16) && mod_estride % 8 == 0 && mod_bstride % 8 == 0)) &&
               ug_aligned(out, 16) && (!add || ug_aligned(add, 16)),
               UG_ERR_BAD_ALIGN, "ug_moe_dispatch_modulate: 16-byte alignment required");
    const int64_t nslots = (int64_t)E * capacity;
    hipLaunchKernelGGL(moe_dispatch_kernel<T>, dim3((unsigned)((nslots + 3) / 4)), dim3(256), 0, (hipStream_t)stream, (const T*)x, ldx,
                       (const T*)add, (const T*)mod, mod_estride, mod_bstride, token_of_slot, nslots, (int)capacity, (int)tokens_per_sample,
                       (int)D, (T*)out);
    UG_CHECK_LAUNCH("ug_moe_dispatch_modulate");
    return UG_OK;
}

template <typename T>
int moe_combine_impl(const void* yh, const void* yc, const float* gates, const int32_t* idx, const int32_t* slot, int32_t E,
                     int64_t capacity, const void* xs, const void* cs, int64_t ld_s, int64_t s_rpb, int64_t s_bstride, void* out, int64_t ldo, int64_t S,
                     int64_t D, int32_t accumulate, ug_stream_t stream) {
    if (S == 0) return UG_OK;
    UG_REQUIRE(yh && yc && gates && idx && slot && out && E > 0 && capacity > 0, UG_ERR_BAD_SHAPE, "ug_moe_combine: bad arguments");
    UG_REQUIRE((xs == nullptr) == (cs == nullptr), UG_ERR_BAD_SHAPE, "ug_moe_combine: xs and cs must both be given or both NULL");
    UG_REQUIRE(s_rpb >= 0 && s_bstride >= 0, UG_ERR_BAD_SHAPE, "ug_moe_combine: bad row map");
    UG_REQUIRE(D % 8 == 0 && ldo % 8 == 0 && (!xs || ld_s % 8 == 0) && ug_aligned(yh, 16) && ug_aligned(yc, 16) && ug_aligned(out, 16) &&
               (!xs || (ug_aligned(xs, 16) && ug_aligned(cs, 16))), UG_ERR_BAD_ALIGN, "ug_moe_combine: 16-byte alignment required");
    hipLaunchKernelGGL(moe_combine_kernel<T>, dim3((unsigned)((S + 3) / 4)), dim3(256), 0, (hipStream_t)stream, (const T*)yh,
                       (const T*)yc, gates, idx, slot, (int)E, (int)capacity, (const T*)xs, (const T*)cs, ld_s, s_rpb, s_bstride,
                       (T*)out, ldo, S, (int)D, (int)accumulate);
    UG_CHECK_LAUNCH("ug_moe_combine");
    return UG_OK;
}

template <typename T>
int moe_combine_topk_impl(const void* yh, const void* yc, const float* weights, const int32_t* idx, const int32_t* slot, int32_t K, int64_t kstride,
                          int32_t E, int64_t capacity, const void* xs, const void* cs, int64_t ld_s, int64_t s_rpb, int64_t s_bstride, void* out,
                          int64_t ldo, int64_t S, int64_t D, int32_t accumulate, ug_stream_t stream) {
    if (S == 0) return UG_OK;
    UG_REQUIRE(yh && yc && weights && idx && slot && out && E > 0 && capacity > 0 && K >= 1 && K <= GATE_MAXE && kstride >= S, UG_ERR_BAD_SHAPE,
               "ug_moe_combine_topk: bad arguments");
    UG_REQUIRE((xs == nullptr) == (cs == nullptr), UG_ERR_BAD_SHAPE, "ug_moe_combine_topk: xs and cs must both be given or both NULL");
    UG_REQUIRE(s_rpb >= 0 && s_bstride >= 0, UG_ERR_BAD_SHAPE, "ug_moe_combine_topk: bad row map");
    UG_REQUIRE(D % 8 == 0 && ldo % 8 == 0 && (!xs || ld_s % 8 == 0) && ug_aligned(yh, 16) && ug_aligned(yc, 16) && ug_aligned(out, 16) &&
               (!xs || (ug_aligned(xs, 16) && ug_aligned(cs, 16))), UG_ERR_BAD_ALIGN, "ug_moe_combine_topk: 16-byte alignment required");
    hipLaunchKernelGGL(moe_combine_topk_kernel<T>, dim3((unsigned)((S + 3) / 4)), dim3(256), 0, (hipStream_t)stream, (const T*)yh,
                       (const T*)yc, weights, idx, slot, (int)K, kstride, (int)capacity, (const T*)xs, (const T*)cs, ld_s, s_rpb, s_bstride,
                       (T*)out, ldo, S, (int)D, (int)accumulate);
    UG_CHECK_LAUNCH("ug_moe_combine_topk");
    return UG_OK;
}

}  // namespace

extern "C" int ug_moe_gate_top1(const void* x, const void* c, int64_t ld, const void* wg, int64_t S, int64_t D, int32_t E, float* gates, int32_t* idx, ug_stream_t s) {
    return moe_gate_impl<bf16_t>(x, c, ld, wg, S, D, E, nullptr, false, gates, idx, s);
}
extern "C" int ug_moe_gate_top1_f32(const void* x, const void* c, int64_t ld, const void* wg, int64_t S, int64_t D, int32_t E, float* gates, int32_t* idx, ug_stream_t s) {
    return moe_gate_impl<float>(x, c, ld, wg, S, D, E, nullptr, false, gates, idx, s);
}
extern "C" int ug_moe_gate_top2(const void* x, const void* c, int64_t ld, const void* wg, int64_t S, int64_t D, int32_t E, const float* noise, float* gates,
                                int32_t* idx, ug_stream_t s) {
    return moe_gate_impl<bf16_t>(x, c, ld, wg, S, D, E, noise, true, gates, idx, s);
}
extern "C" int ug_moe_gate_top2_f32(const void* x, const void* c, int64_t ld, const void* wg, int64_t S, int64_t D, int32_t E, const float* noise, float* gates,
                                    int32_t* idx, ug_stream_t s) {
    return moe_gate_impl<float>(x, c, ld, wg, S, D, E, noise, true, gates, idx, s);
}

extern "C" int ug_moe_capacity_rts(const float* gates, const int32_t* idx, const float* uniform, int64_t S, int32_t E,
                                   int64_t capacity, int32_t* slot, int32_t* token_of_slot, int64_t* exp_counts, float* l_aux,
                                   ug_stream_t stream) {
    UG_REQUIRE(gates && idx && uniform && slot && token_of_slot && exp_counts && l_aux, UG_ERR_BAD_SHAPE, "ug_moe_capacity_rts: null argument");
    UG_REQUIRE(S > 0 && S < (1ll << 30) && E >= 1 && E <= GATE_MAXE && capacity > 0 && capacity < (1ll << 30), UG_ERR_BAD_SHAPE,
               "ug_moe_capacity_rts: bad S/E/capacity");
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(moe_capacity_kernel, dim3((unsigned)E), dim3(1024), 0, s, idx, uniform, (int)S, (int)E, (int)capacity, slot,
                       token_of_slot, exp_counts);
    UG_CHECK_LAUNCH("ug_moe_capacity_rts");
    hipLaunchKernelGGL(moe_laux_kernel, dim3(1), dim3(1024), 0, s, gates, (const int64_t*)exp_counts, (const int32_t*)nullptr, (int)S, (int)E, l_aux, (float)E);
    UG_CHECK_LAUNCH("ug_moe_capacity_rts(l_aux)");
    return UG_OK;
}

extern "C" int ug_moe_capacity_top2(const float* gates, const int32_t* idx, int64_t S, int32_t E, int64_t capacity, int32_t* slot,
                                    int32_t* token_of_slot, float* weights, int64_t* exp_counts, float* l_aux, ug_stream_t stream) {
    UG_REQUIRE(gates && idx && slot && token_of_slot && weights && exp_counts && l_aux, UG_ERR_BAD_SHAPE, "ug_moe_capacity_top2: null argument");
    UG_REQUIRE(S > 0 && S < (1ll << 29) && E >= 2 && E <= GATE_MAXE && capacity > 0 && capacity < (1ll << 30), UG_ERR_BAD_SHAPE,
               "ug_moe_capacity_top2: bad S/E/capacity");
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(moe_capacity_top2_kernel, dim3((unsigned)E), dim3(1024), 0, s, idx, (int)S, (int)capacity, slot, token_of_slot, exp_counts);
    UG_CHECK_LAUNCH("ug_moe_capacity_top2");
    hipLaunchKernelGGL(moe_weights_top2_kernel, dim3((unsigned)((S + 255) / 256)), dim3(256), 0, s, gates, idx, (const int32_t*)slot, (int)S, (int)E, weights);
    UG_CHECK_LAUNCH("ug_moe_capacity_top2(weights)");
    hipLaunchKernelGGL(moe_laux_kernel, dim3(1), dim3(1024), 0, s, gates, (const int64_t*)exp_counts, idx, (int)S, (int)E, l_aux, (float)E);
    UG_CHECK_LAUNCH("ug_moe_capacity_top2(l_aux)");
    return UG_OK;
}

template <typename T>
static int moe_gate_topk_impl(const void* x, const void* c, int64_t ld, const void* wg, int64_t S, int64_t D, int32_t E, int32_t K, float* gates,
                              float* logits, int32_t* idx, ug_stream_t stream) {
    if (S == 0) return UG_OK;
    UG_REQUIRE(x && c && wg && gates && logits && idx && S > 0 && D > 0, UG_ERR_BAD_SHAPE, "ug_moe_gate_topk: bad arguments");
    UG_REQUIRE(E >= 1 && E <= GATE_MAXE && K >= 1 && K <= E, UG_ERR_UNSUPPORTED, "ug_moe_gate_topk: E=%d K=%d not in 1 <= K <= E <= %d", E, K, GATE_MAXE);
    UG_REQUIRE(D % 8 == 0 && ld % 8 == 0 && ug_aligned(x, 16) && ug_aligned(c, 16) && ug_aligned(wg, 16), UG_ERR_BAD_ALIGN,
               "ug_moe_gate_topk: 16-byte alignment required");
    hipLaunchKernelGGL((moe_gate_topk_kernel<T>), dim3((unsigned)((S + 3) / 4)), dim3(256), 0, (hipStream_t)stream, (const T*)x, (const T*)c, ld,
                       (const T*)wg, S, (int)D, (int)E, (int)K, gates, logits, idx);
    UG_CHECK_LAUNCH("ug_moe_gate_topk");
    return UG_OK;
}
extern "C" int ug_moe_gate_topk(const void* x, const void* c, int64_t ld, const void* wg, int64_t S, int64_t D, int32_t E, int32_t K, float* gates,
                                float* logits, int32_t* idx, ug_stream_t s) {
    return moe_gate_topk_impl<bf16_t>(x, c, ld, wg, S, D, E, K, gates, logits, idx, s);
}
extern "C" int ug_moe_gate_topk_f32(const void* x, const void* c, int64_t ld, const void* wg, int64_t S, int64_t D, int32_t E, int32_t K, float* gates,
                                    float* logits, int32_t* idx, ug_stream_t s) {
    return moe_gate_topk_impl<float>(x, c, ld, wg, S, D, E, K, gates, logits, idx, s);
}

extern "C" int ug_moe_capacity_topk(const float* gates, const float* logits, const int32_t* idx, int64_t S, int32_t E, int32_t K, int64_t capacity,
                                    int32_t* slot, int32_t* token_of_slot, float* weights, int64_t* exp_counts, float* l_aux, ug_stream_t stream) {
    UG_REQUIRE(gates && logits && idx && slot && token_of_slot && weights && exp_counts && l_aux, UG_ERR_BAD_SHAPE, "ug_moe_capacity_topk: null argument");
    UG_REQUIRE(S > 0 && S < (1ll << 27) && E >= 1 && E <= GATE_MAXE && K >= 1 && K <= E && capacity > 0 && capacity < (1ll << 30), UG_ERR_BAD_SHAPE,
               "ug_moe_capacity_topk: bad S/E/K/capacity");
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(moe_capacity_topk_kernel, dim3((unsigned)E), dim3(1024), 0, s, idx, logits, (int)S, (int)E, (int)K, (int)capacity, slot, token_of_slot,
                       exp_counts);
    UG_CHECK_LAUNCH("ug_moe_capacity_topk");
    hipLaunchKernelGGL(moe_weights_topk_kernel, dim3((unsigned)((S + 255) / 256)), dim3(256), 0, s, gates, idx, (const int32_t*)slot, (int)S, (int)E, (int)K, weights);
    UG_CHECK_LAUNCH("ug_moe_capacity_topk(weights)");
    hipLaunchKernelGGL(moe_laux_kernel, dim3(1), dim3(1024), 0, s, gates, (const int64_t*)exp_counts, (const int32_t*)nullptr, (int)S, (int)E, l_aux,
                       (float)E / (float)K);
    UG_CHECK_LAUNCH("ug_moe_capacity_topk(l_aux)");
    return UG_OK;
}

extern "C" int ug_moe_dispatch_modulate(const void* x, int64_t ldx, const void* add, const void* mod, int64_t mod_estride, int64_t mod_bstride,
                                        const int32_t* token_of_slot, int32_t E, int64_t capacity, int64_t tokens_per_sample,
                                        int64_t D, void* out, ug_stream_t s) {
    return moe_dispatch_impl<bf16_t>(x, ldx, add, mod, mod_estride, mod_bstride, token_of_slot, E, capacity, tokens_per_sample, D, out, s);
}
extern "C" int ug_moe_dispatch_modulate_f32(const void* x, int64_t ldx, const void* add, const void* mod, int64_t mod_estride, int64_t mod_bstride,
                                            const int32_t* token_of_slot, int32_t E, int64_t capacity, int64_t tokens_per_sample,
                                            int64_t D, void* out, ug_stream_t s) {
    return moe_dispatch_impl<float>(x, ldx, add, mod, mod_estride, mod_bstride, token_of_slot, E, capacity, tokens_per_sample, D, out, s);
}

extern "C" int ug_moe_combine(const void* yh, const void* yc, const float* gates, const int32_t* idx, const int32_t* slot, int32_t E,
                              int64_t capacity, const void* xs, const void* cs, int64_t ld_s, int64_t s_rpb, int64_t s_bstride, void* out, int64_t ldo,
                              int64_t S, int64_t D, int32_t accumulate, ug_stream_t s) {
    return moe_combine_impl<bf16_t>(yh, yc, gates, idx, slot, E, capacity, xs, cs, ld_s, s_rpb, s_bstride, out, ldo, S, D, accumulate, s);
}
extern "C" int ug_moe_combine_f32(const void* yh, const void* yc, const float* gates, const int32_t* idx, const int32_t* slot, int32_t E,
                                  int64_t capacity, const void* xs, const void* cs, int64_t ld_s, int64_t s_rpb, int64_t s_bstride, void* out, int64_t ldo,
                                  int64_t S, int64_t D, int32_t accumulate, ug_stream_t s) {
    return moe_combine_impl<float>(yh, yc, gates, idx, slot, E, capacity, xs, cs, ld_s, s_rpb, s_bstride, out, ldo, S, D, accumulate, s);
}

extern "C" int ug_moe_combine_topk(const void* yh, const void* yc, const float* weights, const int32_t* idx, const int32_t* slot, int32_t K, int64_t kstride,
                                   int32_t E, int64_t capacity, const void* xs, const void* cs, int64_t ld_s, int64_t s_rpb, int64_t s_bstride, void* out,
                                   int64_t ldo, int64_t S, int64_t D, int32_t accumulate, ug_stream_t s) {
    return moe_combine_topk_impl<bf16_t>(yh, yc, weights, idx, slot, K, kstride, E, capacity, xs, cs, ld_s, s_rpb, s_bstride, out, ldo, S, D, accumulate, s);
}
extern "C" int ug_moe_combine_topk_f32(const void* yh, const void* yc, const float* weights, const int32_t* idx, const int32_t* slot, int32_t K, int64_t kstride,
                                       int32_t E, int64_t capacity, const void* xs, const void* cs, int64_t ld_s, int64_t s_rpb, int64_t s_bstride, void* out,
                                       int64_t ldo, int64_t S, int64_t D, int32_t accumulate, ug_stream_t s) {
    return moe_combine_topk_impl<float>(yh, yc, weights, idx, slot, K, kstride, E, capacity, xs, cs, ld_s, s_rpb, s_bstride, out, ldo, S, D, accumulate, s);
}
