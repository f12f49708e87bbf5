/*
 * unigen_hip.h — C ABI of libunigen_hip.so (gfx950 / MI355X).
 *
 * This is the drop-in boundary for ONE hot path of gavin-gqzhang/UniGen: the
 * condition-weaving + expert-modulation diffusion forward pass
 * (reference: src/UniGenTransformer.py:712-1450 UniGenFlux / MultiCondtionUniGenFlux,
 *  src/UniGenUtils.py:17-228,340-622, denoise loop src/UniGenPipeline.py:721-789).
 *
 * The reference has no FFI of its own (it is pure PyTorch eager); every entry point
 * below names the reference torch call site it replaces. Conventions:
 *   - extern "C", plain pointers and sizes, no torch types.
 *   - every pointer is a DEVICE pointer unless the name ends in _host.
 *   - asynchronous, stream-ordered on `stream` (a hipStream_t passed as void*);
 *     no allocation, no ownership transfer, no host synchronisation.
 *   - bf16 storage (uint16 bit patterns), fp32 accumulate / statistics / gate / RoPE tables.
 *   - every compute entry point has an fp32 VERIFICATION twin `<name>_f32` (section at the end): identical argument
 *     meaning, fp32 storage for every tensor that is bf16 here, no intermediate rounding. Slow, correctness-first
 *     kernels; the Python host runs the SAME orchestration through them when the model's parameters are fp32, which
 *     is how forward-level parity with the reference is shown to <= 1e-3 (a bf16 forward cannot be: eps = 7.8e-3).
 *   - return 0 on success, a negative UG_ERR_* otherwise; ug_last_error() gives the
 *     thread-local message of the most recent failure.
 *   - "row map": a logical row m of an operand lives at physical row
 *         (m / rows_per_batch) * batch_stride + (m % rows_per_batch)
 *     of its buffer (rows_per_batch == 0 means identity). It lets a GEMM read or write
 *     a token slice of a concatenated (text|image|condition) buffer without a copy.
 */
#ifndef UNIGEN_HIP_H
#define UNIGEN_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* ug_stream_t; /* hipStream_t */

enum {
    UG_OK = 0,
    UG_ERR_BAD_SHAPE = -1,
    UG_ERR_BAD_ALIGN = -2,
    UG_ERR_UNSUPPORTED = -3,
    UG_ERR_HIP = -4
};

/* GEMM epilogues (ug_gemm_desc.epilogue). v = bf16(acc + bias) is rounded first, as the
 * reference's separate nn.Linear does, then: */
enum {
    UG_EPI_BIAS = 0,       /* C = v                                            nn.Linear */
    UG_EPI_BIAS_GELU = 1,  /* C = gelu_tanh(v)                                 FeedForward net.0 (diffusers GELU approximate="tanh") */
    UG_EPI_RES_GATE = 2,   /* C = R + gate[sample(m)][n] * v                   x + gate.unsqueeze(1) * attn/ff  (FluxTransformerBlock) */
    UG_EPI_RES_SCALE = 3,  /* C = R + alpha * v                                x + cn_block(z) * conditioning_scale (UniGenTransformer.py:1104,1141) */
    UG_EPI_F32 = 4,        /* C(fp32) = acc + bias (no rounding; verification / gate logits) */
    UG_EPI_QKV_ROPE = 5    /* fused [to_q; to_k; to_v (; proj_mlp)] projection: columns [0, qk_until_n) are q | k heads of 128 and get
                              Attention.norm_q / norm_k (RMSNorm) + apply_rotary_emb in the epilogue (src/UniGenUtils.py:561-599, the work of
                              ug_qk_rmsnorm_rope); the other columns behave as UG_EPI_BIAS, or GELU from gelu_from_n > 0 on */
};

typedef struct ug_gemm_desc {
    /* C[m][n] = epilogue( sum_k A[m][k] * W[n][k] + bias[n] ), all bf16, W in nn.Linear layout [N][K].
     * Row maps (x_rpb, x_bstride): logical row m sits at physical row (m / rpb) * bstride + m % rpb (rpb 0: m itself). The A map may be
     * anything (bstride < rpb broadcasts a batch); a NON-MONOTONIC A map (bstride < rpb) is served by the 128^2 kernel only - the 256^2
     * kernel's DMA offsets are unsigned distances from a tile's first row - and is refused (UG_ERR_BAD_SHAPE) with UG_EPI_QKV_ROPE. */
    const void* A; int64_t lda; int64_t a_rpb; int64_t a_bstride;
    const void* W; int64_t ldw;
    const void* bias;                 /* [N] bf16 or NULL */
    void* C; int64_t ldc; int64_t c_rpb; int64_t c_bstride;
    const void* R; int64_t ldr; int64_t r_rpb; int64_t r_bstride; /* residual; MAY ALIAS C (same base, ld and row map): every kernel path reads a residual element before the store that overwrites it (tests/test_kernels_gpu.py::test_gemm_residual_aliased_to_output_every_dispatch_path) */
    const void* gate; int64_t gate_ld; int64_t rows_per_sample;   /* gate[(m / rows_per_sample) * gate_ld + n] bf16 */
    float alpha;
    int32_t epilogue;
    int64_t M, N, K;
    /* grouped / batched over `groups` experts (blockIdx.z); strides in ELEMENTS */
    int32_t groups; int32_t _pad0;
    int64_t a_gstride, w_gstride, bias_gstride, c_gstride;
    /* optional LoRA epilogue (peft 0.15 Linear; src/lora_switching_module.py:11-38):
     *   acc += lora_scale-premultiplied T[m][0..r) . B[n][0..r)   with T = scale * (x A^T) computed by a prior ug_gemm_bf16 */
    const void* lora_T; int64_t ldt;
    const void* lora_B; int64_t ldb;
    int32_t lora_r; int32_t _pad1;
    /* group strides (ELEMENTS) of the residual and gate operands for grouped residual epilogues (per-expert transformer blocks) */
    int64_t r_gstride, gate_gstride;
    /* optional caller-owned scratch (>= ug_gemm_workspace_bytes(), 16-byte aligned) for the split-K treatment of the last, partially
     * filled round of tiles; NULL = plain tiles only. Its first 4096 bytes (arrival tickets) must be ZERO before the first call;
     * every call leaves them zero again. The rest needs no initialisation. One workspace per stream. */
    void* workspace; int64_t workspace_bytes;
    /* Column split, for one launch over the concatenated weights of two Linear layers that read the same input (the single block's
     * [to_q; to_k; to_v; proj_mlp], diffusers FluxSingleTransformerBlock, called at src/UniGenTransformer.py:1151):
     *   UG_EPI_BIAS_GELU applies GELU only to columns n >= gelu_from_n (0: all columns);
     *   output column n is stored at column n + c_shift when n >= c_shift_from_n > 0 (a gap in the destination row).
     * Both boundaries must be multiples of 256 (a tile never straddles them); c_shift a multiple of 8. */
    int64_t gelu_from_n, c_shift_from_n, c_shift;
    /* UG_EPI_QKV_ROPE only. Heads are qk_dh wide (128; 0 means 128; or 64: SD3.5); q heads fill columns [0, qk_until_n / 2), k heads
     * [qk_until_n / 2, qk_until_n).
     *   qk_wq, qk_wk : [qk_dh] bf16 RMSNorm weights;  rope_cs : fp32 [positions][qk_dh / 2][2] = (cos, sin) of each rotation pair (may be NULL
     *   at qk_dh = 64: RMSNorm only, JointAttnProcessor2_0 has no RoPE);
     *   row m sits at position rope_pos0 + (m % rope_rpb) (rope_rpb 0: m itself).
     * Needs M, N, qk_until_n multiples of 256, groups 1, no LoRA segment, 16-byte aligned C rows. */
    const void* qk_wq; const void* qk_wk; const float* rope_cs;
    int64_t rope_rpb, rope_pos0, qk_until_n;
    float qk_eps; int32_t qk_dh;
} ug_gemm_desc;

/* bytes of ug_gemm_desc.workspace that are always sufficient (any shape) */
int64_t ug_gemm_workspace_bytes(void);

/* replaces every nn.Linear on the path (torch F.linear -> BLAS) incl. fused epilogues. */
int ug_gemm_bf16(const ug_gemm_desc* d, ug_stream_t stream);

/* out[m][n] = R[m][n] + bf16( sum_k act(x[m][k]) * W[n][k] + b[n] ),  M <= 16 rows (per-sample vectors).
 * act: 0 none, 1 SiLU. Replaces AdaLayerNormZero.linear(silu(emb)), TimestepEmbedding, PixArtAlphaTextProjection
 * (diffusers 0.32.2 normalization.py / embeddings.py; called from UniGenTransformer.py:1222,1048-1049). */
int ug_small_linear_bf16(const void* x, int64_t ldx, const void* W, int64_t ldw, const void* bias,
                         const void* R, int64_t ldr, void* out, int64_t ldo,
                         int64_t M, int64_t N, int64_t K, int32_t act_in, ug_stream_t stream);

/* out = LayerNorm(x; eps, no affine) * (1 + scale[sample]) + shift[sample]   (AdaLayerNormZero / norm2+modulate /
 * AdaLayerNormContinuous; diffusers normalization.py; src/UniGenUtils.py:354-373). shift/scale are bf16 rows of the
 * adaLN linear output with leading dimension mod_ld. */
int ug_adaln_modulate(const void* x, int64_t ldx, int64_t x_rpb, int64_t x_bstride,
                      const void* shift, const void* scale, int64_t mod_ld, int64_t rows_per_sample,
                      void* out, int64_t ldo, int64_t rows, int64_t D, float eps, ug_stream_t stream);

/* In-place RMSNorm(q), RMSNorm(k) (per head, weight) then RoPE on a fused [.. q | k | v ..] projection buffer.
 * Row r in [0, rows_per_batch) of batch b is at buf + (b*batch_stride_rows + r)*ld and has sequence position
 * pos = pos_offset + r. Positions < split use (wq_a, wk_a) (text: norm_added_q/k), positions >= split use (wq_b, wk_b)
 * (image: norm_q/k). cos/sin: fp32 [>= pos_offset + rows_per_batch][dh] indexed by pos, or NULL (no RoPE);
 * weights may be NULL (no qk-norm). q_off < 0 skips q (k/v-only context streams).
 * Replaces Attention.norm_q/norm_k + apply_rotary_emb (FluxAttnProcessor2_0; src/UniGenUtils.py:561-599). */
int ug_qk_rmsnorm_rope(void* buf, int64_t ld, int64_t batches, int64_t rows_per_batch, int64_t batch_stride_rows,
                       int64_t pos_offset, int64_t q_off, int64_t k_off, int32_t heads, int32_t dh,
                       const void* wq_a, const void* wk_a, const void* wq_b, const void* wk_b, int64_t split,
                       const float* cos_tab, const float* sin_tab, float eps, ug_stream_t stream);

/* O = softmax(Q K^T / sqrt(dh)) V, non-causal, no mask; Q rows [Lq], K/V rows [Lkv]; head h occupies
 * columns [h*dh, (h+1)*dh) of each row. Strides in elements. Replaces F.scaled_dot_product_attention
 * (src/UniGenUtils.py:601 and diffusers FluxAttnProcessor2_0). dh in {64, 128}. */
int ug_flash_attn_fwd(const void* q, int64_t q_row_stride, int64_t q_batch_stride,
                      const void* k, int64_t k_row_stride, int64_t k_batch_stride,
                      const void* v, int64_t v_row_stride, int64_t v_batch_stride,
                      void* o, int64_t o_row_stride, int64_t o_batch_stride,
                      int64_t batches, int32_t heads, int64_t Lq, int64_t Lkv, int32_t dh,
                      float softmax_scale, ug_stream_t stream);

/* Sinusoidal Timesteps(256, flip_sin_to_cos=True, downscale_freq_shift=0): out[b] = [cos(t f) | sin(t f)] as bf16.
 * (diffusers embeddings.get_timestep_embedding; called inside time_text_embed, UniGenTransformer.py:1222). */
int ug_timestep_embed(const float* t, void* out, int64_t ldo, int64_t B, int32_t dim, ug_stream_t stream);

/* x = bf16( float(x) + bf16( bf16(dt) * v ) )  FlowMatchEulerDiscreteScheduler.step (called at src/UniGenPipeline.py:768 / :411) as torch evaluates
 * `sample.float() + (sigma_next - sigma) * model_output`: the step is a 0-dim fp32 tensor, so the product is a bf16 op (both operands cast, result rounded)
 * before the fp32 add. dt = the fp32 difference of the scheduler's fp32 sigmas. (dt = -0.25, FLUX-schnell's four steps: exact.) _f32 twin: no rounding. */
int ug_euler_step(void* x, const void* v, float dt, int64_t n, ug_stream_t stream);

/* out = uncond + gs * (text - uncond), each step rounded to bf16 as the reference's tensor ops do: classifier-free guidance of
 * UniGenSD3Pipeline (src/UniGenPipeline.py:404-407). n elements, contiguous. */
int ug_cfg_combine(const void* uncond, const void* text, float guidance_scale, void* out, int64_t n, ug_stream_t stream);

/* out = bf16(a + b) elementwise (bf16), rows x D with leading dims. */
int ug_add_bf16(const void* a, int64_t lda, const void* b, int64_t ldb, void* out, int64_t ldo,
                int64_t rows, int64_t D, ug_stream_t stream);

/* x[r][:] = bf16( float(x[r][:]) + table[r % rows_per_batch][:] ), table fp32: PatchEmbed's `(latent + pos_embed).to(latent.dtype)`
 * with the centre-cropped sincos table (diffusers embeddings.PatchEmbed.forward; UniGenSD3, src/UniGenTransformer.py:663,510). */
int ug_add_rowbcast_f32(void* x, int64_t ldx, const float* table, int64_t ldt, int64_t rows, int64_t rows_per_batch, int64_t D,
                        ug_stream_t stream);

/* out[i][:W] = src[idx[i]][:W] (zeros when idx[i] < 0): per-slot rows of per-sample tables (per-token AdaLN embeddings of the
 * SD3 transformer-block experts: the reference broadcasts temb per token and dispatches it, src/UniGenUtils.py:107-109). */
int ug_gather_rows(const void* src, int64_t ld_src, const int32_t* idx, void* out, int64_t ld_out, int64_t n, int64_t W,
                   ug_stream_t stream);

/* ---- CoMoE (src/UniGenUtils.py:74-191 MOELayer + deepspeed 0.16.5 top1gating; UniGenTransformer.py:925-1026) ---- */

/* Gate: xc = bf16(x + c); logits = xc.float() @ wg.float()^T (fp32); gates = softmax; idx = argmax.
 * Writes gates fp32 [S][E], idx int32 [S]. */
int ug_moe_gate_top1(const void* x, const void* c, int64_t ld, const void* wg, int64_t S, int64_t D, int32_t E,
                     float* gates, int32_t* idx, ug_stream_t stream);

/* Capacity selection with Random Token Selection (use_rts=True, drop_tokens=True): per expert keep the `capacity`
 * tokens with the largest uniform[s][idx[s]]; slot = rank of the token among kept tokens of its expert (token order).
 * Writes slot int32 [S] (-1 = dropped), token_of_slot int32 [E][capacity] (-1 = empty), exp_counts int64 [E]
 * (pre-drop counts, as deepspeed returns), l_aux fp32 scalar = E * sum_e mean_s(gates[s][e]) * mean_s(idx[s]==e). */
int ug_moe_capacity_rts(const float* gates, const int32_t* idx, const float* uniform, int64_t S, int32_t E,
                        int64_t capacity, int32_t* slot, int32_t* token_of_slot, int64_t* exp_counts, float* l_aux,
                        ug_stream_t stream);

/* ---- top-2 gating (control_params.top_num = 2 -> MoE(..., k = 2), src/UniGenTransformer.py:162,197 / :808,857 / :1565,1650; TopKGate then calls
 * deepspeed 0.16.5 sharded_moe.top2gating(logits, capacity_factor = 1, min_capacity = 4, drop_tokens = True, top2_2nd_expert_sampling = True),
 * a dependency that is not in /root/reference: restated from its published source, parity unpinned). Arrays over the two choices are
 * choice-major [2][S]. ---- */

/* Gate: gates = softmax(logits) fp32 [S][E] as ug_moe_gate_top1; idx[0][s] = argmax; idx[1][s] = argmax over the OTHER experts of
 * logits[s][e] + noise[s][e] (noise: the Gumbel(0, 1) sample deepspeed adds to the logits for its second choice, fp32 [S][E]; NULL = no
 * sampling, the second-largest logit). 2 <= E <= 16. */
int ug_moe_gate_top2(const void* x, const void* c, int64_t ld, const void* wg, int64_t S, int64_t D, int32_t E, const float* noise,
                     float* gates, int32_t* idx, ug_stream_t stream);

/* Capacity rule of top2gating (no random token selection): per expert the first choices take slots in token order, the second choices
 * follow behind ALL its first choices; a choice at or beyond `capacity` (= max(ceil(S / E * 2), 4) for the reference's settings) is dropped.
 * Writes slot int32 [2][S] (-1 = dropped), token_of_slot int32 [E][capacity] (-1 = empty; the layout ug_moe_dispatch_modulate reads),
 * weights fp32 [2][S] = the kept choices' gate probabilities over their sum clamped at FLT_EPSILON (0 for a dropped choice), exp_counts
 * int64 [E] = first + second choices before the drop, l_aux fp32 scalar = E * sum_e mean_s(gates[s][e]) * mean_s(idx[0][s] == e). */
int ug_moe_capacity_top2(const float* gates, const int32_t* idx, int64_t S, int32_t E, int64_t capacity, int32_t* slot,
                         int32_t* token_of_slot, float* weights, int64_t* exp_counts, float* l_aux, ug_stream_t stream);

/* deepspeed topkgating (TopKGate with k > 2: control_params.top_num > 2, src/UniGenTransformer.py:808 -> src/UniGenUtils.py:33-36): the gate.
 * gates fp32 [S][E] = softmax of the fp32 logits of bf16(x + c), logits fp32 [S][E] (the capacity rule ranks LOGITS), idx int32 [K][S] = the
 * token's K choices by descending logit (the lower expert index first among equal logits). 1 <= K <= E <= 16. No random draw. */
int ug_moe_gate_topk(const void* x, const void* c, int64_t ld, const void* wg, int64_t S, int64_t D, int32_t E, int32_t K, float* gates,
                     float* logits, int32_t* idx, ug_stream_t stream);

/* Capacity rule of topkgating (drop_policy "probs"): expert e keeps the `capacity` (= max(ceil(S / E * K), 4) for the reference's settings) largest
 * entries of its column of [logit if e is one of the token's K choices, else 0] over ALL S tokens - a chosen logit below zero loses to every
 * non-chooser's zero - ties in token order; a choice survives if its entry is kept. Writes slot int32 [K][S] (-1 = dropped; slots in token order
 * among the kept), token_of_slot int32 [E][capacity] (-1 = empty), weights fp32 [K][S] = the kept choices' gate probabilities over their sum
 * clamped at FLT_EPSILON, exp_counts int64 [E] = choosers before the drop, l_aux fp32 scalar = (E / K) * sum_e mean_s(gates[s][e]) * exp_counts[e] / S. */
int ug_moe_capacity_topk(const float* gates, const float* logits, const int32_t* idx, int64_t S, int32_t E, int32_t K, int64_t capacity,
                         int32_t* slot, int32_t* token_of_slot, float* weights, int64_t* exp_counts, float* l_aux, ug_stream_t stream);

/* Combine over a token's K <= 16 (expert, slot) pairs - einsum("sec,ecm->sm") src/UniGenUtils.py:183 with top2gating's / topkgating's combine weights:
 *   eh = bf16( sum_k bf16(weights[k][s]) * yh[idx[k][s]][slot[k][s]] )   (fp32 sum, one rounding; dropped choices contribute nothing),
 * ec likewise; residual sums, row map and `accumulate` exactly as ug_moe_combine. weights / idx / slot: [K][kstride], kstride >= S
 * (a slice of a longer token axis keeps its parent's stride). K = 1 with weights = the gate probability reproduces ug_moe_combine. */
int ug_moe_combine_topk(const void* yh, const void* yc, const float* weights, const int32_t* idx, const int32_t* slot, int32_t K,
                        int64_t kstride, int32_t E, int64_t capacity, const void* xs, const void* cs, int64_t ld_s, int64_t s_rpb,
                        int64_t s_bstride, void* out, int64_t ldo, int64_t S, int64_t D, int32_t accumulate, ug_stream_t stream);

/* Dispatch + expert modulation prologue (replaces einsum("sec,sm->ecm") src/UniGenUtils.py:140 and the s-scaling of
 * modulated_flatten src/UniGenUtils.py:204-228):
 *   out[e][slot][:] = bf16( mod[e][sample(tok)][:] * bf16( x[tok][:] + (add ? add[e][slot][:] : 0) ) ), zeros for empty slots.
 * mod: bf16, row of (expert e, sample b) at mod + e * mod_estride + b * mod_bstride (elements) = Linear(768->D)(pooled) of that
 * expert (all experts' linears run as ONE launch over their stacked weights, so the natural layout is [B][E][D]); NULL for a plain
 * dispatch (transformer-block experts). tokens_per_sample = N. */
int ug_moe_dispatch_modulate(const void* x, int64_t ldx, const void* add, const void* mod, int64_t mod_estride, int64_t mod_bstride,
                             const int32_t* token_of_slot, int32_t E, int64_t capacity, int64_t tokens_per_sample,
                             int64_t D, void* out, ug_stream_t stream);

/* Combine (replaces einsum("sec,ecm->sm") src/UniGenUtils.py:183) fused with the CoMoE residual sums
 * (UniGenTransformer.py:1024,1089):
 *   eh = bf16(bf16(p) * yh[e][slot]) (0 if dropped), ec likewise from yc;
 *   out = bf16( bf16(xs + eh) + bf16(cs + ec) )    when xs/cs given (shared experts),
 *   out = bf16( eh + ec )                          otherwise.
 * If `accumulate` != 0, out = bf16(out_prev + that)  (MultiCondtionUniGenFlux sum over conditions, :1316).
 * Token s of xs / cs sits at physical row rowmap(s; s_rpb, s_bstride) (file header: "row map"; 0, 0 = identity): the shared experts'
 * image and condition streams are the two halves of one [B][2N][D] buffer, so all B samples combine in one launch. */
int ug_moe_combine(const void* yh, const void* yc, const float* gates, const int32_t* idx, const int32_t* slot,
                   int32_t E, int64_t capacity, const void* xs, const void* cs, int64_t ld_s, int64_t s_rpb, int64_t s_bstride,
                   void* out, int64_t ldo, int64_t S, int64_t D, int32_t accumulate, ug_stream_t stream);

/* FluxPipeline._pack_latents / _unpack_latents (diffusers; called at src/UniGenPipeline.py:641 and :796):
 *   packed[b][(i, j)][c*4 + dy*2 + dx] = latents[b][c][2i + dy][2j + dx],   latents [B][C][H][W], packed [B][(H/2)(W/2)][4C], contiguous. */
int ug_pack_latents(const void* latents, void* packed, int64_t B, int64_t C, int64_t H, int64_t W, ug_stream_t stream);
int ug_unpack_latents(const void* packed, void* latents, int64_t B, int64_t C, int64_t H, int64_t W, ug_stream_t stream);

/* ---- AutoencoderKL (the VAE either side of the hot path; reference src/UniGenPipeline.py:635-636 vae.encode of the
 * condition image, :797-798 vae.decode of the latents; diffusers 0.32.2 Encoder / Decoder / ResnetBlock2D / Upsample2D / Downsample2D /
 * Attention). Activations are NHWC [B][H][W][C]; 1x1 convolutions and the attention projections are ug_gemm_bf16 on [B H W][C]. ---- */

typedef struct ug_conv_desc {
    /* out[b][oy][ox][n] = R[..] + bf16( bias[n] + sum_{ky,kx,c} x[b][(oy*stride + ky - pad_t) >> up][(ox*stride + kx - pad_l) >> up][c] * w[n][ky][kx][c] ),
     * taps whose (virtual) coordinate falls outside [0, H << up) x [0, W << up) contribute zero. up = 1 folds Upsample2D's nearest-2x
     * interpolation into the gather; Downsample2D(padding=0)'s F.pad(x, (0, 1, 0, 1)) + stride-2 conv is pad_t = pad_l = 0, stride = 2. */
    const void* x; int64_t B, H, W, Cin;        /* NHWC input; Cin a multiple of 64 (zero-pad the channels) */
    const void* w;                              /* [Cout][KH][KW][Cin] (torch's [Cout][Cin][KH][KW] permuted once at load time) */
    const void* bias;                           /* [Cout] or NULL */
    const void* R;                              /* residual, NHWC [B][Ho][Wo][Cout] or NULL (may alias out) */
    void* out; int64_t Ho, Wo, Cout;            /* Cout a multiple of 4 */
    int32_t KH, KW, stride, pad_t, pad_l, up;
    const void* zero_page;                      /* >= 128 bytes of zeros, 16-byte aligned (device): what padding taps read */
    int64_t zero_page_bytes;                    /* size of zero_page (0 is read as 128). With >= 2 * (Cin + 64) bytes, Cin / 64 a power of two >= 2 and Cout,
                                                 * B Ho Wo multiples of 256, the convolution runs on the 256^2 GEMM kernel (its A operand gathered per tap) */
} ug_conv_desc;

/* torch F.conv2d (called by diffusers Conv2d layers of the VAE) as an implicit GEMM on MFMA. */
int ug_conv2d_nhwc(const ug_conv_desc* d, ug_stream_t stream);

/* out = [SiLU](GroupNorm(x; G groups, eps, gamma, beta)), x / out NHWC [B][HW][C] (F.group_norm + F.silu in ResnetBlock2D / Attention.group_norm /
 * conv_norm_out). Deterministic two-pass statistics in fp64; workspace >= ug_groupnorm_workspace_bytes(B, HW, G), 8-byte aligned, no init needed. */
int64_t ug_groupnorm_workspace_bytes(int64_t B, int64_t HW, int32_t G);
int ug_groupnorm_nhwc(const void* x, const void* gamma, const void* beta, void* out, void* workspace, int64_t workspace_bytes,
                      int64_t B, int64_t HW, int64_t C, int32_t G, float eps, int32_t silu, ug_stream_t stream);

/* P[r][c] = softmax_c(scale * S[r][c]), S fp32 (ug_gemm_bf16 with UG_EPI_F32), P bf16: the VAE mid-block attention has one head of dim C
 * (F.scaled_dot_product_attention in AttnProcessor2_0), run as scores GEMM -> this -> P.V GEMM. */
int ug_softmax_rows(const float* S, int64_t ld_s, void* P, int64_t ld_p, int64_t rows, int64_t cols, float scale, ug_stream_t stream);

/* Layout changes at the VAE boundary: NCHW [B][C][HW] <-> NHWC [B][HW][Cp], Cp >= C (extra channels zero / ignored). div != 0 applies the
 * decode side's latent un-scaling on the way in: v -> bf16(bf16(v / div) + add)  (latents / scaling_factor + shift_factor, :797). */
int ug_nchw_to_nhwc(const void* in, void* out, int64_t B, int64_t C, int64_t HW, int64_t Cp, float div, float add, ug_stream_t stream);
int ug_nhwc_to_nchw(const void* in, void* out, int64_t B, int64_t C, int64_t HW, int64_t Cp, ug_stream_t stream);

/* z = ((mean + exp(0.5 clamp(logvar, -30, 20)) * noise) - shift) * scale: DiagonalGaussianDistribution.sample() and the latent scaling of
 * :635-636. moments NHWC [B][HW][Cp] (mean = channels [0, L), logvar = [L, 2L)); noise, z NCHW [B][L][HW]. */
int ug_vae_sample(const void* moments, int64_t Cp, const void* noise, void* z, int64_t B, int64_t L, int64_t HW, float shift, float scale,
                  ug_stream_t stream);

/* ---- fp32 verification twins (see the conventions at the top). Same arguments as the functions they mirror; every `void*` tensor
 * that is bf16 there is fp32 here (weights included); fp32 / integer arguments are unchanged. ug_gemm_f32: K, N unconstrained,
 * the column-split boundaries multiples of 64, no workspace. ug_small_linear_f32: M <= 64. ---- */
int ug_gemm_f32(const ug_gemm_desc* d, ug_stream_t stream);
int ug_small_linear_f32(const void* x, int64_t ldx, const void* W, int64_t ldw, const void* bias,
                        const void* R, int64_t ldr, void* out, int64_t ldo,
                        int64_t M, int64_t N, int64_t K, int32_t act_in, ug_stream_t stream);
int ug_adaln_modulate_f32(const void* x, int64_t ldx, int64_t x_rpb, int64_t x_bstride,
                          const void* shift, const void* scale, int64_t mod_ld, int64_t rows_per_sample,
                          void* out, int64_t ldo, int64_t rows, int64_t D, float eps, ug_stream_t stream);
int ug_qk_rmsnorm_rope_f32(void* buf, int64_t ld, int64_t batches, int64_t rows_per_batch, int64_t batch_stride_rows,
                           int64_t pos_offset, int64_t q_off, int64_t k_off, int32_t heads, int32_t dh,
                           const void* wq_a, const void* wk_a, const void* wq_b, const void* wk_b, int64_t split,
                           const float* cos_tab, const float* sin_tab, float eps, ug_stream_t stream);
int ug_flash_attn_fwd_f32(const void* q, int64_t q_row_stride, int64_t q_batch_stride,
                          const void* k, int64_t k_row_stride, int64_t k_batch_stride,
                          const void* v, int64_t v_row_stride, int64_t v_batch_stride,
                          void* o, int64_t o_row_stride, int64_t o_batch_stride,
                          int64_t batches, int32_t heads, int64_t Lq, int64_t Lkv, int32_t dh,
                          float softmax_scale, ug_stream_t stream);
int ug_timestep_embed_f32(const float* t, void* out, int64_t ldo, int64_t B, int32_t dim, ug_stream_t stream);
int ug_euler_step_f32(void* x, const void* v, float dt, int64_t n, ug_stream_t stream);
int ug_cfg_combine_f32(const void* uncond, const void* text, float guidance_scale, void* out, int64_t n, ug_stream_t stream);
int ug_add_f32(const void* a, int64_t lda, const void* b, int64_t ldb, void* out, int64_t ldo,
               int64_t rows, int64_t D, ug_stream_t stream);
int ug_add_rowbcast_f32_f32(void* x, int64_t ldx, const float* table, int64_t ldt, int64_t rows, int64_t rows_per_batch, int64_t D,
                            ug_stream_t stream);
int ug_gather_rows_f32(const void* src, int64_t ld_src, const int32_t* idx, void* out, int64_t ld_out, int64_t n, int64_t W,
                       ug_stream_t stream);
int ug_moe_gate_top1_f32(const void* x, const void* c, int64_t ld, const void* wg, int64_t S, int64_t D, int32_t E,
                         float* gates, int32_t* idx, ug_stream_t stream);
int ug_moe_gate_top2_f32(const void* x, const void* c, int64_t ld, const void* wg, int64_t S, int64_t D, int32_t E, const float* noise,
                         float* gates, int32_t* idx, ug_stream_t stream);
int ug_moe_gate_topk_f32(const void* x, const void* c, int64_t ld, const void* wg, int64_t S, int64_t D, int32_t E, int32_t K, float* gates,
                         float* logits, int32_t* idx, ug_stream_t stream);
int ug_moe_combine_topk_f32(const void* yh, const void* yc, const float* weights, const int32_t* idx, const int32_t* slot, int32_t K,
                            int64_t kstride, int32_t E, int64_t capacity, const void* xs, const void* cs, int64_t ld_s, int64_t s_rpb,
                            int64_t s_bstride, void* out, int64_t ldo, int64_t S, int64_t D, int32_t accumulate, ug_stream_t stream);
int ug_moe_dispatch_modulate_f32(const void* x, int64_t ldx, const void* add, const void* mod, int64_t mod_estride, int64_t mod_bstride,
                                 const int32_t* token_of_slot, int32_t E, int64_t capacity, int64_t tokens_per_sample,
                                 int64_t D, void* out, ug_stream_t stream);
int ug_moe_combine_f32(const void* yh, const void* yc, const float* gates, const int32_t* idx, const int32_t* slot,
                       int32_t E, int64_t capacity, const void* xs, const void* cs, int64_t ld_s, int64_t s_rpb, int64_t s_bstride,
                       void* out, int64_t ldo, int64_t S, int64_t D, int32_t accumulate, ug_stream_t stream);
int ug_conv2d_nhwc_f32(const ug_conv_desc* d, ug_stream_t stream);      /* any Cin / Cout, no zero_page */
int ug_groupnorm_nhwc_f32(const void* x, const void* gamma, const void* beta, void* out, void* workspace, int64_t workspace_bytes,
                          int64_t B, int64_t HW, int64_t C, int32_t G, float eps, int32_t silu, ug_stream_t stream);
int ug_softmax_rows_f32(const float* S, int64_t ld_s, void* P, int64_t ld_p, int64_t rows, int64_t cols, float scale, ug_stream_t stream);
int ug_nchw_to_nhwc_f32(const void* in, void* out, int64_t B, int64_t C, int64_t HW, int64_t Cp, float div, float add, ug_stream_t stream);
int ug_nhwc_to_nchw_f32(const void* in, void* out, int64_t B, int64_t C, int64_t HW, int64_t Cp, ug_stream_t stream);
int ug_vae_sample_f32(const void* moments, int64_t Cp, const void* noise, void* z, int64_t B, int64_t L, int64_t HW, float shift, float scale,
                      ug_stream_t stream);
int ug_pack_latents_f32(const void* latents, void* packed, int64_t B, int64_t C, int64_t H, int64_t W, ug_stream_t stream);
int ug_unpack_latents_f32(const void* packed, void* latents, int64_t B, int64_t C, int64_t H, int64_t W, ug_stream_t stream);

/* Diagnostic, not on the hot path: one launch of a bare bf16 MFMA loop (operands in registers, one wave per SIMD, pseudo-random
 * operand values) on `blocks` workgroups of 256 threads; shape 0 = v_mfma_f32_32x32x16_bf16, 1 = v_mfma_f32_16x16x32_bf16. The caller
 * times it (HIP events) to get the measured MFMA peak that bench.py reports as `roofline.peak_measured`.
 * scratch: blocks * 256 floats (device). *flops_out_host (HOST pointer, may be NULL) receives the FLOPs of the launch. */
int ug_probe_mfma_bf16(int32_t shape, int64_t blocks, int64_t iters, void* scratch, double* flops_out_host, ug_stream_t stream);

/* ---- Backward pass of the control-module training step (reference train.py:622-662 `accelerator.backward(loss)` over the same
 * forward). The matrix work of the backward is ug_gemm_bf16 again (dX = dY W through a transposed copy of W, dW = dY^T X through transposed copies
 * of dY and X, the attention backward as grouped GEMMs per sample); these entry points are what sits between those GEMMs. Each replaces the
 * autograd formula of the torch op named; each has an `_f32` twin like the forward entry points. ---- */
/* dst[b][c][r] = src[b][r][c] for r < rows, c < cols; dst[b][c][rows .. rows_pad) = 0. (Tensor.t().contiguous() of an operand) */
int ug_transpose(const void* src, int64_t ld_src, int64_t src_bstride, void* dst, int64_t ld_dst, int64_t dst_bstride, int64_t batch, int64_t rows,
                 int64_t cols, int64_t rows_pad, ug_stream_t stream);
/* out[g][c] = alpha * sum over the rows r of group g (rows_per_group consecutive rows) of a[r][c] * (b ? b[r][c] : 1), fp32 accumulation.
 * (grad of a bias / a per-sample gate, shift, scale: `.sum(dim)` in the backward of a broadcast) */
int ug_colsum(const void* a, int64_t lda, const void* b, int64_t ldb, void* out, int64_t ldo, int64_t rows, int64_t cols, int64_t rows_per_group,
              float alpha, void* workspace, int64_t workspace_bytes, ug_stream_t stream);
/* bytes of the caller-owned fp32 scratch of ug_colsum (per-chunk partial sums, added in a fixed order: run-to-run reproducible) */
int64_t ug_colsum_workspace_bytes(int64_t rows, int64_t cols, int64_t rows_per_group);
/* y[r] = (x ? x[r] : 0) + gate[r / rows_per_sample] * a[r], the product rounded first: `x + gate.unsqueeze(1) * a` of every transformer block
 * (FluxTransformerBlock / FluxSingleTransformerBlock, called at src/UniGenTransformer.py:1129,1151) as one op of the training forward; with x = NULL
 * its backward d a = gate * d y (d x = d y; d gate = ug_colsum(d y, a) per sample). */
int ug_gate_residual(const void* x, int64_t ldx, const void* a, int64_t lda, const void* gate, int64_t gate_ld, int64_t rows_per_sample, void* y,
                     int64_t ldy, int64_t rows, int64_t D, ug_stream_t stream);
int ug_gate_residual_f32(const void* x, int64_t ldx, const void* a, int64_t lda, const void* gate, int64_t gate_ld, int64_t rows_per_sample, void* y,
                         int64_t ldy, int64_t rows, int64_t D, ug_stream_t stream);
/* backward of ug_moe_gate_top1 (deepspeed TopKGate / top1gating gate softmax, src/UniGenUtils.py:99; reference: torch autograd through
 * F.linear(x.float(), wg.float()) + softmax, train.py:654): d gates [S, E] fp32 -> dx [S, D] = d(x + c) (the gradient of BOTH x and c) and
 * dwg_partials fp32 [ug_moe_gate_bwd_slices(S)][E][D], whose sum over the first axis (fixed order: reproducible) is d wg. E <= 16. */
int ug_moe_gate_bwd(const float* gates, const float* dgates, const void* x, const void* c, int64_t ld, const void* wg, int64_t S, int64_t D, int32_t E,
                    void* dx, int64_t lddx, float* dwg_partials, ug_stream_t stream);
int ug_moe_gate_bwd_f32(const float* gates, const float* dgates, const void* x, const void* c, int64_t ld, const void* wg, int64_t S, int64_t D, int32_t E,
                        void* dx, int64_t lddx, float* dwg_partials, ug_stream_t stream);
int64_t ug_moe_gate_bwd_slices(int64_t S);
/* y = gelu_tanh(x); dx = dy * gelu_tanh'(x)   (F.gelu(approximate="tanh") of FeedForward net.0 / proj_mlp and its backward) */
int ug_gelu_tanh(const void* x, void* y, int64_t n, ug_stream_t stream);
int ug_gelu_tanh_bwd(const void* x, const void* dy, void* dx, int64_t n, ug_stream_t stream);
/* backward of ug_adaln_modulate (LayerNorm(x) * (1 + scale) + shift): dx, and partials: fp32 [samples][P][2][D], P =
 * ug_adaln_modulate_bwd_partials(rows, rows_per_sample): partial column sums over disjoint row sets of each sample of dy ([..][0][D], whose sum
 * over P is d shift) and of dy * xhat ([..][1][D]: d scale); the caller adds them. scale: [samples][mod_ld] as in the forward. D <= 4096, D % 8 == 0. */
int64_t ug_adaln_modulate_bwd_partials(int64_t rows, int64_t rows_per_sample);
int ug_adaln_modulate_bwd(const void* x, int64_t ldx, const void* dy, int64_t lddy, const void* scale, int64_t mod_ld, int64_t rows_per_sample,
                          void* dx, int64_t lddx, void* partials, int64_t rows, int64_t D, float eps, ug_stream_t stream);
/* backward of ug_qk_rmsnorm_rope for ONE of q / k: x = the projection's output (pre-norm) [rows][heads * dh] at ldx, dy = gradient of the
 * normalised + rotated heads; dx likewise; dw_partial: fp32 [ug_qk_rmsnorm_rope_bwd_partials(rows, heads)][dh], partial sums of d(un) * xhat
 * over disjoint sets of (row, head) vectors - their sum over the first index, taken by the caller, is d weight (deterministic: the assignment of
 * vectors to partial rows depends only on rows and heads). w NULL: no RMSNorm, dw_partial unused; cos NULL: no RoPE. dh <= 256.
 * Row r sits at position pos_offset + r % rows_per_batch; cos / sin tables [positions][dh] fp32 as in the forward. */
int64_t ug_qk_rmsnorm_rope_bwd_partials(int64_t rows, int32_t heads);
int ug_qk_rmsnorm_rope_bwd(const void* x, int64_t ldx, const void* dy, int64_t lddy, void* dx, int64_t lddx, void* dw_partial, const void* w,
                           const float* cos_tab, const float* sin_tab, int64_t rows, int64_t rows_per_batch, int64_t pos_offset, int32_t heads,
                           int32_t dh, float eps, ug_stream_t stream);
/* attention backward between its GEMMs (F.scaled_dot_product_attention, src/UniGenUtils.py:601): lse[r] = logsumexp(scale * S[r][:]);
 * P = exp(scale * S - lse) for the first valid_cols columns, 0 for the rest (zero-padded keys: sequence lengths are padded to the GEMM's
 * contraction granularity); delta[g][r] = sum_c dO[r][g*cols + c] * O[r][g*cols + c]; dS = scale * P * (dP - delta). S, dP fp32. */
int ug_row_lse(const float* S, int64_t ld, float* lse, int64_t rows, int64_t cols, float scale, ug_stream_t stream);
int ug_attn_prob(const float* S, int64_t ld_s, const float* lse, void* P, int64_t ld_p, int64_t rows, int64_t cols, int64_t valid_cols, float scale,
                 ug_stream_t stream);
int ug_attn_dscore(const void* P, int64_t ld_p, const float* dP, int64_t ld_dp, const float* delta, void* dS, int64_t ld_ds, int64_t rows, int64_t cols,
                   float scale, ug_stream_t stream);
int ug_rowdot(const void* a, int64_t lda, const void* b, int64_t ldb, float* out, int64_t rows, int64_t groups, int64_t cols, ug_stream_t stream);
/* F.scaled_dot_product_attention backward on the forward kernel's tiling (bf16): dq, dk, dv from q, k, v, o, dout in the forward's [batch][row][head*dh]
 * strided layout. Four launches of one kernel (row statistics lse, then dQ, dK, dV, each recomputing its score tiles; csrc/attention.hip) plus
 * delta = rowsum(dout * o). workspace: ug_flash_attn_bwd_workspace_bytes() bytes, 16-byte aligned (lse and delta, fp32 per (batch, head, query)).
 * The fp32 verification path keeps the GEMM-based formulation of unigen_amd/autograd.py. */
int64_t ug_flash_attn_bwd_workspace_bytes(int64_t batches, int32_t heads, int64_t Lq);
/* ug_flash_attn_fwd that also writes lse2[b][h][q] = log2 sum_k 2^(scale log2(e) s_qk) (fp32, row length lse_ld >= Lq, typically Lq rounded up to 64
 * with the padding zeroed by the caller): the forward of a training step; hand the buffer to ug_flash_attn_bwd as lse_in. */
int ug_flash_attn_fwd_lse(const void* q, int64_t q_row_stride, int64_t q_batch_stride, const void* k, int64_t k_row_stride, int64_t k_batch_stride,
                          const void* v, int64_t v_row_stride, int64_t v_batch_stride, void* o, int64_t o_row_stride, int64_t o_batch_stride,
                          int64_t batches, int32_t heads, int64_t Lq, int64_t Lkv, int32_t dh, float softmax_scale, float* lse2, int64_t lse_ld,
                          ug_stream_t stream);
int ug_flash_attn_bwd(const void* q, int64_t q_row_stride, int64_t q_batch_stride, const void* k, int64_t k_row_stride, int64_t k_batch_stride,
                      const void* v, int64_t v_row_stride, int64_t v_batch_stride, const void* o, int64_t o_row_stride, int64_t o_batch_stride,
                      const void* dout, int64_t do_row_stride, int64_t do_batch_stride, void* dq, int64_t dq_row_stride, int64_t dq_batch_stride,
                      void* dk, int64_t dk_row_stride, int64_t dk_batch_stride, void* dv, int64_t dv_row_stride, int64_t dv_batch_stride,
                      int64_t batches, int32_t heads, int64_t Lq, int64_t Lkv, int32_t dh, float softmax_scale,
                      const float* lse_in /* [batches][heads][Lq rounded up to 64], or NULL: recomputed */, void* workspace, int64_t workspace_bytes,
                      ug_stream_t stream);
int ug_transpose_f32(const void* src, int64_t ld_src, int64_t src_bstride, void* dst, int64_t ld_dst, int64_t dst_bstride, int64_t batch, int64_t rows,
                     int64_t cols, int64_t rows_pad, ug_stream_t stream);
int ug_colsum_f32(const void* a, int64_t lda, const void* b, int64_t ldb, void* out, int64_t ldo, int64_t rows, int64_t cols, int64_t rows_per_group,
                  float alpha, void* workspace, int64_t workspace_bytes, ug_stream_t stream);
int ug_gelu_tanh_f32(const void* x, void* y, int64_t n, ug_stream_t stream);
int ug_gelu_tanh_bwd_f32(const void* x, const void* dy, void* dx, int64_t n, ug_stream_t stream);
int ug_adaln_modulate_bwd_f32(const void* x, int64_t ldx, const void* dy, int64_t lddy, const void* scale, int64_t mod_ld, int64_t rows_per_sample,
                              void* dx, int64_t lddx, void* partials, int64_t rows, int64_t D, float eps, ug_stream_t stream);
int ug_qk_rmsnorm_rope_bwd_f32(const void* x, int64_t ldx, const void* dy, int64_t lddy, void* dx, int64_t lddx, void* dw_partial, const void* w,
                               const float* cos_tab, const float* sin_tab, int64_t rows, int64_t rows_per_batch, int64_t pos_offset, int32_t heads,
                               int32_t dh, float eps, ug_stream_t stream);
int ug_attn_prob_f32(const float* S, int64_t ld_s, const float* lse, void* P, int64_t ld_p, int64_t rows, int64_t cols, int64_t valid_cols, float scale,
                     ug_stream_t stream);
int ug_attn_dscore_f32(const void* P, int64_t ld_p, const float* dP, int64_t ld_dp, const float* delta, void* dS, int64_t ld_ds, int64_t rows,
                       int64_t cols, float scale, ug_stream_t stream);
int ug_rowdot_f32(const void* a, int64_t lda, const void* b, int64_t ldb, float* out, int64_t rows, int64_t groups, int64_t cols, ug_stream_t stream);

int ug_version(void);
const char* ug_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* UNIGEN_HIP_H */
