/* vcoder_kernels.h — per-kernel C entry points of libvcoder_hip.so (unit-test / micro-benchmark surface).
 *
 * Every function enqueues ONE gfx950 kernel on `stream` (NULL = default stream) over DEVICE pointers; bf16 tensors
 * are raw uint16 bit patterns.  The `-m gpu` parity tests call each kernel through these at the true shapes of
 * VCoder-DS LLaVA-1.5-7b and compare with oracle/cpu_ref.py.  Each replaces a torch/HF op of the reference's hot
 * path; the file:line it restates is given per entry.
 */
#ifndef VCODER_KERNELS_H
#define VCODER_KERNELS_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* nn.Linear / Conv2d-as-GEMM with fused epilogue: out = epi(A[M,K] . W[N,K]^T + bias).  epi: 0 bf16, 1 bf16 quick_gelu,
 * 2 bf16 erf-gelu, 3 fp32, 4 fp32 residual add in place, 5 SwiGLU over interleaved (gate,up) rows.
 * [HF] clip/modeling_clip.py:309-311,333,346-350; [HF] llama/modeling_llama.py:174-176,254-256,280;
 * vcoder_llava/model/multimodal_projector/builder.py:42-46 */
void vck_gemm(const uint16_t* A, const uint16_t* W, const float* bias, void* out, int M, int N, int K, int lda, int ldw,
              int ldo, int epi, void* stream);
/* same with an fp32 workspace (>= 64 MiB covers every shape): enables the deterministic split-K of a short last round of
 * 256x256 tiles (partials summed in k order by a fix-up launch) */
void vck_gemm_ws(const uint16_t* A, const uint16_t* W, const float* bias, void* out, int M, int N, int K, int lda, int ldw,
                 int ldo, int epi, float* ws, size_t ws_bytes, void* stream);
/* decode-time skinny GEMM (M<=16) over MFMA-fragment-packed weights.  epi: 0 bf16, 1 fp32, 2 fp32 residual, 3 SwiGLU */
void vck_gemv(const uint16_t* X, const uint16_t* Wp, void* out, int M, int N, int K, int ldo, int epi, void* stream);
/* general form.  RMSNorm ([HF] llama :53-70) is folded across producer and consumer instead of run as a pass:
 *   consumer  (ssq_in != NULL): X is xg = bf16(x * g) written by the producer and out = rstd[m] * (X @ W^T), with
 *             rstd[m] = rsqrt(sum_p ssq_in[m][p] / K + eps) from `npart` deterministic sum-of-squares partials;
 *   producer  (epi 2): ssq_out gets the partials of the updated residual rows and xg_out = bf16(residual * xg_w).
 * wscale != NULL: Wp holds W8A16 e4m3 bytes (vck_quantize_fp8) and wscale the per-output-row scales.
 * sk_scratch/sk_counters != NULL: deterministic split-K for matrices with few output tiles — `ksplit` workgroups per
 * tile (0 = let the launcher choose), partials [ksplit][N/16][2][256] fp32 summed in k order by the last arriver;
 * sk_counters [N/16][2] must be zero before the first launch (the kernel re-arms them).  M <= 32: rows 16..31 form a
 * second MFMA row group served by the same weight pass (the decode pool's steps). */
void vck_gemv_ex(const uint16_t* X, const void* Wp, const float* wscale, void* out, const float* ssq_in, float* ssq_out,
                 const float* xg_w, uint16_t* xg_out, int npart, float eps, float* sk_scratch, unsigned* sk_counters,
                 int ksplit, int M, int N, int K, int ldo, int epi, void* stream);
void vck_pack_weight(const uint16_t* W, uint16_t* Wp, int N, int K, void* stream);
/* W8A16 decode weights (BASELINE config C5): per-output-row power-of-two scale + OCP e4m3 bytes in the gemv's 64-wide
 * k super-tile order; W [N,K] bf16 is overwritten with the dequantised values (what the prefill GEMMs then read).
 * vck_gemv_ex streams the bytes. */
void vck_quantize_fp8(uint16_t* W, uint8_t* Wq, float* scale, int N, int K, void* stream);
/* the same, additionally leaving the bytes row-major in Wrow [N,K]: the weight operand of vck_gemm_f8 */
void vck_quantize_fp8_rows(uint16_t* W, uint8_t* Wq, float* scale, uint8_t* Wrow, int N, int K, void* stream);
/* W8A8 prefill GEMM (BASELINE config C5, "CDNA4 fp8 MFMA"): token rows of A [M,lda] bf16 -> e4m3 bytes Q [M,K] with a
 * per-row power-of-two scale (the rule of the weight rows); then
 *   out = epi( (Q @ Wrow^T) * a_scale[m] * w_scale[n] )
 * on v_mfma_scale_f32_16x16x128_f8f6f4 (unit block scales; twice the bf16 MFMA rate).  K % 128 == 0;
 * epi 0 bf16, 4 fp32 residual add, 5 SwiGLU.  ws: optional split-K workspace as for vck_gemm_ws. */
/* RMSNorm straight into that operand: the bytes / scales of vck_rmsnorm followed by vck_quant_act_rows, in one pass */
void vck_rmsnorm_q8(const float* x, const float* w, uint8_t* q, float* scale, int rows, int D, float eps, void* stream);
void vck_quant_act_rows(const uint16_t* A, int lda, uint8_t* Q, float* scale, int M, int K, void* stream);
void vck_gemm_f8(const uint8_t* A, const float* a_scale, const uint8_t* W, const float* w_scale, void* out, int M, int N,
                 int K, int ldo, int epi, float* ws, size_t ws_bytes, void* stream);
void vck_interleave_rows(const uint16_t* gate, const uint16_t* up, uint16_t* out, int F, int K, void* stream);
/* nn.LayerNorm ([HF] clip :370,379) and LlamaRMSNorm ([HF] llama :53-70); fp32 in, bf16 out */
void vck_layernorm(const float* x, const float* w, const float* b, uint16_t* y, int rows, int D, float eps, void* stream);
void vck_rmsnorm(const float* x, const int* row_idx, const float* w, uint16_t* y, int rows, int D, float eps, void* stream);
/* CLIPVisionEmbeddings ([HF] clip :202-218): im2col of the k=s=14 conv; CLS + position + pre_layrnorm */
void vck_im2col(const float* pixels, uint16_t* cols, int n_img, int image, int patch, int Kpad, void* stream);
void vck_vit_embed_ln(const float* patches, const float* cls, const float* pos, const float* w, const float* b, float* x,
                      int n_img, int T, int D, float eps, void* stream);
/* feature_select (multimodal_encoder/clip_encoder.py:29-37) */
void vck_select_rows_bf16(const float* x, uint16_t* y, int n_img, int T, int skip, int D, void* stream);
/* head split + rotate_half RoPE ([HF] llama :113-160,259-262): qkv bf16 [B*T, 3*H*hd] -> Q [B,H,q_stride,hd],
 * K [B,H,kv_stride,hd], V^T [B,H,hd,kv_stride] (the MFMA A operand of the flash kernel's P.V).  pos0_dev is unused.
 * V^T is a scratch between this kernel and vck_attention and uses the flash kernel's KEY ORDER: inside every aligned block of 32
 * keys, position 8c + e (c = 0..3, e = 0..7) holds key 4c + (e & 3) + 16 (e >> 2) (csrc/attn.hip vt_chunk_key0) — the order in
 * which a lane's softmax numerators are packed into the P operand, so one 16-byte LDS read is a whole MFMA fragment.  A caller
 * that fills V^T itself (vck_attention's vt) must use the same order; strides are multiples of 64. */
void vck_qkv_split(const uint16_t* qkv, uint16_t* q, uint16_t* k, uint16_t* vt, int B, int T, int H, int hd, int q_stride,
                   int kv_stride, const int* pos0_dev, const float* rope_cos, const float* rope_sin, void* stream);
/* the LLM prefill form: K and V rows into the key-major KV cache (kv_stride keys per (b,h)) — what the decode steps stream —
 * and V^T into a per-call scratch of vt_stride columns for this layer's flash attention */
void vck_qkv_split_kv(const uint16_t* qkv, uint16_t* q, uint16_t* k, uint16_t* v, uint16_t* vt, int B, int T, int H, int hd,
                      int q_stride, int kv_stride, int vt_stride, const float* rope_cos, const float* rope_sin, void* stream);
/* softmax(QK^T*scale [+causal]) V, fp32 softmax ([HF] clip :259-277; [HF] llama eager_attention_forward :191-214) */
void vck_attention(const uint16_t* q, const uint16_t* k, const uint16_t* vt, uint16_t* out, int B, int H, int T, int hd,
                   int q_stride, int kv_stride, int causal, float scale, void* stream);
/* decode step, one launch per layer: RoPE of the new q/k + append of the K and V rows + attention over the cache.
 * K and V are both key-major [B,H,kv_stride,hd] (kv_stride <= 4096: a row's scores sit in LDS). */
void vck_attention_decode_fused(const uint16_t* qkv, uint16_t* k, uint16_t* v, uint16_t* out, int B, int H, int hd,
                                int kv_stride, const int* pos_dev, const float* rope_cos, const float* rope_sin, float scale,
                                void* stream);
/* the same with one position per row (row b reads pos_rows[b * pos_stride]; rows with active_rows[b * pos_stride] == 0 are
 * skipped): rows of different requests — different prompt lengths and step counts — share one decode step */
void vck_attention_decode_rows(const uint16_t* qkv, uint16_t* k, uint16_t* v, uint16_t* out, int B, int H, int hd,
                               int kv_stride, const int* pos_rows, int pos_stride, const int* active_rows,
                               const float* rope_cos, const float* rope_sin, float scale, void* stream);
/* embedding gather + feature splice (vcoder_ds_llava_arch.py:173-276,305) */
void vck_splice(const int* row_src, int nrows, const uint16_t* embed, const uint16_t* feats, float* x, int D, void* stream);
void vck_embed_tokens(const int* tok, const uint16_t* embed, float* x, int B, int D, void* stream);
/* greedy select with EOS/pad bookkeeping ([HF] generation/utils.py:2894,2925-2929) */
void vck_greedy(const float* logits, int* next_tok, int* out_ids, int* finished, int* step_dev, int B, int V, int max_new,
                int eos_id, int pad_id, void* stream);
/* tail of a decode step, one workgroup per row (csrc/select.hip): token selection — greedy, or temperature / top-k / top-p
 * sampling ([HF] generation/logits_process.py warpers + multinomial; serve/cli.py:122-132, serve/chat.py:141-151) — EOS /
 * pad / keyword-stop bookkeeping (mm_utils.py:128-151), embedding of the selected token (fp32 row x, RMSNorm partials ssq,
 * first GEMV operand xg = bf16(x * xg_w)) and the row's step / position advance (bit 0 / bit 1 of `advance`).  Every
 * per-row parameter lives in the row's record of `rows` ([nrows][vck_row_state_stride()] ints, csrc/kernels.h
 * RowStateField); row r writes its id to out_ids[rows[r][RS_OUT_OFF] + step]. */
void vck_select_embed(const float* logits, int ldl, int* rows, int* next_tok, int* out_ids, const uint16_t* embed, float* x,
                      float* ssq, const float* xg_w, uint16_t* xg, int D, int npart, int V, int nrows, int advance,
                      void* stream);
int vck_row_state_stride(void);
/* test hook of the sampler: u[i] = the uniform in (0,1) the Gumbel-max draw derives from 32-bit hash h[i] (strictly
 * inside the interval for EVERY h, so the Gumbel term -log(-log(u)) written to gumbel[i] is finite) */
void vck_uniform_probe(const uint32_t* h, float* u, float* gumbel, int n, void* stream);
void vck_embed_tokens_ssq(const int* tok, const uint16_t* embed, float* x, float* ssq, const float* xg_w, uint16_t* xg, int B,
                          int D, int npart, void* stream);
void vck_advance(int* step_dev, int* pos_dev, int* ctx_dev, void* stream);
/* strict (fp32-faithful) kernels behind vc_model_set_precision(m, 1): fp32 activations x bf16 weights on the exact
 * v_mfma_f32_16x16x4_f32, fp32 attention, fp32 head split + RoPE.  epi ids as vck_gemm (all outputs fp32). */
void vck_gemm_f32(const float* A, const uint16_t* W, const float* bias, float* out, int M, int N, int K, int lda, int ldw,
                  int ldo, int epi, void* stream);
void vck_attention_f32(const float* q, const float* k, const float* v, float* out, int B, int H, int Tq, int hd, int q_stride,
                       int kv_stride, int causal, int Tk, const int* pos0_dev, float scale, void* stream);
void vck_qkv_rope_f32(const float* qkv, float* q, float* k, float* v, int B, int T, int H, int hd, int q_stride, int kv_stride,
                      const int* pos0_dev, const float* rope_cos, const float* rope_sin, void* stream);
/* ---- precision mode "split" behind vc_model_set_precision(m, 2): the FAST kernels with every MFMA operand carried as two bf16
 * values, x = hi + lo (hi = bf16(x), lo = bf16(x - hi): ~16 mantissa bits; weights are exactly bf16), fp32 everywhere between
 * two MFMAs.  Meets the "logits within 1e-3, greedy ids bit-exact" bar of BASELINE.json at ~half the fast path's MFMA rate.
 * vck_gemm_split: A is [M, lda] with columns [0, Kw) = hi and [Kw, 2 Kw) = lo of the activation row; W [N, Kw] is contracted
 *   against both halves (its k index wraps); epi as vck_gemm; split_out > 0: bf16-valued epilogues write [hi | lo] again, the lo
 *   plane split_out columns to the right of the hi value.
 * vck_gemv_split: X is [G + M, K] — rows [0, M) hi, rows [G, G + M) lo (G = 8 for M <= 8, 16 for M <= 16); bf16-valued outputs
 *   (epi 0 / 3, xg_out) come back as hi row m + lo row G + m.
 * vck_*norm_split: normalised row r -> hi at y + r * ldy, lo at y + r * ldy + lo_off.
 * vck_qkv_split32: fp32 fused-QKV rows -> RoPE in fp32 -> fp32 K / V cache rows (k32 / v32, may be NULL) + bf16 hi / lo planes
 *   of Q [B,H,q_stride,hd], K [B,H,ks_stride,hd] and V^T [B,H,hd,vt_stride] (key order of vck_qkv_split's V^T).
 * vck_attention_split: flash attention over those planes (3 MFMAs per product), output rows [hi | lo]: stride ldo, lo at lo_off.
 * vck_attention_decode_kv32: the fused decode attention over fp32 qkv / fp32 K, V caches; output as stacked groups of G hi rows
 *   + G lo rows (row b -> hi at row (b / G) * 2G + b % G). */
void vck_gemm_split(const uint16_t* A, const uint16_t* W, const float* bias, void* out, int M, int N, int Kw, int lda, int ldo,
                    int epi, int split_out, float* ws, size_t ws_bytes, void* stream);
void vck_gemv_split(const uint16_t* X, const void* Wp, const float* wscale, void* out, const float* ssq_in, float* ssq_out,
                    const float* xg_w, uint16_t* xg_out, int npart, float eps, int M, int N, int K, int ldo, int epi, int G,
                    void* stream);
/* the decode GEMV with every argument: G = split rows (0: bf16 step; 8 / 16 / 32: X holds hi rows [0, M) and lo rows [G, G + M)),
 * the split-K buffers and their capacity (floats / counters; 0 = the historical [4][512][2][256] / [512][2]) */
void vck_gemv_full(const uint16_t* X, const void* Wp, const float* wscale, void* out, const float* ssq_in, float* ssq_out,
                   const float* xg_w, uint16_t* xg_out, int npart, float eps, float* sk_scratch, unsigned long long sk_scratch_floats,
                   unsigned* sk_counters, int sk_counters_n, int ksplit, int M, int N, int K, int ldo, int epi, int G, void* stream);
/* Inexact checkpoints (fp16 / fp32 values bf16 cannot hold; vc_model_inexact_tensors): w = bf16 hi + bf16 lo.
 *  vck_f32_to_bf16_planes: hi = bf16(in), lo = bf16(in - hi) (lo may be NULL), *inexact = 1 if any in != hi
 *  vck_gemm_f32_wlo:   the strict GEMM over W + W_lo
 *  vck_gemm_split_wlo: the split prefill GEMM with a third K segment a_hi . w_lo (W_lo [N, Kw] like W; NULL = vck_gemm_split)
 *  vck_gemv_split_wlo: the split decode GEMV (G > 0, workgroup-shared form) with the packed lo plane */
void vck_f32_to_bf16_planes(const float* in, uint16_t* hi, uint16_t* lo, uint64_t n, unsigned* inexact, void* stream);
void vck_gemm_f32_wlo(const float* A, const uint16_t* W, const uint16_t* W_lo, const float* bias, float* out, int M, int N, int K,
                      int lda, int ldw, int ldo, int epi, void* stream);
void vck_gemm_split_wlo(const uint16_t* A, const uint16_t* W, const uint16_t* W_lo, const float* bias, void* out, int M, int N, int Kw,
                        int lda, int ldo, int epi, int split_out, float* ws, size_t ws_bytes, void* stream);
void vck_gemv_split_wlo(const uint16_t* X, const void* Wp, const void* Wp_lo, void* out, const float* ssq_in, float* ssq_out,
                        const float* xg_w, uint16_t* xg_out, int npart, float eps, float* sk_scratch, unsigned long long sk_scratch_floats,
                        unsigned* sk_counters, int sk_counters_n, int ksplit, int M, int N, int K, int ldo, int epi, int G, void* stream);
/* which kernel serves the decode GEMV of precision mode "split": 0 = per-wave rings (gemv_dma_kernel; two weight passes of 16
 * rows per 32-row step), 1 / -1 (default) = workgroup-shared activation chunks (gemv_wg_kernel; hi + lo planes in one pass) with
 * four or six tiles per workgroup as the launcher chooses per matrix; 2 = four everywhere (A/B), 3 = six everywhere (tests) */
void vck_set_gemv_variant(int v);
/* K12 + K13 + K14 in one launch (round 6): the fused-QKV projection of a prefill layer whose epilogue applies RoPE, splits the heads
 * and writes Q [B,H,q_stride,128], roped K rows and V rows [B,H,kv_stride,128] (and / or their e4m3 cache rows k8 / v8
 * [B,H,kv8_stride,128]) and the V^T scratch [B,H,128,vt_stride] in the flash kernel's key order — bit for bit what vck_gemm
 * (EPI_BF16) + vck_qkv_split_kv / _kv8 produce.  A [B * T, lda] bf16, W [3 * H * 128, K] bf16 (HF q / k / v rows stacked); f8 != 0:
 * A and W are e4m3 bytes with a_scale [B * T] / w_scale [3 * H * 128].  (H * 128) % 256 == 0.  ws: optional split-K workspace. */
void vck_gemm_qkv(const void* A, const float* a_scale, const void* W, const float* w_scale, const float* bias, int B, int T, int H, int K,
                  int lda, uint16_t* q, uint16_t* k, uint16_t* v, uint16_t* vt, uint8_t* k8, uint8_t* v8, int q_stride, int kv_stride,
                  int vt_stride, int kv8_stride, const float* rope_cos, const float* rope_sin, int f8, float* ws, size_t ws_bytes,
                  void* stream);
/* overrides VC_GEMM_VARIANT inside one process (< 0: back to the environment's value): 1 = default (8-phase 256 x 256 for large
 * problems on v_mfma_f32_16x16x32_bf16), 6 = the same schedule on v_mfma_f32_32x32x16_bf16, 7 = that form for every size */
void vck_set_gemm_variant(int v);
/* NT = ceil(tiles / 256) tiles per workgroup for bf16 matrices of more than 512 tiles, one deep-ringed workgroup per CU (NT in
 * 3, 4, 6, 7): -1 / 1 = the classes that measured faster (default), 0 = off, 2 = every class.  Results are bit-identical
 * whichever is set. */
void vck_set_gemv_wide(int v);
unsigned long long vck_gemv_wide_launches(void);
unsigned long long vck_gemv_wg_launches(void);   /* launches the workgroup-shared form has served (tests) */
void vck_rmsnorm_split(const float* x, const int* row_idx, const float* w, uint16_t* y, int rows, int D, float eps, int ldy,
                       uint64_t lo_off, void* stream);
void vck_layernorm_split(const float* x, const float* w, const float* b, uint16_t* y, int rows, int D, float eps, int ldy,
                         uint64_t lo_off, void* stream);
void vck_qkv_split32(const float* qkv, uint16_t* q_hi, uint16_t* q_lo, uint16_t* k_hi, uint16_t* k_lo, uint16_t* vt_hi,
                     uint16_t* vt_lo, float* k32, float* v32, int B, int T, int H, int hd, int q_stride, int ks_stride, int vt_stride,
                     int kv_stride, const float* rope_cos, const float* rope_sin, void* stream);
void vck_attention_split(const uint16_t* q_hi, const uint16_t* q_lo, const uint16_t* k_hi, const uint16_t* k_lo,
                         const uint16_t* vt_hi, const uint16_t* vt_lo, uint16_t* out, int B, int H, int T, int hd, int q_stride,
                         int kv_stride, int causal, float scale, int ldo, int lo_off, void* stream);
void vck_attention_decode_kv32(const float* qkv, float* k, float* v, uint16_t* out, int B, int H, int hd, int kv_stride,
                               const int* pos_rows, int pos_stride, const int* active_rows, const float* rope_cos,
                               const float* rope_sin, float scale, int G, void* stream);
/* fp24 KV caches of precision mode "split" (rows of hd x u16 | hd x u8: the top 24 bits of the fp32 values, RNE — 0.75 of the fp32
 * bytes at 2^-17 relative precision): the writer of a prefill and the fused decode attention over them */
void vck_qkv_split24(const float* qkv, uint16_t* q_hi, uint16_t* q_lo, uint16_t* k_hi, uint16_t* k_lo, uint16_t* vt_hi,
                     uint16_t* vt_lo, void* k24, void* v24, int B, int T, int H, int hd, int q_stride, int ks_stride, int vt_stride,
                     int kv_stride, const float* rope_cos, const float* rope_sin, void* stream);
void vck_attention_decode_kv24(const float* qkv, void* k, void* v, uint16_t* out, int B, int H, int hd, int kv_stride,
                               const int* pos_rows, int pos_stride, const int* active_rows, const float* rope_cos,
                               const float* rope_sin, float scale, int G, void* stream);
/* e4m3 KV caches of the fp8 weight format (rows of hd bytes, no scale, saturating at 448): the prefill's writer (the bf16 K rows go to
 * a per-call scratch for the flash kernel) and the bf16 decode step's fused attention over them */
void vck_qkv_split_kv8(const uint16_t* qkv, uint16_t* q, uint16_t* k, uint8_t* k8, uint8_t* v8, uint16_t* vt, int B, int T, int H, int hd,
                       int q_stride, int kv_stride, int vt_stride, int kv8_stride, const float* rope_cos, const float* rope_sin,
                       void* stream);
void vck_attention_decode_kv8(const uint16_t* qkv, uint8_t* k, uint8_t* v, uint16_t* out, int B, int H, int hd, int kv_stride,
                              const int* pos_rows, int pos_stride, const int* active_rows, const float* rope_cos,
                              const float* rope_sin, float scale, void* stream);
/* deterministic synthetic tensors (vcoder_amd/synth.py) and dtype converts */
void vck_synth_bf16(uint16_t* out, uint64_t n, uint32_t tseed, float offset, float halfwidth, void* stream);
void vck_synth_f32(float* out, uint64_t n, uint32_t tseed, float offset, float halfwidth, void* stream);
/* the generator's value before the bf16 rounding, as an fp16 checkpoint holds it (rounding 1) or unrounded fp32 (2):
 * vcoder_amd/synth.py synth_tensor(rounding="fp16" | "fp32") */
void vck_synth_f32_rounded(float* out, uint64_t n, uint32_t tseed, float offset, float halfwidth, int rounding, void* stream);
void vck_f32_to_bf16(const float* in, uint16_t* out, uint64_t n, void* stream);
void vck_bf16_to_f32(const uint16_t* in, float* out, uint64_t n, void* stream);

#ifdef __cplusplus
}
#endif
#endif
