// kernel_api.cpp — C-ABI entry points over the individual kernel launchers (declared in
// include/vcoder_kernels.h).  The `-m gpu` parity tests call every kernel through these with device
// pointers; the CPU emulator build (tests/emu) exports the same symbols over host pointers.
#include "kernels.h"
using namespace vc;

#define VCK_EXPORT extern "C" __attribute__((visibility("default")))
static inline hipStream_t S(void* s) { return (hipStream_t)s; }

VCK_EXPORT void vck_gemm(const uint16_t* A, const uint16_t* W, const float* bias, void* out, int M, int N, int K,
                         int lda, int ldw, int ldo, int epi, void* stream) {
    GemmArgs a{A, W, bias, out, M, N, K, lda, ldw, ldo};
    launch_gemm(a, epi, S(stream));
}
VCK_EXPORT void vck_gemm_ws(const uint16_t* A, const uint16_t* W, const float* bias, void* out, int M, int N, int K,
                            int lda, int ldw, int ldo, int epi, float* ws, size_t ws_bytes, void* stream) {
    GemmArgs a{A, W, bias, out, M, N, K, lda, ldw, ldo};
    a.ws = ws;
    a.ws_bytes = ws_bytes;
    launch_gemm(a, epi, S(stream));
}
VCK_EXPORT void vck_gemm_f8(const uint8_t* A, const float* a_scale, const uint8_t* W, const float* w_scale, void* out, int M,
                            int N, int K, int ldo, int epi, float* ws, size_t ws_bytes, void* stream) {
    GemmArgs a{reinterpret_cast<const bf16_t*>(A), reinterpret_cast<const bf16_t*>(W), nullptr, out, M, N, K, K, K, ldo};
    a.ws = ws;
    a.ws_bytes = ws_bytes;
    a.f8 = 1;
    a.a_scale = a_scale;
    a.w_scale = w_scale;
    launch_gemm(a, epi, S(stream));
}
/* the QKV GEMM of a prefill layer with RoPE + head split + KV write in its epilogue (EPI_QKV; kernels.h QkvEpiArgs): A [B * T, lda]
 * bf16 (or e4m3 rows + a_scale with f8 != 0: W then e4m3 [3 D, K] + w_scale), W [3 D, K]; hd = 128 */
VCK_EXPORT void vck_gemm_qkv(const void* A, const float* a_scale, const void* W, const float* w_scale, const float* bias, int B, int T,
                             int H, int K, int lda, uint16_t* q, uint16_t* k, uint16_t* v, uint16_t* vt, uint8_t* k8, uint8_t* v8,
                             int q_stride, int kv_stride, int vt_stride, int kv8_stride, const float* rope_cos, const float* rope_sin,
                             int f8, float* ws, size_t ws_bytes, void* stream) {
    const int Tp = (T + 31) / 32 * 32;
    GemmArgs a{reinterpret_cast<const bf16_t*>(A), reinterpret_cast<const bf16_t*>(W), bias, nullptr, B * Tp, 3 * H * 128, K, lda, K, 0};
    a.ws = ws;
    a.ws_bytes = ws_bytes;
    a.f8 = f8;
    a.a_scale = a_scale;
    a.w_scale = w_scale;
    a.qe = QkvEpiArgs{q, k, v, vt, k8, v8, rope_cos, rope_sin, B, T, Tp, H, q_stride, kv_stride, vt_stride, kv8_stride};
    launch_gemm(a, EPI_QKV, S(stream));
}
VCK_EXPORT void vck_quant_act_rows(const uint16_t* A, int lda, uint8_t* Q, float* scale, int M, int K, void* stream) {
    launch_quant_act_rows(A, lda, Q, scale, M, K, S(stream));
}
VCK_EXPORT void vck_gemv(const uint16_t* X, const uint16_t* Wp, void* out, int M, int N, int K, int ldo, int epi,
                         void* stream) {
    GemvArgs a{};
    a.X = X; a.Wp = Wp; a.out = out; a.M = M; a.N = N; a.K = K; a.ldo = ldo;
    launch_gemv(a, epi, S(stream));
}
VCK_EXPORT void vck_gemv_ex(const uint16_t* X, const void* Wp, const float* wscale, void* out, const float* ssq_in,
                            float* ssq_out, const float* xg_w, uint16_t* xg_out, int npart, float eps, float* sk_scratch,
                            unsigned* sk_counters, int ksplit, int M, int N, int K, int ldo, int epi, void* stream) {
    GemvArgs a{};
    a.X = X; a.Wp = reinterpret_cast<const uint16_t*>(Wp); a.wscale = wscale; a.out = out; a.M = M; a.N = N; a.K = K; a.ldo = ldo;
    a.ssq_in = ssq_in; a.ssq_out = ssq_out; a.xg_w = xg_w; a.xg_out = xg_out; a.npart = npart; a.eps = eps;
    a.sk_scratch = sk_scratch; a.sk_counters = sk_counters; a.ksplit = ksplit;
    launch_gemv(a, epi, S(stream));
}
VCK_EXPORT void vck_attention_decode_fused(const uint16_t* qkv, uint16_t* k, uint16_t* v, uint16_t* out, int B, int H,
                                           int hd, int kv_stride, const int* pos_dev, const float* rope_cos,
                                           const float* rope_sin, float scale, void* stream) {
    AttnDecodeFusedArgs a{qkv, k, v, out, B, H, hd, kv_stride, pos_dev, rope_cos, rope_sin, scale, 0, nullptr};
    launch_attention_decode_fused(a, S(stream));
}
VCK_EXPORT void vck_attention_decode_rows(const uint16_t* qkv, uint16_t* k, uint16_t* v, uint16_t* out, int B, int H, int hd,
                                          int kv_stride, const int* pos_rows, int pos_stride, const int* active_rows,
                                          const float* rope_cos, const float* rope_sin, float scale, void* stream) {
    AttnDecodeFusedArgs a{qkv, k, v, out, B, H, hd, kv_stride, pos_rows, rope_cos, rope_sin, scale, pos_stride, active_rows};
    launch_attention_decode_fused(a, S(stream));
}
VCK_EXPORT void vck_select_embed(const float* logits, int ldl, int* rows, int* next_tok, int* out_ids, const uint16_t* embed,
                                 float* x, float* ssq, const float* xg_w, uint16_t* xg, int D, int npart, int V, int nrows,
                                 int advance, void* stream) {
    SelectArgs a{logits, ldl, rows, next_tok, out_ids, embed, x, ssq, xg_w, xg, D, npart, V, nrows, advance, 0, 0};
    launch_select_embed(a, S(stream));
}
VCK_EXPORT int vck_row_state_stride() { return RS_STRIDE; }
VCK_EXPORT void vck_uniform_probe(const uint32_t* h, float* u, float* gumbel, int n, void* stream) {
    launch_uniform_probe(h, u, gumbel, n, S(stream));
}
VCK_EXPORT void vck_embed_tokens_ssq(const int* tok, const uint16_t* embed, float* x, float* ssq, const float* xg_w,
                                     uint16_t* xg, int B, int D, int npart, void* stream) {
    launch_embed_tokens_ssq(tok, embed, x, ssq, xg_w, xg, B, D, npart, S(stream));
}
VCK_EXPORT void vck_quantize_fp8(uint16_t* W, uint8_t* Wq, float* scale, int N, int K, void* stream) {
    launch_quantize_fp8(W, Wq, scale, N, K, S(stream));
}
VCK_EXPORT void vck_quantize_fp8_rows(uint16_t* W, uint8_t* Wq, float* scale, uint8_t* Wrow, int N, int K, void* stream) {
    launch_quantize_fp8(W, Wq, scale, N, K, S(stream), Wrow);
}
VCK_EXPORT void vck_pack_weight(const uint16_t* W, uint16_t* Wp, int N, int K, void* stream) {
    launch_pack_weight(W, Wp, N, K, S(stream));
}
VCK_EXPORT void vck_interleave_rows(const uint16_t* g, const uint16_t* u, uint16_t* out, int F, int K, void* stream) {
    launch_interleave_rows(g, u, out, F, K, S(stream));
}
VCK_EXPORT void vck_layernorm(const float* x, const float* w, const float* b, uint16_t* y, int rows, int D, float eps,
                              void* stream) {
    launch_layernorm(x, w, b, y, rows, D, eps, S(stream));
}
VCK_EXPORT void vck_rmsnorm(const float* x, const int* row_idx, const float* w, uint16_t* y, int rows, int D, float eps,
                            void* stream) {
    if (row_idx) launch_rmsnorm_rows(x, row_idx, w, y, rows, D, eps, S(stream));
    else launch_rmsnorm(x, w, y, rows, D, eps, S(stream));
}
VCK_EXPORT void vck_rmsnorm_q8(const float* x, const float* w, uint8_t* q, float* scale, int rows, int D, float eps, void* stream) {
    launch_rmsnorm_q8(x, w, q, scale, rows, D, eps, S(stream));
}
VCK_EXPORT void vck_im2col(const float* pixels, uint16_t* cols, int n_img, int image, int patch, int Kpad, void* stream) {
    launch_im2col(pixels, cols, n_img, image, patch, Kpad, S(stream));
}
VCK_EXPORT void vck_vit_embed_ln(const float* patches, const float* cls, const float* pos, const float* w,
                                 const float* b, float* x, int n_img, int T, int D, float eps, void* stream) {
    launch_vit_embed_ln(patches, cls, pos, w, b, x, n_img, T, D, eps, S(stream));
}
VCK_EXPORT void vck_select_rows_bf16(const float* x, uint16_t* y, int n_img, int T, int skip, int D, void* stream) {
    launch_select_rows_bf16(x, y, n_img, T, skip, D, S(stream));
}
VCK_EXPORT void vck_qkv_split(const uint16_t* qkv, uint16_t* q, uint16_t* k, uint16_t* vt, int B, int T, int H, int hd,
                              int q_stride, int kv_stride, const int* pos0_dev, const float* rope_cos,
                              const float* rope_sin, void* stream) {
    QkvSplitArgs a{qkv, q, k, vt, B, T, H, hd, q_stride, kv_stride, nullptr, rope_cos, rope_sin, nullptr, 0};
    (void)pos0_dev;
    launch_qkv_split(a, S(stream));
}
VCK_EXPORT void vck_qkv_split_kv(const uint16_t* qkv, uint16_t* q, uint16_t* k, uint16_t* v, uint16_t* vt, int B, int T, int H,
                                 int hd, int q_stride, int kv_stride, int vt_stride, const float* rope_cos,
                                 const float* rope_sin, void* stream) {
    QkvSplitArgs a{qkv, q, k, vt, B, T, H, hd, q_stride, kv_stride, nullptr, rope_cos, rope_sin, v, vt_stride};
    launch_qkv_split(a, S(stream));
}
/* the prefill's split + RoPE writing an e4m3 KV cache (k8 / v8 [B,H,kv8_stride,hd] bytes; the fp8 weight format): K bf16 rows go
 * to the per-call scratch `k` (stride kv_stride) the flash kernel reads */
VCK_EXPORT void vck_qkv_split_kv8(const uint16_t* qkv, uint16_t* q, uint16_t* k, uint8_t* k8, uint8_t* v8, uint16_t* vt, int B, int T,
                                  int H, int hd, int q_stride, int kv_stride, int vt_stride, int kv8_stride, const float* rope_cos,
                                  const float* rope_sin, void* stream) {
    QkvSplitArgs a{qkv, q, k, vt, B, T, H, hd, q_stride, kv_stride, nullptr, rope_cos, rope_sin, nullptr, vt_stride, k8, v8, kv8_stride};
    launch_qkv_split(a, S(stream));
}
/* the fused decode attention of the bf16 step over e4m3 caches */
VCK_EXPORT void vck_attention_decode_kv8(const uint16_t* qkv, uint8_t* k, uint8_t* v, uint16_t* out, int B, int H, int hd, int kv_stride,
                                         const int* pos_rows, int pos_stride, const int* active_rows, const float* rope_cos,
                                         const float* rope_sin, float scale, void* stream) {
    AttnDecodeFusedArgs a{qkv, reinterpret_cast<uint16_t*>(k), reinterpret_cast<uint16_t*>(v), out, B, H, hd, kv_stride, pos_rows,
                          rope_cos, rope_sin, scale, pos_stride, active_rows, 3};
    launch_attention_decode_fused(a, S(stream));
}
VCK_EXPORT void vck_attention(const uint16_t* q, const uint16_t* k, const uint16_t* vt, uint16_t* out, int B, int H,
                              int T, int hd, int q_stride, int kv_stride, int causal, float scale, void* stream) {
    AttnArgs a{q, k, vt, out, B, H, T, hd, q_stride, kv_stride, causal, scale};
    launch_attention(a, S(stream));
}
VCK_EXPORT void vck_splice(const int* row_src, int nrows, const uint16_t* embed, const uint16_t* feats, float* x, int D,
                           void* stream) {
    launch_splice(row_src, nrows, embed, feats, x, D, S(stream));
}
VCK_EXPORT void vck_embed_tokens(const int* tok, const uint16_t* embed, float* x, int B, int D, void* stream) {
    launch_embed_tokens(tok, embed, x, B, D, S(stream));
}
VCK_EXPORT void vck_greedy(const float* logits, int* next_tok, int* out_ids, int* finished, int* step_dev, int B, int V,
                           int max_new, int eos_id, int pad_id, void* stream) {
    GreedyArgs a{logits, next_tok, out_ids, finished, step_dev, B, V, max_new, eos_id, pad_id};
    launch_greedy(a, S(stream));
}
VCK_EXPORT void vck_advance(int* step_dev, int* pos_dev, int* ctx_dev, void* stream) {
    launch_advance(step_dev, pos_dev, ctx_dev, S(stream));
}
VCK_EXPORT void vck_synth_bf16(uint16_t* out, uint64_t n, uint32_t tseed, float offset, float halfwidth, void* stream) {
    launch_synth_bf16(out, (size_t)n, tseed, offset, halfwidth, S(stream));
}
VCK_EXPORT void vck_synth_f32_rounded(float* out, uint64_t n, uint32_t tseed, float offset, float halfwidth, int rounding, void* stream) {
    launch_synth_f32_rounded(out, (size_t)n, tseed, offset, halfwidth, rounding, S(stream));
}
VCK_EXPORT void vck_synth_f32(float* out, uint64_t n, uint32_t tseed, float offset, float halfwidth, void* stream) {
    launch_synth_f32(out, (size_t)n, tseed, offset, halfwidth, S(stream));
}
VCK_EXPORT void vck_f32_to_bf16(const float* in, uint16_t* out, uint64_t n, void* stream) {
    launch_f32_to_bf16(in, out, (size_t)n, S(stream));
}
VCK_EXPORT void vck_bf16_to_f32(const uint16_t* in, float* out, uint64_t n, void* stream) {
    launch_bf16_to_f32(in, out, (size_t)n, S(stream));
}

// ---- strict (fp32) kernels -----------------------------------------------------------------------------------------
VCK_EXPORT void vck_gemm_f32(const float* A, const uint16_t* W, const float* bias, float* out, int M, int N, int K, int lda,
                             int ldw, int ldo, int epi, void* stream) {
    GemmF32Args a{A, W, bias, out, M, N, K, lda, ldw, ldo};
    launch_gemm_f32(a, epi, S(stream));
}
/* ... with the lo plane of an inexact checkpoint's weight: w = W + W_lo */
VCK_EXPORT void vck_gemm_f32_wlo(const float* A, const uint16_t* W, const uint16_t* W_lo, const float* bias, float* out, int M, int N,
                                 int K, int lda, int ldw, int ldo, int epi, void* stream) {
    GemmF32Args a{A, W, bias, out, M, N, K, lda, ldw, ldo};
    a.W_lo = W_lo;
    launch_gemm_f32(a, epi, S(stream));
}
VCK_EXPORT void vck_f32_to_bf16_planes(const float* in, uint16_t* hi, uint16_t* lo, uint64_t n, unsigned* inexact, void* stream) {
    launch_f32_to_bf16_planes(in, hi, lo, (size_t)n, inexact, S(stream));
}
VCK_EXPORT void vck_attention_f32(const float* q, const float* k, const float* v, float* out, int B, int H, int Tq, int hd,
                                  int q_stride, int kv_stride, int causal, int Tk, const int* pos0_dev, float scale,
                                  void* stream) {
    AttnF32Args a{q, k, v, out, B, H, Tq, hd, q_stride, kv_stride, causal, Tk, pos0_dev, scale};
    launch_attention_f32(a, S(stream));
}
VCK_EXPORT void vck_qkv_rope_f32(const float* qkv, float* q, float* k, float* v, int B, int T, int H, int hd, int q_stride,
                                 int kv_stride, const int* pos0_dev, const float* rope_cos, const float* rope_sin,
                                 void* stream) {
    QkvF32Args a{qkv, q, k, v, B, T, H, hd, q_stride, kv_stride, pos0_dev, rope_cos, rope_sin};
    launch_qkv_rope_f32(a, S(stream));
}

// ---- precision mode "split" (DESIGN.md section 5b): every MFMA operand as bf16 hi + lo -------------------------------------
VCK_EXPORT void vck_gemm_split(const uint16_t* A, const uint16_t* W, const float* bias, void* out, int M, int N, int Kw, int lda,
                               int ldo, int epi, int split_out, float* ws, size_t ws_bytes, void* stream) {
    GemmArgs a{A, W, bias, out, M, N, 2 * Kw, lda, Kw, ldo};
    a.kwrap = Kw / 64;
    a.split_out = split_out;
    a.ws = ws;
    a.ws_bytes = ws_bytes;
    launch_gemm(a, epi, S(stream));
}
VCK_EXPORT void vck_gemv_split(const uint16_t* X, const void* Wp, const float* wscale, void* out, const float* ssq_in,
                               float* ssq_out, const float* xg_w, uint16_t* xg_out, int npart, float eps, int M, int N, int K,
                               int ldo, int epi, int G, void* stream) {
    GemvArgs a{};
    a.X = X; a.Wp = reinterpret_cast<const uint16_t*>(Wp); a.wscale = wscale; a.out = out; a.M = M; a.N = N; a.K = K; a.ldo = ldo;
    a.ssq_in = ssq_in; a.ssq_out = ssq_out; a.xg_w = xg_w; a.xg_out = xg_out; a.npart = npart; a.eps = eps;
    a.split_rows = G;
    launch_gemv(a, epi, S(stream));
}
/* ... with the lo plane of an inexact checkpoint's weight ([N, Kw] like W): a third K segment a_hi . w_lo (K = 3 Kw) */
VCK_EXPORT void vck_gemm_split_wlo(const uint16_t* A, const uint16_t* W, const uint16_t* W_lo, const float* bias, void* out, int M,
                                   int N, int Kw, int lda, int ldo, int epi, int split_out, float* ws, size_t ws_bytes, void* stream) {
    GemmArgs a{A, W, bias, out, M, N, (W_lo ? 3 : 2) * Kw, lda, Kw, ldo};
    a.kwrap = Kw / 64;
    a.split_out = split_out;
    if (W_lo) a.w_lo_off = (long long)(reinterpret_cast<const char*>(W_lo) - reinterpret_cast<const char*>(W));
    a.ws = ws;
    a.ws_bytes = ws_bytes;
    launch_gemm(a, epi, S(stream));
}
/* the split decode GEMV (G > 0) with the packed lo plane of an inexact checkpoint's weight */
VCK_EXPORT void vck_gemv_split_wlo(const uint16_t* X, const void* Wp, const void* Wp_lo, void* out, const float* ssq_in, float* ssq_out,
                                   const float* xg_w, uint16_t* xg_out, int npart, float eps, float* sk_scratch,
                                   unsigned long long sk_scratch_floats, unsigned* sk_counters, int sk_counters_n, int ksplit, int M,
                                   int N, int K, int ldo, int epi, int G, void* stream) {
    GemvArgs a{};
    a.X = X; a.Wp = reinterpret_cast<const uint16_t*>(Wp); a.Wp_lo = reinterpret_cast<const uint16_t*>(Wp_lo); a.out = out;
    a.M = M; a.N = N; a.K = K; a.ldo = ldo;
    a.ssq_in = ssq_in; a.ssq_out = ssq_out; a.xg_w = xg_w; a.xg_out = xg_out; a.npart = npart; a.eps = eps;
    a.sk_scratch = sk_scratch; a.sk_counters = sk_counters; a.ksplit = ksplit;
    a.sk_scratch_floats = (size_t)sk_scratch_floats; a.sk_counters_n = sk_counters_n;
    a.split_rows = G;
    launch_gemv(a, epi, S(stream));
}
/* the decode GEMV with every argument (tests / tools): split rows G (0 = bf16 step), split-K buffers with their capacity */
VCK_EXPORT void vck_gemv_full(const uint16_t* X, const void* Wp, const float* wscale, void* out, const float* ssq_in,
                              float* ssq_out, const float* xg_w, uint16_t* xg_out, int npart, float eps, float* sk_scratch,
                              unsigned long long sk_scratch_floats, unsigned* sk_counters, int sk_counters_n, int ksplit, int M,
                              int N, int K, int ldo, int epi, int G, void* stream) {
    GemvArgs a{};
    a.X = X; a.Wp = reinterpret_cast<const uint16_t*>(Wp); a.wscale = wscale; a.out = out; a.M = M; a.N = N; a.K = K; a.ldo = ldo;
    a.ssq_in = ssq_in; a.ssq_out = ssq_out; a.xg_w = xg_w; a.xg_out = xg_out; a.npart = npart; a.eps = eps;
    a.sk_scratch = sk_scratch; a.sk_counters = sk_counters; a.ksplit = ksplit;
    a.sk_scratch_floats = (size_t)sk_scratch_floats; a.sk_counters_n = sk_counters_n;
    a.split_rows = G;
    launch_gemv(a, epi, S(stream));
}
/* split-mode GEMV: 0 = per-wave rings, 1 / -1 = workgroup-shared activation chunks (default) */
VCK_EXPORT void vck_set_gemv_variant(int v) { set_gemv_variant(v); }
VCK_EXPORT void vck_set_gemm_variant(int v) { set_gemm_variant(v); }
VCK_EXPORT void vck_set_gemv_wide(int v) { set_gemv_wide(v); }
VCK_EXPORT unsigned long long vck_gemv_wide_launches() { return gemv_wide_launches(); }
VCK_EXPORT unsigned long long vck_gemv_wg_launches() { return gemv_wg_launches(); }
VCK_EXPORT void vck_rmsnorm_split(const float* x, const int* row_idx, const float* w, uint16_t* y, int rows, int D, float eps,
                                  int ldy, uint64_t lo_off, void* stream) {
    launch_rmsnorm_split(x, row_idx, w, y, rows, D, eps, ldy, (size_t)lo_off, S(stream));
}
VCK_EXPORT void vck_layernorm_split(const float* x, const float* w, const float* b, uint16_t* y, int rows, int D, float eps,
                                    int ldy, uint64_t lo_off, void* stream) {
    launch_layernorm_split(x, w, b, y, rows, D, eps, ldy, (size_t)lo_off, S(stream));
}
VCK_EXPORT void vck_qkv_split32(const float* qkv, uint16_t* q_hi, uint16_t* q_lo, uint16_t* k_hi, uint16_t* k_lo, uint16_t* vt_hi,
                                uint16_t* vt_lo, float* k32, float* v32, int B, int T, int H, int hd, int q_stride, int ks_stride,
                                int vt_stride, int kv_stride, const float* rope_cos, const float* rope_sin, void* stream) {
    QkvSplit32Args a{qkv, q_hi, q_lo, k_hi, k_lo, vt_hi, vt_lo, k32, v32, B, T, H, hd, q_stride, ks_stride, vt_stride, kv_stride,
                     rope_cos, rope_sin};
    launch_qkv_split32(a, S(stream));
}
/* the same with fp24 cache rows (k24 / v24: [B,H,kv_stride] x 3*hd bytes: hd x u16 | hd x u8) */
VCK_EXPORT void vck_qkv_split24(const float* qkv, uint16_t* q_hi, uint16_t* q_lo, uint16_t* k_hi, uint16_t* k_lo, uint16_t* vt_hi,
                                uint16_t* vt_lo, void* k24, void* v24, int B, int T, int H, int hd, int q_stride, int ks_stride,
                                int vt_stride, int kv_stride, const float* rope_cos, const float* rope_sin, void* stream) {
    QkvSplit32Args a{qkv, q_hi, q_lo, k_hi, k_lo, vt_hi, vt_lo, reinterpret_cast<float*>(k24), reinterpret_cast<float*>(v24), B, T, H, hd,
                     q_stride, ks_stride, vt_stride, kv_stride, rope_cos, rope_sin, 1};
    launch_qkv_split32(a, S(stream));
}
VCK_EXPORT void vck_attention_split(const uint16_t* q_hi, const uint16_t* q_lo, const uint16_t* k_hi, const uint16_t* k_lo,
                                    const uint16_t* vt_hi, const uint16_t* vt_lo, uint16_t* out, int B, int H, int T, int hd,
                                    int q_stride, int kv_stride, int causal, float scale, int ldo, int lo_off, void* stream) {
    AttnArgs a{q_hi, k_hi, vt_hi, out, B, H, T, hd, q_stride, kv_stride, causal, scale, 0, q_lo, k_lo, vt_lo, ldo, lo_off};
    launch_attention(a, S(stream));
}
VCK_EXPORT void vck_attention_decode_kv32(const float* qkv, float* k, float* v, uint16_t* out, int B, int H, int hd, int kv_stride,
                                          const int* pos_rows, int pos_stride, const int* active_rows, const float* rope_cos,
                                          const float* rope_sin, float scale, int G, void* stream) {
    AttnDecodeFusedArgs a{reinterpret_cast<const uint16_t*>(qkv), reinterpret_cast<uint16_t*>(k), reinterpret_cast<uint16_t*>(v), out,
                          B, H, hd, kv_stride, pos_rows, rope_cos, rope_sin, scale, pos_stride, active_rows, 1, G};
    launch_attention_decode_fused(a, S(stream));
}
/* the same over fp24 caches */
VCK_EXPORT void vck_attention_decode_kv24(const float* qkv, void* k, void* v, uint16_t* out, int B, int H, int hd, int kv_stride,
                                          const int* pos_rows, int pos_stride, const int* active_rows, const float* rope_cos,
                                          const float* rope_sin, float scale, int G, void* stream) {
    AttnDecodeFusedArgs a{reinterpret_cast<const uint16_t*>(qkv), reinterpret_cast<uint16_t*>(k), reinterpret_cast<uint16_t*>(v), out,
                          B, H, hd, kv_stride, pos_rows, rope_cos, rope_sin, scale, pos_stride, active_rows, 2, G};
    launch_attention_decode_fused(a, S(stream));
}

