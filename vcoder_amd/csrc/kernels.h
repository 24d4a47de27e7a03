// kernels.h — launcher declarations for the gfx950 kernels of the VCoder hot path.
// Every launcher enqueues on `stream` and returns immediately; pointers are device pointers
// (host pointers under the test-only emulator build).  SURVEY.md §2 work-list ids (K1..K19) are
// quoted next to each launcher.
#pragma once
#include <stdint.h>
#ifdef VC_EMU
#include "hip_emu.h"
#else
#include <hip/hip_runtime.h>
#endif

// 16-bit operand format of this build (vc_device.h): bfloat16, or IEEE fp16 with -DVC_F16 (libvcoder_hip_f16.so)
#ifndef VC_OPERAND_FP16
#ifdef VC_F16
#define VC_OPERAND_FP16 1
#else
#define VC_OPERAND_FP16 0
#endif
#endif

namespace vc {

typedef uint16_t bf16_t;

enum GemmEpilogue : int {
    EPI_BF16 = 0,        // out bf16 [M,N]   = acc + bias
    EPI_BF16_QGELU = 1,  // out bf16 [M,N]   = quick_gelu(acc + bias)            (CLIP fc1, K7)
    EPI_BF16_GELU = 2,   // out bf16 [M,N]   = erf_gelu(acc + bias)              (adapter, K9)
    EPI_F32 = 3,         // out fp32 [M,N]   = acc + bias                        (patchify K1, logits K18)
    EPI_RESID_F32 = 4,   // out fp32 [M,N]  += acc + bias  (in-place residual)   (K6, K7, K16, K17)
    EPI_SWIGLU = 5,      // W rows interleaved (gate,up): out bf16 [M,N/2] = silu(g)*u   (K17)
    EPI_QKV = 6,         // fused-QKV projection of a prefill: RoPE + head split + KV-cache write in the epilogue (K13/K14, GemmArgs::qe)
};

// EPI_QKV (round 6; SURVEY K13 "fused into K12 epilogue / KV-write"; [HF] llama/modeling_llama.py:113-160,254-262): the QKV GEMM of
// a prefill layer writes, instead of the fused [M, 3D] bf16 rows qkv_split_kernel used to re-read, directly
//   q  [B,H,q_stride,hd]   roped queries                      k  [B,H,kv_stride,hd]   roped keys (the cache, or the flash scratch)
//   v  [B,H,kv_stride,hd]  value rows of the cache            vt [B,H,hd,vt_stride]   V^T scratch in the flash kernel's key order
//   k8 / v8 [B,H,kv8_stride,hd] e4m3 cache rows (fp8 weight format; v may then be nullptr)
// with the bits qkv_split_kernel produces (projection rounded to bf16, RoPE in fp32 on the rounded values, rounded again).
// hd = 128, (H * hd) % 256 == 0, N = 3 H hd.  The kernel reads the weight rows of a 256-row tile in the order
// {head 2a: d 0..63 | head 2a+1: d 0..63 | head 2a: d 64..127 | head 2a+1: d 64..127} (a row permutation of its LDS-DMA source
// addresses: the weights stay as they are), so a lane's accumulators of the tile's two halves are the rotate-half partners
// d and d + 64 of ONE head and the RoPE is lane-local.  The token rows of the GEMM are the samples padded to Tp = rup(T, 32)
// (M = B * Tp; row b * Tp + t reads A row b * T + min(t, T - 1), stores are masked to t < T): a 32-token block of the tile never
// straddles two samples, and a quad transpose inside the wave turns a lane's 4 features x 2 tokens into the 16-byte
// 8-key chunks of the V^T scratch.
struct QkvEpiArgs {
    bf16_t *q, *k, *v, *vt;
    uint8_t *k8, *v8;
    const float *rope_cos, *rope_sin;   // fp32 [max_pos, hd/2]; position = t
    int B, T, Tp, H;
    int q_stride, kv_stride, vt_stride, kv8_stride;
};

struct GemmArgs {
    const bf16_t* A;   // activations [M, lda], K contiguous
    const bf16_t* W;   // weights     [N, ldw], K contiguous (HF Linear layout: out_features x in_features)
    const float* bias; // [N] or nullptr
    void* out;         // see GemmEpilogue
    int M, N, K;       // K % 64 == 0, N % 8 == 0
    int lda, ldw, ldo; // leading dims in elements
    // optional fp32 workspace: lets the launcher split K for the tiles of a short last round (256x256 tiles on 256 CUs:
    // 608 tiles = 2.4 rounds) — `sk_ks` workgroups per remainder tile write partial tiles here and a fix-up launch adds
    // them in k order and applies the epilogue (deterministic).  sk_full / sk_ks are filled in by launch_gemm.
    float* ws;
    size_t ws_bytes;
    int sk_full, sk_ks;
    int tile_group, xcd_remap_on;  // tile order (launch_gemm fills them: 4 m-tiles per sweep group, XCD remap on)
    // f8 != 0: A and W point at OCP e4m3 bytes (lda / ldw in elements = bytes, K % 128 == 0) and the accumulator is
    // multiplied by a_scale[m] * w_scale[n] before bias / epilogue (EPI_BF16, EPI_RESID_F32, EPI_SWIGLU only)
    int f8;
    const float* a_scale;  // [M]
    const float* w_scale;  // [N]
    // precision mode "split": kwrap > 0 — the activation rows are K-concatenated [hi | lo] bf16 planes (K = 2 * kwrap * 64)
    // and the weight's k index wraps after kwrap k-tiles of 64 (W is [N, K / 2]); split_out > 0 — bf16 epilogues write
    // hi at column n and lo = bf16(v - hi) at column split_out + n (EPI_SWIGLU: at n / 2)
    int kwrap;
    int split_out;
    // ... with the lo plane of an inexact checkpoint's weight (w = bf16 hi + bf16 lo): w_lo_off != 0 is the BYTE offset from W to
    // a second [N, ldw] plane and K = 3 * kwrap * 64 — a third K segment contracts the activation HI plane (the A k index wraps
    // back to 0 there) against W_lo: a.w = a_hi.w_hi + a_lo.w_hi + a_hi.w_lo (the a_lo.w_lo term, 2^-16 of the product, is dropped)
    long long w_lo_off;
    // RMSNorm folded across two GEMMs of a prefill, as in the decode steps (rmsnorm(x) . W^T = rstd * ((x * g) . W^T)):
    //  consumer: row_scale[m] (= rstd of row m; launch_rstd_from_partials) multiplies the accumulator before the epilogue; A is
    //            then the producer's xg rows;
    //  producer (EPI_RESID_F32 only, N % 16 == 0): next to the updated fp32 residual rows it writes the NEXT GEMM's operand
    //            xg_out[m][n] = bf16(x * xg_w[n]) (row stride ld_xg; xg_lo > 0: the lo plane bf16(t - hi) xg_lo columns to the
    //            right — precision mode "split") and ssq_out[m * npart + n / 16] = sum of the 16 new residual values squared.
    const float* row_scale;
    bf16_t* xg_out;
    const float* xg_w;
    float* ssq_out;
    int ld_xg, xg_lo, npart;
    QkvEpiArgs qe;   // EPI_QKV only
};
void launch_gemm(const GemmArgs& a, int epilogue, hipStream_t s);
// rstd[m] = rsqrt(sum_{p < nparts} ssq[m * npart + p] / D + eps): the row scales of the consumer of a folded RMSNorm
void launch_rstd_from_partials(const float* ssq, int npart, int nparts, float* rstd, int rows, int D, float eps, hipStream_t s);

// Skinny GEMM for decode (M <= VC_GEMV_MAX_M token rows per weight pass): weights pre-packed in MFMA-fragment order, streamed once.  K12/K16/K17/K18.
// packed W layout: [N/16][K/32][64 lanes][8 bf16]; lane l of tile (nt,kt) holds W[nt*16+(l&15)][kt*32+(l>>4)*8 + e].
enum GemvEpilogue : int {
    GEMV_BF16 = 0,       // out bf16 [M, N]
    GEMV_F32 = 1,        // out fp32 [M, N]             (logits)
    GEMV_RESID_F32 = 2,  // out fp32 [M, N] += acc     (o_proj / down_proj into the residual stream)
    GEMV_SWIGLU = 3,     // interleaved (gate,up) rows: out bf16 [M, N/2]
};
constexpr int VC_GEMV_MAX_M = 32;  // token rows one weight pass of launch_gemv serves (two MFMA row groups above 16)
struct GemvArgs {
    const bf16_t* X;   // [M, K] bf16 activations (M <= VC_GEMV_MAX_M)
    const bf16_t* Wp;  // packed weights
    void* out;
    int M, N, K;       // N % 16 == 0, K % 32 == 0 (K % 64 == 0 for W8A16)
    int ldo;
    // W8A16: Wp holds e4m3 bytes in the 64-wide super-tile layout (launch_quantize_fp8) and wscale[n] the per-output-row
    // power-of-two scale; nullptr = bf16 weights.
    const float* wscale;
    // RMSNorm folded into producer + consumer (K11 of SURVEY.md §2 spread over K12/K16/K17/K18):
    //  consumer: ssq_in != nullptr -> out = rstd[m] * acc with rstd[m] = rsqrt(sum_p ssq_in[m][p] / K + eps); X is then
    //            the producer's xg = bf16(x * g)
    //  producer (RESID epilogue): ssq_out[m][n_tile] = sum over the tile's 16 columns of the updated residual^2, and
    //            xg_out[m][n] = bf16(residual * xg_w[n]) — the next consumer's activation operand
    const float* ssq_in;   // [M, npart] or nullptr
    float* ssq_out;        // [M, npart] or nullptr
    const float* xg_w;     // [N] norm weight of the consumer, or nullptr
    bf16_t* xg_out;        // [M, N]
    int npart;             // partials per row (multiple of 16)
    float eps;
    // deterministic split-K for matrices with few output tiles (o_proj, down): `ksplit` workgroups share one tile,
    // each writes its fp32 partial to sk_scratch[ks][tile][64 lanes][4]; the LAST to arrive (sk_counters[tile]) sums
    // the partials in k order — the result does not depend on arrival order — runs the epilogue and re-arms the counter
    float* sk_scratch;       // [ksplit][N/16][2 row groups][256] or nullptr
    unsigned* sk_counters;   // [N/16][2 row groups], zero between launches
    int ksplit;              // 0/1 = off (launcher decides when the two buffers are given)
    // precision mode "split" (0 = off; G = 8 for M <= 8, G = 16 for M <= 16): X is [G + M, K] — rows [0, M) the bf16 hi parts
    // of the activation rows, rows [G, G + M) the lo parts (x = hi + lo, ~16 mantissa bits) — and out[m] = (hi[m] + lo[m]) . W.
    // bf16-valued outputs (GEMV_BF16, GEMV_SWIGLU, xg_out) are written the same way: hi at row m, lo at row G + m.
    int split_rows;
    // capacity of the split-K buffers (0 = the historical [4][512][2][256] floats / [512][2] counters)
    size_t sk_scratch_floats;
    int sk_counters_n;
    // in-situ timing slot ([STAMP_WGS] x {start, end} wall-clock ticks, one entry per workgroup), or nullptr (vc_device.h stamp_begin)
    unsigned* stamp;
    // the packed lo plane of an inexact checkpoint's weight (w = bf16 hi + bf16 lo; same layout as Wp), or nullptr.  Precision mode
    // "split" on the workgroup-shared kernel only: out[m] = x_hi.w_hi + x_lo.w_hi + x_hi.w_lo
    const bf16_t* Wp_lo;
};
void launch_gemv(const GemvArgs& a, int epilogue, hipStream_t s);
// the decode GEMV of precision mode "split": 0 = per-wave rings (two weight passes of 16 rows), -1 / 1 = the workgroup-shared form
void set_gemv_variant(int v);
void set_gemm_variant(int v);   // overrides VC_GEMM_VARIANT inside one process (< 0: the environment's); 6 / 7 = the 32 x 32 x 16 MFMA form
void set_gemv_wide(int v);       // -1 / 1 = the measured classes (default), 0 = off, 2 = every class (launch_gemv_wide)
unsigned long gemv_wide_launches();
bool gemv_wg_enabled();                          // the workgroup-shared form serves the split step's GEMVs
bool gemv_wg_applies(int K, bool fp8_weights);   // ... for a matrix with this K / weight format
unsigned long gemv_wg_launches();   // launches served by the workgroup-shared form so far (tests)
void launch_pack_weight(const bf16_t* W, bf16_t* Wp, int N, int K, hipStream_t s);
// W [N,K] bf16 -> e4m3 packed + per-row scales; W is overwritten with the dequantised values (see decode.hip)
void launch_quantize_fp8(bf16_t* W, uint8_t* Wq, float* scale, int N, int K, hipStream_t s, uint8_t* Wrow = nullptr);
// token rows of A [M, lda] bf16 -> e4m3 bytes Q [M, K] + per-row power-of-two scales (K % 16 == 0)
void launch_quant_act_rows(const bf16_t* A, int lda, uint8_t* Q, float* scale, int M, int K, hipStream_t s);
// interleave gate/up rows: out[2f] = gate[f], out[2f+1] = up[f]
void launch_interleave_rows(const bf16_t* gate, const bf16_t* up, bf16_t* out, int F, int K, hipStream_t s);

// ---- norms (wave-per-row, fp32 statistics) --------------------------------------------------
// K3: LayerNorm(D) fp32 in -> bf16 out.  K11: RMSNorm.
void launch_layernorm(const float* x, const float* w, const float* b, bf16_t* y, int rows, int D, float eps,
                      hipStream_t s);
// ldy: row stride of y in elements (0 = D)
void launch_rmsnorm(const float* x, const float* w, bf16_t* y, int rows, int D, float eps, hipStream_t s, int ldy = 0);
// rows gathered through an index (final norm over the last token of each sample): row r reads x[idx[r]]
// RMSNorm + per-row e4m3 quantisation in one pass: q [rows, D] bytes + scale [rows] (= launch_rmsnorm then launch_quant_act_rows)
void launch_rmsnorm_q8(const float* x, const float* w, uint8_t* q, float* scale, int rows, int D, float eps, hipStream_t s);
void launch_rmsnorm_rows(const float* x, const int* row_idx, const float* w, bf16_t* y, int rows, int D, float eps,
                         hipStream_t s);
// precision mode "split": the normalised row as two bf16 planes, hi = bf16(o) at y + r * ldy and lo = bf16(o - hi) at
// y + r * ldy + lo_off (lo_off = D with ldy >= 2 D: the K-concatenated operand of a kwrap GEMM; lo_off = G * D with ldy = D:
// the stacked hi / lo row groups of a split decode GEMV)
void launch_layernorm_split(const float* x, const float* w, const float* b, bf16_t* y, int rows, int D, float eps, int ldy,
                            size_t lo_off, hipStream_t s);
void launch_rmsnorm_split(const float* x, const int* row_idx, const float* w, bf16_t* y, int rows, int D, float eps, int ldy,
                          size_t lo_off, hipStream_t s);

// ---- ViT front end ---------------------------------------------------------------------------
// K1 (im2col half): pixels fp32 [N,3,S,S] -> cols bf16 [N*g*g, Kpad] (zero padded from 3*P*P to Kpad)
// split: cols is [N*g*g, 2 * Kpad], columns [0, Kpad) = bf16(pixel), [Kpad, 2 Kpad) = bf16(pixel - hi)
void launch_im2col(const float* pixels, bf16_t* cols, int n_img, int image, int patch, int Kpad, hipStream_t s, bool split = false);
// K2: x[n][0] = cls + pos[0]; x[n][1+p] = patches[n*g2+p] + pos[1+p]; then pre-LayerNorm, fp32 out [N,T,D]
void launch_vit_embed_ln(const float* patches, const float* cls, const float* pos, const float* w, const float* b,
                         float* x, int n_img, int T, int D, float eps, hipStream_t s);

// K8 feature_select: x fp32 [N,T,D] -> y bf16 [N*(T-skip), D] dropping the first `skip` rows (CLS) of each image
// split: y is [N*(T-skip), 2 * D] = [hi | lo]
void launch_select_rows_bf16(const float* x, bf16_t* y, int n_img, int T, int skip, int D, hipStream_t s, bool split = false);

// ---- attention -------------------------------------------------------------------------------
// qkv split (+RoPE for the LLM, K13/K14): qkv bf16 [B*T, 3*D] -> Q [B,H,Tq_stride,hd], K [B,H,S_stride,hd] at pos0..,
// V^T [B,H,hd,S_stride].
struct QkvSplitArgs {
    const bf16_t* qkv;
    bf16_t* q;
    bf16_t* k;
    bf16_t* vt;
    int B, T, H, hd;
    int q_stride;   // rows per (b,h) in Q
    int kv_stride;  // rows (keys) per (b,h) in K (and V), columns in V^T unless vt_stride is given
    const int* pos0_dev;  // unused (the decode append is fused into the decode attention kernel); keep nullptr
    const float* rope_cos;  // fp32 [max_pos, hd/2] tables (LlamaRotaryEmbedding); nullptr disables RoPE (ViT)
    const float* rope_sin;
    bf16_t* v;      // V key-major [B,H,kv_stride,hd] (the KV cache the decode steps stream), or nullptr (ViT)
    int vt_stride;  // columns per row of V^T (0: kv_stride) — the LLM prefill keeps V^T in a per-call scratch
    // e4m3 KV cache of the fp8 weight format: k8 / v8 [B,H,kv8_stride,hd] bytes receive e4m3(bf16 K after RoPE) / e4m3(V) — what
    // the decode steps stream (half the bf16 bytes); `k` is then a per-call bf16 scratch for THIS prefill's flash kernel
    uint8_t* k8;
    uint8_t* v8;
    int kv8_stride;
};
void launch_qkv_split(const QkvSplitArgs& a, hipStream_t s);

// K5/K15 flash attention (prefill / ViT): out bf16 [B*T, H*hd]
struct AttnArgs {
    const bf16_t* q;
    const bf16_t* k;
    const bf16_t* vt;
    bf16_t* out;
    int B, H, T, hd;  // T queries == T keys (self attention from position 0)
    int q_stride, kv_stride;
    int causal;
    float scale;
    int vt_stride;    // columns per row of V^T (0: kv_stride)
    // precision mode "split" (q_lo != nullptr): q / k / vt are the hi planes and these the lo planes of the same layouts
    // (x = hi + lo); the output row is [hi | lo]: row stride ldo elements, lo plane at column lo_off
    const bf16_t* q_lo;
    const bf16_t* k_lo;
    const bf16_t* vt_lo;
    int ldo, lo_off;
    // padded batches: key_mask[b * mask_stride + key] == 0 hides that key from every query of sequence b (the caller's 2-D
    // attention_mask, left-extended over the spliced rows as the reference does); nullptr = no mask
    const uint8_t* key_mask;
    int mask_stride;
    // set by the launcher: workgroup order (0 = 3-D grid as is; see attention_kernel) and query blocks per (b, h)
    int sched, nqb;
};
void launch_attention(const AttnArgs& a, hipStream_t s);

// precision mode "split": fp32 fused-QKV rows -> RoPE (fp32) -> fp32 K / V cache rows + bf16 hi / lo planes of Q, K, V^T
struct QkvSplit32Args {
    const float* qkv;        // [B*T, 3*H*hd] fp32
    bf16_t *q_hi, *q_lo;     // [B,H,q_stride,hd]
    bf16_t *k_hi, *k_lo;     // [B,H,ks_stride,hd]   per-call scratch of the flash kernel
    bf16_t *vt_hi, *vt_lo;   // [B,H,hd,vt_stride]
    float *k32, *v32;        // fp32 caches [B,H,kv_stride,hd], rows 0..T-1 written; nullptr: none (ViT)
    int B, T, H, hd, q_stride, ks_stride, vt_stride, kv_stride;
    const float* rope_cos;   // nullptr disables RoPE (ViT)
    const float* rope_sin;
    int kv24;                // != 0: k32 / v32 point at fp24 caches (3 * hd bytes per row: hd x u16 hi plane | hd x u8 lo plane)
};
void launch_qkv_split32(const QkvSplit32Args& a, hipStream_t s);

// fused decode attention (K13+K14+K15 for q_len = 1): RoPE of the new q/k, KV-cache append and attention over the cache
// in ONE launch per layer; K and V are both key-major; positions come from device memory (hipGraph-replayable)
struct AttnDecodeFusedArgs {
    const bf16_t* qkv;   // [B, 3*H*hd] fused projection output of the new token
    bf16_t* k;           // [B,H,kv_stride,hd]   (row `pos` is written)
    bf16_t* v;           // [B,H,kv_stride,hd]   (row `pos` is written)
    bf16_t* out;         // [B, H*hd]
    int B, H, hd, kv_stride;   // kv_stride <= 4096 (the scores of a row sit in LDS)
    const int* pos_dev;  // position of the new token == number of keys already cached; row b reads pos_dev[b * pos_stride]
    const float* rope_cos;
    const float* rope_sin;
    float scale;
    int pos_stride;         // 0: one position for every row
    const int* active_dev;  // nullptr, or row b is skipped when active_dev[b * pos_stride] == 0
    // precision mode "split" (kv32 != 0): qkv is fp32 [B, 3*H*hd], k / v are fp32 caches (kv32 == 1) or fp24 caches (kv32 == 2:
    // rows of hd x u16 | hd x u8, vc_device.h), and the output is written as bf16 hi / lo rows in stacked groups of out_G rows:
    // row b -> hi at row (b / G) * 2G + b % G, lo G rows further.
    // kv32 == 3: the bf16 step over e4m3 caches (rows of hd bytes; the fp8 weight format): qkv / q / out as for kv32 == 0
    int kv32;
    int out_G;
    // keys hidden by the row's attention_mask: key_mask[b * mask_stride + key] == 0 (nullptr = none)
    const uint8_t* key_mask;
    int mask_stride;
    unsigned* stamp;   // in-situ timing slot, or nullptr (GemvArgs::stamp)
};
void launch_attention_decode_fused(const AttnDecodeFusedArgs& a, hipStream_t s);

// ---- splice / embedding (K10) ------------------------------------------------------------------
// per destination row r of inputs_embeds [B*S, D]: row_src[2r] = kind (0 = token id -> embed_tokens row,
// 1 = row of the projected feature buffer, 2 = zero padding row), row_src[2r+1] = token id / feature row.
// The table is the host-side splice plan of prepare_inputs_labels_for_multimodal (vcoder_ds_llava_arch.py:175-305).
void launch_splice(const int* row_src, int nrows, const bf16_t* embed, const bf16_t* feats, float* x, int D,
                   hipStream_t s);
// embed_lo (here and below): the lo plane of an inexact checkpoint's table (x = hi + lo), or nullptr
void launch_embed_tokens(const int* tok, const bf16_t* embed, float* x, int B, int D, hipStream_t s, const bf16_t* embed_lo = nullptr);

// ---- greedy select (K19) ----------------------------------------------------------------------
// logits fp32 [B,V] -> next token (lowest index on ties); EOS/pad bookkeeping; appends to out_ids[b*max_new+step]
struct GreedyArgs {
    const float* logits;
    int* next_tok;      // [B]
    int* out_ids;       // [B, max_new]
    int* finished;      // [B]
    int* step_dev;      // device scalar (read here; advanced by launch_advance)
    int B, V, max_new, eos_id, pad_id;
    // device-side keyword stop (greedy_embed only): a row is finished as soon as the ids it has produced (continued
    // backwards into the tail of its prompt) end with one of the stop sequences.
    //   stop_tab: [0] = number of sequences (<= VC_MAX_STOP), then per sequence 1 + VC_MAX_STOP_LEN ints (length, ids)
    //   prompt_tail: [B][VC_MAX_STOP_LEN - 1] last prompt ids of every row (right-aligned)
    const int* stop_tab;     // nullptr = none
    const int* prompt_tail;
};
constexpr int VC_MAX_STOP = 8, VC_MAX_STOP_LEN = 8;
void launch_greedy(const GreedyArgs& a, hipStream_t s);
// ---- per-row decode state (select.hip) ------------------------------------------------------------------------------
// One record of RS_STRIDE ints per row of the decode loop, in device memory.  Every kernel of a decode step that needs a
// position, a step count or a parameter of a row reads it here, so one captured graph serves rows of different requests.
enum RowStateField : int {
    RS_ACTIVE = 0,    // 0: the row is skipped by attention and selection (its GEMV lanes compute garbage nobody reads)
    RS_FINISHED = 1,  // EOS seen / stop sequence matched: later tokens are pad
    RS_STEP = 2,      // tokens produced so far = index of the next entry of the row's out_ids
    RS_POS = 3,       // position of the token the next step processes = keys already in the row's KV cache
    RS_MAXNEW = 4,    // out_ids entries of the row (0: nothing is recorded — vc_decode_step)
    RS_EOS = 5,       // < 0: disabled
    RS_PAD = 6,
    RS_NSTOP = 7,     // stop sequences in RS_STOP (<= VC_MAX_STOP)
    RS_SAMPLE = 8,    // 0 greedy, 1 temperature / top-k / top-p sampling
    RS_INVTEMP = 9,   // float bits: 1 / temperature
    RS_TOPK = 10,     // <= 0: off
    RS_TOPP = 11,     // float bits; >= 1: off
    RS_SEED_LO = 12,
    RS_SEED_HI = 13,
    RS_OUT_OFF = 14,  // offset (ints) of the row's out_ids inside SelectArgs::out_ids
    RS_TAIL = 16,     // VC_MAX_STOP_LEN - 1 last prompt ids, right-aligned (suffix matches that reach into the prompt)
    RS_STOP = 24,     // VC_MAX_STOP x (length, VC_MAX_STOP_LEN ids)
    RS_STRIDE = 128,
};
static_assert(RS_STOP + VC_MAX_STOP * (1 + VC_MAX_STOP_LEN) <= RS_STRIDE, "row record overflow");

// selection (greedy / sampled) + EOS / stop bookkeeping + embedding of the selected token (fp32 residual row x, RMSNorm
// partials ssq, first GEMV operand xg = bf16(x * xg_w)) + per-row step / position advance: one workgroup per row
struct SelectArgs {
    const float* logits;  // [nrows, ldl] fp32
    int ldl;
    int* rows;            // [nrows][RS_STRIDE]
    int* next_tok;        // [nrows]
    int* out_ids;         // base of the id store; row r writes out_ids[rows[r][RS_OUT_OFF] + step]
    const bf16_t* embed;  // [V, D]; nullptr: no embedding (selection only)
    float* x;             // [nrows, D]
    float* ssq;           // [nrows, npart]
    const float* xg_w;    // [D]
    bf16_t* xg;           // [nrows, D]
    int D, npart, V;
    int nrows;
    int advance;          // bit 0: step += 1; bit 1: pos += 1 (after the selection)
    int lds_floats;       // filled by the launcher: floats of LDS staging available to the sampler
    int row0;             // workgroup i handles state row row0 + i (rows / next_tok / x / ssq / xg) with logits row i: rows
                          // that join a running loop are selected from their prefill's own logits buffer
    int xg_G;             // precision mode "split" (0 = off): xg holds stacked groups of G hi rows + G lo rows
    const bf16_t* embed_lo;  // lo plane of an inexact checkpoint's embedding table (strict / split: x = hi + lo), or nullptr
};
void launch_select_embed(const SelectArgs& a, hipStream_t s);
// in-situ timing of the decode-step launches (GemvArgs::stamp): slot j = [STAMP_WGS] x {start, end} u32 ticks (zero = not stamped);
// `n` slots laid out 5 per layer (qkv, attention, o, gate/up, down) + lm_head; acc[kind] = {exec ticks (latest end - earliest
// start), period ticks (latest end - the previous launch's latest end), launches}, kinds as in vc_pool_profile_read.  The slots
// and `scratch` ([1 + 2 n] words) must be zero before the first step (the kernel re-zeroes / re-arms them).
constexpr int PROF_KINDS = 6;
constexpr size_t STAMP_SLOT_WORDS = 2 * 2048;   // == 2 * STAMP_WGS (vc_device.h)
void launch_stamp_accumulate(unsigned* stamps, int n, int layers, unsigned long long* acc, unsigned* scratch, hipStream_t s);
// embedding + sum-of-squares partials for tokens supplied by the host (vc_decode_step with explicit tokens)
void launch_embed_tokens_ssq(const int* tok, const bf16_t* embed, float* x, float* ssq, const float* xg_w, bf16_t* xg, int B, int D, int npart,
                             hipStream_t s, int xg_G = 0, const bf16_t* embed_lo = nullptr);
void launch_advance(int* step_dev, int* pos_dev, int* ctx_dev, hipStream_t s);
// test hook: u[i] = the sampler's uniform for hash value h[i], gumbel[i] = -log(-log(u[i]))
void launch_uniform_probe(const uint32_t* h, float* u, float* gumbel, int n, hipStream_t s);

// ---- strict (fp32-faithful) path: see strict.hip -------------------------------------------------------------------
struct GemmF32Args {
    const float* A;     // fp32 activations [M, lda]
    const bf16_t* W;    // bf16 weights [N, ldw] (widened exactly)
    const float* bias;  // [N] or nullptr
    float* out;         // fp32; epilogue ids reuse GemmEpilogue: EPI_F32 (also for EPI_BF16), *_QGELU, *_GELU, RESID, SWIGLU
    int M, N, K;
    int lda, ldw, ldo;
    const bf16_t* W_lo; // [N, ldw] lo plane of an inexact checkpoint (w = bf16 hi + bf16 lo), or nullptr
};
void launch_gemm_f32(const GemmF32Args& a, int epilogue, hipStream_t s);
struct AttnF32Args {
    const float* q;   // [B,H,q_stride,hd]
    const float* k;   // [B,H,kv_stride,hd]
    const float* v;   // [B,H,kv_stride,hd]
    float* out;       // [B*Tq, H*hd]
    int B, H, Tq, hd, q_stride, kv_stride;
    int causal;       // 1: query t attends keys 0..pos0+t ; 0: keys 0..Tk-1
    int Tk;
    const int* pos0_dev;  // device scalar: absolute position of query 0 (nullptr -> 0)
    float scale;
    const uint8_t* key_mask;  // [B, mask_stride]: 0 hides the key from every query of the sequence (nullptr = none)
    int mask_stride;
};
void launch_attention_f32(const AttnF32Args& a, hipStream_t s);
// output_attentions (diagnostic): probabilities [B, H, T, T] fp32 of a causal prefill layer from its q / k — fp32 (q32 / k32),
// bf16 (q_hi / k_hi) or bf16 hi + lo planes
struct AttnProbsArgs {
    const float* q32;      // [B,H,q_stride,hd] or nullptr
    const bf16_t* q_hi;    // used when q32 == nullptr
    const bf16_t* q_lo;    // nullptr: none
    const float* k32;      // [B,H,kv_stride,hd] or nullptr
    const bf16_t* k_hi;
    const bf16_t* k_lo;
    const void* k24;       // fp24 cache rows [B,H,kv_stride] x (hd x u16 | hd x u8), used when k32 == nullptr and k_hi == nullptr
    float* out;            // [B,H,T,Tk]
    int B, H, T, hd, q_stride, kv_stride;   // T queries; keys <= 4096
    float scale;
    const uint8_t* key_mask;
    int mask_stride;
    int Tk;                // key columns of an output row (0: T — the prefill form)
    int q_pos0;            // position of query 0: query t sees keys 0 .. q_pos0 + t (a cached decode step: T = 1, q_pos0 = its position)
    const uint8_t* k8;     // e4m3 cache rows [B,H,kv_stride,hd] (the fp8 weight format's KV), or nullptr
};
void launch_attn_probs(const AttnProbsArgs& a, hipStream_t s);
// q of ONE new token per row, rotated to position `pos`, for launch_attn_probs on a cached decode step: qkv rows [B, 3 H hd] (bf16, or
// fp32 when qkv_f32) -> q [B,H,1,hd] fp32; round_bf16: the values the fused decode attention uses (q rounded to bf16 after RoPE)
void launch_rope_q_decode(const void* qkv, bool qkv_f32, float* q, int B, int H, int hd, int pos, const float* rope_cos,
                          const float* rope_sin, bool round_bf16, hipStream_t s);
struct QkvF32Args {
    const float* qkv;  // [B*T, 3*H*hd]
    float* q;          // [B,H,q_stride,hd]
    float* k;          // [B,H,kv_stride,hd], rows pos0..pos0+T-1 written
    float* v;
    int B, T, H, hd, q_stride, kv_stride;
    const int* pos0_dev;
    const float* rope_cos;  // nullptr: no RoPE (ViT)
    const float* rope_sin;
};
void launch_qkv_rope_f32(const QkvF32Args& a, hipStream_t s);
void launch_layernorm_f32(const float* x, const float* w, const float* b, float* y, int rows, int D, float eps, hipStream_t s);
void launch_rmsnorm_f32(const float* x, const int* row_idx, const float* w, float* y, int rows, int D, float eps,
                        hipStream_t s);
void launch_im2col_f32(const float* pixels, float* cols, int n_img, int image, int patch, hipStream_t s);
void launch_select_rows_f32(const float* x, float* y, int n_img, int T, int skip, int D, hipStream_t s);
void launch_splice_f32(const int* row_src, int nrows, const bf16_t* embed, const float* feats, float* x, int D, hipStream_t s,
                       const bf16_t* embed_lo = nullptr);

// ---- device-side image preprocessing (preprocess.hip) --------------------------------------------------------------
void launch_pad_square(const uint8_t* src, int h, int w, uint8_t* dst, int side, int ox, int oy, const int fill[3],
                       hipStream_t s);
void launch_resample(const uint8_t* in, int in_h, int in_w, uint8_t* out, int out_h, int out_w, const int* bounds,
                     const int* kk, int ksize, int horizontal, hipStream_t s);
void launch_crop_normalize(const uint8_t* in, int in_h, int in_w, int top, int left, float* out, int S, const float mean[3],
                           const float stdv[3], hipStream_t s);

// ---- misc ---------------------------------------------------------------------------------------
void launch_synth_bf16(bf16_t* out, size_t n, uint32_t tseed, float offset, float halfwidth, hipStream_t s);
void launch_synth_f32(float* out, size_t n, uint32_t tseed, float offset, float halfwidth, hipStream_t s);
// the unrounded generator value, as fp16-representable (rounding 1) or fp32 (2): synth.py synth_tensor(rounding="fp16" | "fp32")
void launch_synth_f32_rounded(float* out, size_t n, uint32_t tseed, float offset, float halfwidth, int rounding, hipStream_t s);
constexpr int ROW_SUM_PARTS = 32;  // out: [rows][ROW_SUM_PARTS] partial sums
void launch_row_sum(const float* x, size_t n_per_row, int rows, float* out, hipStream_t s);
// rows of one layer's K or V cache permuted in place through `tmp`: row r <- old row perm[r], live prefix only (beam search)
void launch_kv_permute(void* cache, void* tmp, const int* perm, int rows, int H, size_t cap_row_bytes, size_t live_row_bytes,
                       hipStream_t s);
void launch_f32_to_bf16(const float* in, bf16_t* out, size_t n, hipStream_t s);
// hi = bf16(in), lo = bf16(in - hi) (nullptr: skipped), *inexact = 1 if any in != hi (checkpoints bf16 cannot hold: the weight lo
// planes of the strict / split precision modes)
void launch_f32_to_bf16_planes(const float* in, bf16_t* hi, bf16_t* lo, size_t n, unsigned* inexact, hipStream_t s);
void launch_bf16_to_f32(const bf16_t* in, float* out, size_t n, hipStream_t s);   // `in` in the build's 16-bit operand format
void launch_truebf16_to_f32(const uint16_t* in, float* out, size_t n, hipStream_t s);   // `in` = bfloat16 bits (checkpoint data)

}  // namespace vc
