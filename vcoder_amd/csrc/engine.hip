// engine.hip — host-side engine + C ABI (include/vcoder_hip.h) of the MI355X-native VCoder hot path.
//
// Owns: weights (HF key -> fused / interleaved / MFMA-packed device layouts), the ViT and LLM workspaces,
// the KV cache (K key-major, V transposed), rope tables, the splice planner, and the hipGraph of one
// decode step.  Replaces L0-L2 (+ forward/generate of L3) of the reference:
//   CLIPVisionTower.forward            vcoder_llava/model/multimodal_encoder/clip_encoder.py:39-51
//   encode_*                            vcoder_llava/model/vcoder_ds_llava_arch.py:106-124
//   prepare_inputs_labels_for_multimodal  vcoder_ds_llava_arch.py:126-314, vcoder_llava_arch.py:146-296, llava_arch.py:99-199
//   ...ForCausalLM.forward              vcoder_llava/model/language_model/vcoder_ds_llava_llama.py:57-118
//   HF greedy loop                      SURVEY.md Appendix C
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <deque>
#include <map>
#include <set>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/vcoder_hip.h"
#include "kernels.h"

using namespace vc;

#define VC_API extern "C" __attribute__((visibility("default")))

static const int IMAGE_TOKEN_INDEX = -200;  // vcoder_llava/constants.py:5
static const int SEG_TOKEN_INDEX = -300;    // constants.py:8
static const int DEPTH_TOKEN_INDEX = -400;  // constants.py:11
static const int VC_MAX_ROWS = 16;          // sequences one prefill / one session loop handles
// The prefill's RMSNorm output rows are padded by 64 elements: with both GEMM operands at a row stride of exactly 2^13
// bytes (K = 4096 bf16) the 9728 x 12288 QKV GEMM of the 7b model ran 19 % slower (915 vs 770 us; address aliasing
// between the concurrently fetched panels — tools/experiments/gemm_rounds.py); no other shape cares.
static const int XN_PAD = 64;
static const int VC_POOL_ROWS = 32;         // rows of the shared decode pool (two MFMA token-slot groups = one weight pass per step)
static const int VC_POOL_ROWS_MAX = 64;     // vc_pool_set_rows(64): a pool whose step takes two 32-row weight passes (measurement)

#include "engine_ctx.h"

namespace {

struct Fail {
    int code;
    std::string msg;
};
#define HIPCHK(x)                                                                                       \
    do {                                                                                                \
        hipError_t e_ = (x);                                                                            \
        if (e_ != hipSuccess)                                                                           \
            throw Fail{VC_ERR_HIP, std::string(#x) + ": " + hipGetErrorString(e_)};                     \
    } while (0)
#define REQUIRE(cond, code, ...)                                                                        \
    do {                                                                                                \
        if (!(cond)) {                                                                                  \
            char b_[512];                                                                               \
            snprintf(b_, sizeof b_, __VA_ARGS__);                                                       \
            throw Fail{code, b_};                                                                       \
        }                                                                                               \
    } while (0)

inline size_t rup(size_t x, size_t a) { return (x + a - 1) / a * a; }

// stream of the session whose C-ABI call is running on this host thread (set by USE_DEVICE); zero-fills of freshly
// allocated buffers are enqueued on it — never on the legacy NULL stream, which would implicitly synchronise with (and
// invalidate the graph capture of) other sessions' streams
thread_local hipStream_t t_stream = nullptr;

// Every session and the decode pool get their own non-blocking stream.  (Rounds 3-4 measured CU masks and queue priorities for
// them — disjoint CU ranges for the MFMA-bound and the HBM-bound phase, high / low priority queues: all negative,
// profiles/r03_a_cu_mask_experiment.md, r03_q_stream_priority_and_inflight.md — and the knobs were removed in round 5.)
hipStream_t make_stream() {
    hipStream_t st = nullptr;
    // non-blocking: no implicit synchronisation with the legacy NULL stream (torch's default stream, other sessions)
    HIPCHK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    return st;
}

struct Buf {  // grow-only device buffer
    void* p = nullptr;
    size_t cap = 0;
    void ensure(size_t bytes, bool zero = false) {
        if (bytes <= cap) return;
        if (p) HIPCHK(hipFree(p));
        p = nullptr;
        cap = 0;
        HIPCHK(hipMalloc(&p, rup(bytes, 256)));
        cap = rup(bytes, 256);
        if (zero) {
            HIPCHK(hipMemsetAsync(p, 0, cap, t_stream));
            HIPCHK(hipStreamSynchronize(t_stream));
        }
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
    template <class T> T* as() const { return reinterpret_cast<T*>(p); }
};

struct VitLayer {
    float *ln1_w, *ln1_b, *ln2_w, *ln2_b, *qkv_b, *out_b, *fc1_b, *fc2_b;
    bf16_t *qkv_w, *out_w, *fc1_w, *fc2_w;
};
struct LlmLayer {
    float *in_norm, *post_norm;
    bf16_t *qkv_w, *o_w, *gate_tmp, *up_tmp, *gu_w, *down_w;  // row-major (prefill GEMM)
    bf16_t *qkv_p, *o_p, *gu_p, *down_p;                      // MFMA-fragment packed (decode GEMV); e4m3 bytes if W8A16
    float *qkv_s = nullptr, *o_s = nullptr, *gu_s = nullptr, *down_s = nullptr;  // W8A16 per-output-row scales
    uint8_t *qkv_q = nullptr, *o_q = nullptr, *gu_q = nullptr, *down_q = nullptr;  // weight format 2: e4m3 bytes, row-major
};
struct Projector {
    int depth = 0;
    std::vector<bf16_t*> w;
    std::vector<float*> b;
};

}  // namespace

struct vc_model {
    vc_ctx* ctx;
    vc_model_cfg c;
    hipStream_t st;
    bool finalized = false;
    bool owns_weights = true;
    int precision = 0;  // 0: bf16 MFMA fast path; 1: strict fp32 path (strict.hip); 2: split — fp32 activations in HBM,
                        // every MFMA operand as bf16 hi + lo fragments on the fast kernels
    int kv_es = 2;      // bytes per element of this session's own KV cache (kc / vc): 2 bf16, 4 fp32 (precision 2)
    int weight_format = 0;  // 0: bf16; 1: W8A16 — decoder linears stored as e4m3 + per-row scales for the decode GEMV;
                            // 2: fp8 — 1 + the prefill GEMMs run e4m3 x e4m3 on the K=128 scaled MFMA (W8A8)
    Buf s_cols, s_patches, s_vx, s_vxn, s_vqkv, s_vq, s_vk, s_vv, s_vattn, s_vh, s_sel, s_mid, s_feats;
    Buf s_xn, s_qkv, s_q, s_attn, s_h, s_kc, s_vc, s_xl;
    Buf pp_src, pp_sq, pp_tmp, pp_out, pp_tab, pp_f32;
    int s_capB = 0, s_capS = 0;  // false: a session created by vc_model_create_shared (weights belong to the parent)
    // derived
    int P, Tv, Kpatch, Kpad, hd, vhd, npart;
    std::vector<void*> owned;  // every weight allocation
    // Checkpoints bf16 cannot hold (the reference's: an fp16 LLM, builder.py:25-40, and an fp32 CLIP hub checkpoint cast to fp16,
    // clip_encoder.py:22-27): every matrix whose fp32 source had a value != bf16(value) keeps a second bf16 plane lo = bf16(w - hi)
    // of the same layout.  lo_of maps a hi plane (row-major, or the decode steps' packed copy) to it.  Precision modes "strict" and
    // "split" contract against hi + lo (w to ~16 mantissa bits; exact for fp16 values); the bf16 fast path uses hi alone.
    // Written while loading / finalizing only; sessions read their root's map.
    std::map<const void*, bf16_t*> lo_of;
    std::set<std::pair<const void*, size_t>> inexact_regions;   // (matrix, offset) of every loaded tensor that needed a lo plane
    Buf cvt_tmp;               // fp16-operand build: fp32 image of a bfloat16 checkpoint tensor on its way to the planes kernel
    unsigned* inexact_flag = nullptr;   // device word for the loader's check
    std::map<std::string, bool> need;
    // weights
    float *vit_cls = nullptr, *vit_pos = nullptr, *vit_pre_w = nullptr, *vit_pre_b = nullptr;
    bf16_t* vit_patch_w = nullptr;
    std::vector<VitLayer> vit;
    std::vector<LlmLayer> llm;
    Projector mm, seg;
    bf16_t *embed = nullptr, *lm_head = nullptr, *lm_head_p = nullptr;
    float* final_norm = nullptr;
    float *rope_cos = nullptr, *rope_sin = nullptr;
    // staging
    Buf stage, stage2;
    // ViT workspace
    Buf v_pixels, v_cols, v_patches, v_x, v_xn, v_qkv, v_q, v_k, v_vt, v_attn, v_h, v_sel, v_mid, feats;
    int feat_rows[3] = {0, 0, 0}, feat_off[3] = {0, 0, 0};
    // list / 5-D image form (vcoder_ds_llava_arch.py:135-169): images per sample and modality for the NEXT prefill
    // (vc_set_image_counts; empty = one image per sample), and the running first-image index of every sample
    std::vector<int> img_counts[3], img_first[3];
    // padded batches: the caller's 2-D attention_mask for the NEXT prefill / generate (vc_set_attention_mask; one-shot), and
    // the left-extended key mask of the CURRENT prefill on the device ([VC_MAX_ROWS][max_positions] bytes, 1 = visible)
    std::vector<uint8_t> mask_next;
    int mask_B = 0, mask_T = 0;
    Buf kmask;
    bool has_kmask = false;         // the current prefill hides keys
    bool kmask_in_decode = false;   // ... and the session's decode steps keep hiding them (vc_decode_step loops)
    bool graph_masked = false;      // what the captured decode graph was built for
    // output_hidden_states of the NEXT vc_prefill (vc_request_hidden_states; one-shot): host buffer [(L + 1), B, S, hidden]
    float* hidden_out = nullptr;
    size_t hidden_cap = 0;        // floats
    Buf hidden_tmp;
    // output_attentions of the NEXT vc_prefill (vc_request_attentions; one-shot): host buffer [L, B, H, S, S]
    float* attn_out = nullptr;
    size_t attn_cap = 0;
    Buf attn_q;                   // roped q of a decode step with output_attentions
    bool plan_only = false;       // do_prefill stops behind the splice plan (vc_plan_spliced_len): no tower pass, no state change
    int reserve_new = 64;         // KV slots a vc_prefill keeps free behind the prompt (vc_model_reserve_decode)
    int layer_limit = 0;          // > 0: a prefill evaluates only the first layer_limit decoder layers (vc_model_set_layer_limit)
    // LLM workspace
    Buf x, xn, qkv, q, attn, h, kc, vc, vt_pre, row_src, last_idx, xl, logits_all;
    Buf k_pre;                    // bf16 K rows of the current prefill layer when the cache holds e4m3 rows (fp8 format)
    Buf p_ssq, p_rstd;            // prefill: sum-of-squares partials [B*S, npart] and 1/rms [B*S] of the folded RMSNorm
    Buf a8, a8_scale;             // weight format 2: e4m3 activation rows of the current prefill GEMM + their scales
    int capB = 0, capS = 0;  // KV capacity
    int curB = 0, curS = 0, cur_pos = -1;
    // decode state of this session's own loop (vc_prefill / vc_decode_step, strict mode, generate with the pool off)
    Buf x_dec, xg_dec, qkv_dec, attn_dec, h_dec, logits, next_tok, out_ids, rows, dsum, ssq;
    Buf sk_scratch, sk_counters;  // split-K partials / arrival counters of the decode GEMV (few-tile matrices)
    Buf gemm_ws;                  // fp32 workspace of the GEMM's split-K remainder round (64 MiB)
    int out_stride = 0;           // out_ids ints per row
    int last_S = 0;               // spliced prompt length of the last prefill / generate
    hipGraphExec_t graph = nullptr;  // one decode step over graph_rows rows (parameters live in the RowState records)
    int graph_rows = 0;
    struct vc_pool* pool = nullptr;  // the root model's shared decode pool (created on first use; sessions point at it)
    bool pool_profile = false;       // root model: the pool's step graphs carry in-situ timing stamps (vc_pool_profile)
    bool fp8_kv = true;              // weight format 2: the KV cache of the bf16-step modes in e4m3 (vc_model_set_fp8_kv)
    bool batch_invariant = false;    // root model: a sample's bits do not depend on the batch it runs in (vc_model_set_batch_invariant)
    int qkv_fused = 1;               // root model: RoPE + head split + KV write in the prefill's QKV GEMM epilogue (vc_model_set_qkv_fused):
                                     // 0 off, 1 for problems the 256 x 256 GEMM kernel serves anyway (>= 1024 token rows), 2 always
    int pool_rows = VC_POOL_ROWS;    // root model: rows of the decode pool when it is next built (vc_pool_set_rows)
    std::atomic<bool> pool_hold{true};   // root model: the pool does not step while a call holding rows is still prefilling
                                         // (vc_pool_set_hold; written by any caller thread, read by the pool's driver thread)
    vc_model* root = nullptr;        // the model that owns the weights (itself for a root)
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    float t_encode = 0, t_prefill = 0, t_decode = 0;
};

struct vc_pool;
namespace {
void pool_destroy(vc_pool* p);
}

namespace {
#include "engine_weights.inc"
#include "engine_linears.inc"
#include "engine_vision.inc"
#include "engine_llm.inc"
}  // namespace

// the one-shot requests of the NEXT prefill / generate call (vc_set_image_counts, vc_set_attention_mask,
// vc_request_hidden_states) never outlive that call — also not when it fails half-way (a stale mask would be applied to,
// and a stale host pointer written by, some later call)
struct OneShotReset {
    vc_model* m;
    ~OneShotReset() {
        for (auto& v : m->img_counts) v.clear();
        m->mask_next.clear();
        m->hidden_out = nullptr;
        m->hidden_cap = 0;
        m->attn_out = nullptr;
        m->attn_cap = 0;
    }
};

#include "engine_abi.inc"
#include "engine_pool.inc"
#include "engine_profile.inc"
