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

template <class T> T* walloc(vc_model* m, size_t n, bool zero = false) {
    void* p = nullptr;
    HIPCHK(hipMalloc(&p, rup(n * sizeof(T), 256)));
    if (zero) {
        HIPCHK(hipMemsetAsync(p, 0, rup(n * sizeof(T), 256), m->st));
        HIPCHK(hipStreamSynchronize(m->st));
    }
    m->owned.push_back(p);
    return reinterpret_cast<T*>(p);
}

void mark_needed(vc_model* m) {
    auto& n = m->need;
    const vc_model_cfg& c = m->c;
    n["model.embed_tokens.weight"] = false;
    n["lm_head.weight"] = false;
    n["model.norm.weight"] = false;
    for (int i = 0; i < c.layers; ++i) {
        const std::string p = "model.layers." + std::to_string(i) + ".";
        for (const char* s : {"input_layernorm.weight", "post_attention_layernorm.weight", "self_attn.q_proj.weight",
                              "self_attn.k_proj.weight", "self_attn.v_proj.weight", "self_attn.o_proj.weight",
                              "mlp.gate_proj.weight", "mlp.up_proj.weight", "mlp.down_proj.weight"})
            n[p + s] = false;
    }
    auto proj = [&](const std::string& prefix, int depth) {
        if (depth == 1) {
            n[prefix + ".weight"] = false;
            n[prefix + ".bias"] = false;
        }
        for (int j = 0; depth > 1 && j < depth; ++j) {
            n[prefix + "." + std::to_string(2 * j) + ".weight"] = false;
            n[prefix + "." + std::to_string(2 * j) + ".bias"] = false;
        }
    };
    proj("model.mm_projector", c.mm_proj_depth);
    if (c.variant != VC_VARIANT_LLAVA) proj("model.seg_mm_projector", c.seg_proj_depth);
    n["vit.embeddings.class_embedding"] = false;
    n["vit.embeddings.patch_embedding.weight"] = false;
    n["vit.embeddings.position_embedding.weight"] = false;
    n["vit.pre_layrnorm.weight"] = false;
    n["vit.pre_layrnorm.bias"] = false;
    for (int j = 0; j < c.vit_layers_used; ++j) {
        const std::string p = "vit.encoder.layers." + std::to_string(j) + ".";
        for (const char* s : {"layer_norm1", "layer_norm2", "self_attn.q_proj", "self_attn.k_proj", "self_attn.v_proj",
                              "self_attn.out_proj", "mlp.fc1", "mlp.fc2"}) {
            n[p + s + ".weight"] = false;
            n[p + s + ".bias"] = false;
        }
    }
}

// canonical key: strips the CLIP prefixes of both Transformers generations
std::string canon_key(const std::string& k) {
    static const char* pre[] = {"model.vision_tower.vision_tower.vision_model.", "model.vision_tower.vision_tower.",
                                "vision_tower.vision_model.", "vision_model."};
    for (const char* p : pre) {
        const size_t n = strlen(p);
        if (k.compare(0, n, p) == 0) return "vit." + k.substr(n);
    }
    if (k.compare(0, 11, "embeddings.") == 0 || k.compare(0, 8, "encoder.") == 0 ||
        k.compare(0, 13, "pre_layrnorm.") == 0 || k.compare(0, 15, "post_layernorm.") == 0)
        return "vit." + k;
    return k;
}

// the lo plane of a weight (hi-plane pointer as loaded / packed), or nullptr: exact checkpoint, or none kept for it
const bf16_t* lo_plane(const vc_model* m, const void* hi) {
    const vc_model* r = m->root ? m->root : m;
    auto it = r->lo_of.find(hi);
    return it == r->lo_of.end() ? nullptr : it->second;
}

// `n` elements of `src` -> elements [off, off + n) of the matrix at `base` (base_elems elements in all: q / k / v land in one
// matrix).  fp32 sources that bf16 cannot hold exactly get (and from then on fill) the matrix's lo plane.
void to_bf16(vc_model* m, bf16_t* base, size_t off, size_t base_elems, const void* src, int dtype, size_t n) {
    bf16_t* dst = base + off;
    auto it = m->lo_of.find(base);
    bf16_t* lo = it == m->lo_of.end() ? nullptr : it->second;
    if (dtype == VC_BF16) {
#if VC_OPERAND_FP16
        // the fp16-operand build: bfloat16 checkpoint bits are widened and take the fp32 path (fp16 holds every bf16 value of
        // magnitude 2^-17 .. 65504 exactly; what it cannot hold gets a lo plane like any other inexact tensor)
        m->cvt_tmp.ensure(n * 4);
        launch_truebf16_to_f32(reinterpret_cast<const uint16_t*>(src), m->cvt_tmp.as<float>(), n, m->st);
        src = m->cvt_tmp.p;
#else
        HIPCHK(hipMemcpyAsync(dst, src, n * 2, hipMemcpyDeviceToDevice, m->st));
        // a key loaded twice — an fp16 / fp32 source first (which made the lo plane), then a bf16 override (a projector-only
        // checkpoint saved in bf16): the exact reload must not leave the first load's lo values behind
        if (lo) HIPCHK(hipMemsetAsync(lo + off, 0, n * 2, m->st));
        m->inexact_regions.erase({base, off});
        return;
#endif
    }
    if (!m->inexact_flag) m->inexact_flag = walloc<unsigned>(m, 64, true);
    HIPCHK(hipMemsetAsync(m->inexact_flag, 0, 4, m->st));
    launch_f32_to_bf16_planes(reinterpret_cast<const float*>(src), dst, lo ? lo + off : nullptr, n, m->inexact_flag, m->st);
    unsigned flag = 0;
    HIPCHK(hipMemcpyAsync(&flag, m->inexact_flag, 4, hipMemcpyDeviceToHost, m->st));
    HIPCHK(hipStreamSynchronize(m->st));
    if (!flag) {
        m->inexact_regions.erase({base, off});   // (an exact reload of a region: its lo values were just rewritten as zeros)
        return;
    }
    m->inexact_regions.insert({base, off});      // counted once per tensor, however often it is loaded
    if (!lo) {
        lo = walloc<bf16_t>(m, base_elems, true);   // zero: the parts of the matrix loaded from exact data
        m->lo_of[base] = lo;
        launch_f32_to_bf16_planes(reinterpret_cast<const float*>(src), dst, lo + off, n, nullptr, m->st);
    }
}
void to_bf16(vc_model* m, bf16_t* dst, const void* src, int dtype, size_t n) { to_bf16(m, dst, 0, n, src, dtype, n); }
void to_f32(vc_model* m, float* dst, const void* src, int dtype, size_t n) {
    if (dtype == VC_F32) HIPCHK(hipMemcpyAsync(dst, src, n * 4, hipMemcpyDeviceToDevice, m->st));
    else launch_truebf16_to_f32(reinterpret_cast<const uint16_t*>(src), dst, n, m->st);   // checkpoint bits: bfloat16 in every build
}

// place a tensor that already sits on the device (`src`, dtype) into its inference layout
int place_tensor(vc_model* m, const std::string& raw_key, const void* src, int dtype, const int64_t* shape, int ndim) {
    const vc_model_cfg& c = m->c;
    const std::string key = canon_key(raw_key);
    size_t numel = 1;
    for (int i = 0; i < ndim; ++i) numel *= (size_t)shape[i];
    // full shape comparison (a transposed [K,N] matrix has the right element count and the wrong meaning); dimensions of
    // extent 1 are ignored so that [D] / [1,D] / [D,1,1] spellings of a vector all pass, and the patch embedding may come
    // as [Dv,3,P,P] or flattened [Dv, 3*P*P]
    auto expect = [&](std::initializer_list<int64_t> dims) {
        std::vector<int64_t> want, got;
        for (auto d : dims)
            if (d != 1) want.push_back(d);
        for (int i = 0; i < ndim; ++i)
            if (shape[i] != 1) got.push_back(shape[i]);
        bool ok = want == got;
        if (!ok && want.size() == 2 && got.size() > 2 && got[0] == want[0]) {  // conv weight [out, c, kh, kw]
            int64_t r = 1;
            for (size_t i = 1; i < got.size(); ++i) r *= got[i];
            ok = r == want[1];
        }
        if (!ok) {
            std::string w, g;
            for (auto d : want) w += (w.empty() ? "" : ",") + std::to_string(d);
            for (auto d : got) g += (g.empty() ? "" : ",") + std::to_string(d);
            REQUIRE(false, VC_ERR_INVALID, "%s: expected shape [%s], got [%s]", raw_key.c_str(), w.c_str(), g.c_str());
        }
    };
    const int D = c.hidden, F = c.ffn, V = c.vocab, Dv = c.vit_hidden, Fv = c.vit_ffn;
    // ---- dead at inference or simply unused
    if (key.find("depth_mm_projector") != std::string::npos || key.find("mm2_projector") != std::string::npos ||
        key == "model.vcoder_lm_emb.weight" || key.compare(0, 19, "vit.post_layernorm.") == 0 ||
        key.find("position_ids") != std::string::npos || key.find("rotary_emb.inv_freq") != std::string::npos)
        return VC_IGNORED;
    if (c.variant == VC_VARIANT_LLAVA && key.find("seg_mm_projector") != std::string::npos) return VC_IGNORED;
    auto it = m->need.find(key);
    int layer = -1;
    char rest[128] = {0};
    if (sscanf(key.c_str(), "vit.encoder.layers.%d.%127s", &layer, rest) == 2 && layer >= c.vit_layers_used)
        return VC_IGNORED;  // layers after hidden_states[select_layer] are never evaluated
    REQUIRE(it != m->need.end(), VC_ERR_INVALID, "unexpected tensor key '%s'", raw_key.c_str());

    if (key == "model.embed_tokens.weight") { expect({V, D}); to_bf16(m, m->embed, src, dtype, numel); }
    else if (key == "lm_head.weight") { expect({V, D}); to_bf16(m, m->lm_head, src, dtype, numel); }
    else if (key == "model.norm.weight") { expect({D}); to_f32(m, m->final_norm, src, dtype, numel); }
    else if (sscanf(key.c_str(), "model.layers.%d.%127s", &layer, rest) == 2) {
        REQUIRE(layer >= 0 && layer < c.layers, VC_ERR_INVALID, "layer index out of range in %s", raw_key.c_str());
        LlmLayer& L = m->llm[layer];
        const std::string r = rest;
        if (r == "input_layernorm.weight") { expect({D}); to_f32(m, L.in_norm, src, dtype, numel); }
        else if (r == "post_attention_layernorm.weight") { expect({D}); to_f32(m, L.post_norm, src, dtype, numel); }
        else if (r == "self_attn.q_proj.weight") { expect({D, D}); to_bf16(m, L.qkv_w, 0, (size_t)3 * D * D, src, dtype, numel); }
        else if (r == "self_attn.k_proj.weight") { expect({D, D}); to_bf16(m, L.qkv_w, (size_t)D * D, (size_t)3 * D * D, src, dtype, numel); }
        else if (r == "self_attn.v_proj.weight") { expect({D, D}); to_bf16(m, L.qkv_w, (size_t)2 * D * D, (size_t)3 * D * D, src, dtype, numel); }
        else if (r == "self_attn.o_proj.weight") { expect({D, D}); to_bf16(m, L.o_w, src, dtype, numel); }
        else if (r == "mlp.gate_proj.weight") { expect({F, D}); to_bf16(m, L.gate_tmp, src, dtype, numel); }
        else if (r == "mlp.up_proj.weight") { expect({F, D}); to_bf16(m, L.up_tmp, src, dtype, numel); }
        else if (r == "mlp.down_proj.weight") { expect({D, F}); to_bf16(m, L.down_w, src, dtype, numel); }
        else REQUIRE(false, VC_ERR_INVALID, "unexpected tensor key '%s'", raw_key.c_str());
    } else if (key.compare(0, 19, "model.mm_projector.") == 0 || key.compare(0, 23, "model.seg_mm_projector.") == 0) {
        const bool is_seg = key.compare(0, 23, "model.seg_mm_projector.") == 0;
        Projector& pj = is_seg ? m->seg : m->mm;
        const std::string r = key.substr(is_seg ? 23 : 19);
        int idx = 0;
        char what[32] = {0};
        if (sscanf(r.c_str(), "%d.%31s", &idx, what) == 2) idx /= 2;
        else { idx = 0; snprintf(what, sizeof what, "%s", r.c_str()); }
        REQUIRE(idx >= 0 && idx < pj.depth, VC_ERR_INVALID, "projector layer out of range in %s", raw_key.c_str());
        const int in = idx == 0 ? Dv : D;
        if (!strcmp(what, "weight")) { expect({D, in}); to_bf16(m, pj.w[idx], src, dtype, numel); }
        else { expect({D}); to_f32(m, pj.b[idx], src, dtype, numel); }
    } else if (key == "vit.embeddings.class_embedding") { expect({Dv}); to_f32(m, m->vit_cls, src, dtype, numel); }
    else if (key == "vit.embeddings.position_embedding.weight") { expect({m->Tv, Dv}); to_f32(m, m->vit_pos, src, dtype, numel); }
    else if (key == "vit.embeddings.patch_embedding.weight") {
        expect({Dv, m->Kpatch});
        m->stage2.ensure(numel * 2);
        m->lo_of.erase(m->stage2.p);
        to_bf16(m, m->stage2.as<bf16_t>(), src, dtype, numel);
        HIPCHK(hipMemcpy2DAsync(m->vit_patch_w, (size_t)m->Kpad * 2, m->stage2.p, (size_t)m->Kpatch * 2,
                                (size_t)m->Kpatch * 2, Dv, hipMemcpyDeviceToDevice, m->st));
        if (auto it2 = m->lo_of.find(m->stage2.p); it2 != m->lo_of.end()) {   // the staged lo plane -> the padded [Dv, Kpad] layout
            bf16_t* lo_pad = walloc<bf16_t>(m, (size_t)Dv * m->Kpad, true);
            HIPCHK(hipMemcpy2DAsync(lo_pad, (size_t)m->Kpad * 2, it2->second, (size_t)m->Kpatch * 2, (size_t)m->Kpatch * 2, Dv,
                                    hipMemcpyDeviceToDevice, m->st));
            m->lo_of.erase(it2);
            m->lo_of[m->vit_patch_w] = lo_pad;
        }
    } else if (key == "vit.pre_layrnorm.weight") { expect({Dv}); to_f32(m, m->vit_pre_w, src, dtype, numel); }
    else if (key == "vit.pre_layrnorm.bias") { expect({Dv}); to_f32(m, m->vit_pre_b, src, dtype, numel); }
    else if (sscanf(key.c_str(), "vit.encoder.layers.%d.%127s", &layer, rest) == 2) {
        VitLayer& L = m->vit[layer];
        const std::string r = rest;
        const size_t DD = (size_t)Dv * Dv;
        if (r == "layer_norm1.weight") { expect({Dv}); to_f32(m, L.ln1_w, src, dtype, numel); }
        else if (r == "layer_norm1.bias") { expect({Dv}); to_f32(m, L.ln1_b, src, dtype, numel); }
        else if (r == "layer_norm2.weight") { expect({Dv}); to_f32(m, L.ln2_w, src, dtype, numel); }
        else if (r == "layer_norm2.bias") { expect({Dv}); to_f32(m, L.ln2_b, src, dtype, numel); }
        else if (r == "self_attn.q_proj.weight") { expect({Dv, Dv}); to_bf16(m, L.qkv_w, 0, 3 * DD, src, dtype, numel); }
        else if (r == "self_attn.k_proj.weight") { expect({Dv, Dv}); to_bf16(m, L.qkv_w, DD, 3 * DD, src, dtype, numel); }
        else if (r == "self_attn.v_proj.weight") { expect({Dv, Dv}); to_bf16(m, L.qkv_w, 2 * DD, 3 * DD, src, dtype, numel); }
        else if (r == "self_attn.q_proj.bias") { expect({Dv}); to_f32(m, L.qkv_b, src, dtype, numel); }
        else if (r == "self_attn.k_proj.bias") { expect({Dv}); to_f32(m, L.qkv_b + Dv, src, dtype, numel); }
        else if (r == "self_attn.v_proj.bias") { expect({Dv}); to_f32(m, L.qkv_b + 2 * Dv, src, dtype, numel); }
        else if (r == "self_attn.out_proj.weight") { expect({Dv, Dv}); to_bf16(m, L.out_w, src, dtype, numel); }
        else if (r == "self_attn.out_proj.bias") { expect({Dv}); to_f32(m, L.out_b, src, dtype, numel); }
        else if (r == "mlp.fc1.weight") { expect({Fv, Dv}); to_bf16(m, L.fc1_w, src, dtype, numel); }
        else if (r == "mlp.fc1.bias") { expect({Fv}); to_f32(m, L.fc1_b, src, dtype, numel); }
        else if (r == "mlp.fc2.weight") { expect({Dv, Fv}); to_bf16(m, L.fc2_w, src, dtype, numel); }
        else if (r == "mlp.fc2.bias") { expect({Dv}); to_f32(m, L.fc2_b, src, dtype, numel); }
        else REQUIRE(false, VC_ERR_INVALID, "unexpected tensor key '%s'", raw_key.c_str());
    } else {
        REQUIRE(false, VC_ERR_INVALID, "unexpected tensor key '%s'", raw_key.c_str());
    }
    it->second = true;
    HIPCHK(hipStreamSynchronize(m->st));  // staging buffers are reused by the next call
    return VC_OK;
}

// output_hidden_states hook (defined with the prefill layers); x_src: the residual rows to copy (default: the prefill's m->x)
void emit_hidden(vc_model* m, int idx, int B, int S, const float* x_src = nullptr);
// output_attentions hook: the probabilities of decoder layer l from its q / k in the precision mode's own form.  S queries
// starting at position q_pos0 against Tk keys (0: S — a prefill; a cached decode step: S = 1, Tk = its position + 1)
void emit_attentions(vc_model* m, int l, int B, int S, AttnProbsArgs a, int Tk = 0, int q_pos0 = 0) {
    if (!m->attn_out) return;
    const int keys = Tk > 0 ? Tk : S;
    const size_t n = (size_t)B * m->c.heads * S * keys;
    REQUIRE((size_t)(l + 1) * n <= m->attn_cap, VC_ERR_INVALID, "attention buffer too small: %zu floats for layer %d of %zu", m->attn_cap,
            l, n);
    REQUIRE(keys <= 4096, VC_ERR_INVALID, "output_attentions: at most 4096 positions");
    m->hidden_tmp.ensure(n * 4);
    a.out = m->hidden_tmp.as<float>();
    a.B = B;
    a.H = m->c.heads;
    a.T = S;
    a.Tk = Tk;
    a.q_pos0 = q_pos0;
    a.hd = m->hd;
    a.scale = 1.0f / sqrtf((float)m->hd);
    if (Tk > 0 ? m->kmask_in_decode : m->has_kmask) {
        a.key_mask = m->kmask.as<uint8_t>();
        a.mask_stride = m->c.max_positions;
    }
    launch_attn_probs(a, m->st);
    HIPCHK(hipMemcpyAsync(m->attn_out + (size_t)l * n, a.out, n * 4, hipMemcpyDeviceToHost, m->st));
    HIPCHK(hipStreamSynchronize(m->st));   // hidden_tmp is shared with the hidden-state hook
}

// ------------------------------------------------------------------------------------------------
// GEMM helpers
// vc_model_set_batch_invariant: no split-K remainder round (its slices — and so the order in which a row's k-blocks are summed —
// depend on the number of output tiles, i.e. on how many rows share the launch)
inline bool batch_invariant(const vc_model* m) { return (m->root ? m->root : m)->batch_invariant; }
inline const vc_model* root_of(const vc_model* m) { return m->root ? m->root : m; }
inline bool prefill_fold_on() {
    const char* e = getenv("VC_PREFILL_FOLD");
    return e && atoi(e) != 0;
}
// folded RMSNorm of a prefill (GemmArgs::row_scale / xg_out): what a GEMM consumes and what it hands to the next one
struct NormFold {
    const float* row_scale = nullptr;   // consumer: 1/rms per row
    bf16_t* xg_out = nullptr;           // producer (EPI_RESID_F32): the next GEMM's operand rows ...
    const float* xg_w = nullptr;        // ... = bf16(x * xg_w)
    float* ssq_out = nullptr;
    int ld_xg = 0, xg_lo = 0, npart = 0;
};
static void apply_fold(GemmArgs& a, const NormFold* f) {
    if (!f) return;
    a.row_scale = f->row_scale;
    a.xg_out = f->xg_out;
    a.xg_w = f->xg_w;
    a.ssq_out = f->ssq_out;
    a.ld_xg = f->ld_xg;
    a.xg_lo = f->xg_lo;
    a.npart = f->npart;
}
void gemm(vc_model* m, const bf16_t* A, const bf16_t* W, const float* bias, void* out, int M, int N, int K, int ldo,
          int epi, int lda = 0, const NormFold* fold = nullptr, const QkvEpiArgs* qe = nullptr) {
    GemmArgs a{A, W, bias, out, M, N, K, lda > 0 ? lda : K, K, ldo};
    apply_fold(a, fold);
    if (qe) a.qe = *qe;
    if (!batch_invariant(m) && (long)((M + 255) / 256) * ((N + 255) / 256) > 256) {  // only problems with more than one round of tiles can use it
        m->gemm_ws.ensure((size_t)64 << 20);
        a.ws = m->gemm_ws.as<float>();
        a.ws_bytes = m->gemm_ws.cap;
    }
    launch_gemm(a, epi, m->st);
}
// weight format 2 (W8A8 prefill): the token rows of A are quantised to e4m3 with per-row power-of-two scales, then
// out = epi((Q @ Wq^T) * a_scale[m] * w_scale[n]) on the K=128 scaled MFMA
// A == nullptr: the e4m3 rows and scales are already in m->a8 / m->a8_scale (launch_rmsnorm_q8)
void gemm_f8(vc_model* m, const bf16_t* A, const uint8_t* Wq, const float* wscale, void* out, int M, int N, int K, int ldo,
             int epi, const QkvEpiArgs* qe = nullptr) {
    if (A) launch_quant_act_rows(A, K, m->a8.as<uint8_t>(), m->a8_scale.as<float>(), M, K, m->st);
    GemmArgs a{reinterpret_cast<const bf16_t*>(m->a8.p), reinterpret_cast<const bf16_t*>(Wq), nullptr, out, M, N, K, K, K, ldo};
    if (qe) a.qe = *qe;
    if (!batch_invariant(m) && (long)((M + 255) / 256) * ((N + 255) / 256) > 256) {
        m->gemm_ws.ensure((size_t)64 << 20);
        a.ws = m->gemm_ws.as<float>();
        a.ws_bytes = m->gemm_ws.cap;
    }
    a.f8 = 1;
    a.a_scale = m->a8_scale.as<float>();
    a.w_scale = wscale;
    launch_gemm(a, epi, m->st);
}
// precision mode "split": A is the K-concatenated [hi | lo] bf16 image of an fp32 activation matrix (row stride lda >= 2 Kw),
// W [N, Kw] is contracted against both halves (kwrap); split_out > 0: a bf16-valued epilogue writes [hi | lo] again, the lo
// plane split_out columns to the right
void gemm_split(vc_model* m, const bf16_t* A, const bf16_t* W, const float* bias, void* out, int M, int N, int Kw, int ldo,
                int epi, int lda, int split_out = 0, const NormFold* fold = nullptr) {
    GemmArgs a{A, W, bias, out, M, N, 2 * Kw, lda, Kw, ldo};
    a.kwrap = Kw / 64;
    a.split_out = split_out;
    if (const bf16_t* Wl = lo_plane(m, W)) {   // an inexact checkpoint: a third K segment, a_hi . w_lo (gemm.hip w_koff)
        a.K = 3 * Kw;
        a.w_lo_off = (long long)(reinterpret_cast<const char*>(Wl) - reinterpret_cast<const char*>(W));
    }
    apply_fold(a, fold);
    if (!batch_invariant(m) && (long)((M + 255) / 256) * ((N + 255) / 256) > 256) {
        m->gemm_ws.ensure((size_t)64 << 20);
        a.ws = m->gemm_ws.as<float>();
        a.ws_bytes = m->gemm_ws.cap;
    }
    launch_gemm(a, epi, m->st);
}
// row stride of a [hi | lo] operand of width K: padded like XN_PAD (2 K bf16 is a power-of-two stride at K = 4096)
inline int split_ld(int K) { return 2 * K + XN_PAD; }

// split-K buffers of the decode GEMVs: the widest [K-slices][output tiles] product of the model's matrices (the workgroup-shared
// form slices qkv / gate-up / lm_head as well: at most 8 slices, decode.hip wg_geometry), two row groups of 256 floats each
inline size_t sk_floats(const vc_model_cfg& c) {
    const size_t tiles = (size_t)std::max(std::max(3 * c.hidden, 2 * c.ffn), c.vocab) / 16 + 1;
    return std::max((size_t)4 * 512, (size_t)8 * tiles) * 2 * 256;
}
inline int sk_counters_n(const vc_model_cfg& c) { return (int)((size_t)std::max(std::max(3 * c.hidden, 2 * c.ffn), c.vocab) / 16 + 1) * 2; }

// bytes per KV-cache element of precision mode "split": 3 = fp24 (default: hd x u16 | hd x u8 per row, 2^-17 relative, 0.75 of the
// fp32 bytes the decode attention streams), 4 = fp32 (VC_SPLIT_KV=32: regression / A-B)
inline int split_kv_es() {
    static const int es = (getenv("VC_SPLIT_KV") && atoi(getenv("VC_SPLIT_KV")) == 32) ? 4 : 3;
    return es;
}

// the fp8 weight format (2) keeps its KV cache in e4m3 as well (1 byte per element: at 13b the pooled decode attention reads 2.6x
// the bytes of the e4m3 weights otherwise); vc_model_set_fp8_kv(m, 0) keeps bf16 rows (root model's setting, before finalize)
// bytes per KV element of the bf16-step modes (precision 0) of a model
inline int step_kv_es(const vc_model* m) {
    const vc_model* r = m->root ? m->root : m;
    return (r->weight_format == 2 && r->fp8_kv) ? 1 : 2;
}

// Everything one decode step touches besides the weights: the buffers of a session's own loop or of the shared pool.
struct LoopView {
    hipStream_t st;
    bf16_t *kc, *vc;   // [L][capR][H][capS][hd]: K and V, both key-major (fp32 elements when es == 4)
    int es;            // bytes per cache element: 2 (bf16); precision mode "split": 3 (fp24) or 4 (fp32)
    int split_G;       // 0: bf16 decode step.  G = 8 / 16: split decode step — xg_dec / attn_dec / h_dec hold stacked groups of
                       // G bf16 hi rows + G lo rows (row r -> group r / G), qkv_dec is fp32
    int capR, capS;
    int* rows;         // RowState records
    const uint8_t* kmask;  // keys hidden from the rows' decode steps ([rows][kmask_stride] bytes, 0 = hidden), or nullptr
    int kmask_stride;
    float* x_dec;
    bf16_t *xg_dec, *qkv_dec, *attn_dec, *h_dec;
    float* logits;
    int *next_tok, *out_ids;
    float *ssq, *sk_scratch;
    unsigned* sk_counters;
    // in-situ timing (vc_pool_profile): slot s of the step = stamps + s * STAMP_SLOT_WORDS; *stamp_next = the next free slot while
    // the step is being enqueued; prof_acc = this span's accumulators.  All nullptr when off.
    unsigned* stamps;
    int* stamp_next;
    unsigned long long* prof_acc;
    unsigned* stamp_scratch;
};

// the next timing slot of the step being enqueued (nullptr: profiling off)
inline unsigned* next_stamp(const LoopView& v) {
    if (!v.stamps || !v.stamp_next) return nullptr;
    return v.stamps + STAMP_SLOT_WORDS * (size_t)(*v.stamp_next)++;
}

// decode-time linear over `X` (bf16 [M, K]).  use_rstd: X is the xg operand (bf16(x * g)) and the output is scaled by the
// rows' 1/rms from the ssq partials; next_norm_w (RESID epilogue only): publish the ssq partials of the updated residual
// rows and the next consumer's xg operand
// split decode step (v.split_G = G): X and every bf16-valued output are stacked groups of G hi rows + G lo rows, one
// weight pass serves one group (G rows: hi + lo fill the kernel's 16 / 32 token slots)
void gemv(vc_model* m, const LoopView& v, const bf16_t* X, const bf16_t* Wp, const float* wscale, void* out, int M, int N, int K,
          int ldo, int epi, bool use_rstd = false, const float* next_norm_w = nullptr) {
    const bool f32out = epi == GEMV_F32 || epi == GEMV_RESID_F32;
    const size_t esz = f32out ? 4 : 2;
    const int np = m->npart;
    const int G = v.split_G;
    const int pass = G ? G : VC_GEMV_MAX_M;
    for (int m0 = 0; m0 < M; m0 += pass) {  // the skinny kernel holds VC_GEMV_MAX_M token slots per weight pass
        const size_t xrow0 = G ? (size_t)(m0 / G) * 2 * G : (size_t)m0;   // first row of the pass in a stacked bf16 buffer
        GemvArgs a{};
        a.X = X + xrow0 * K;
        a.Wp = Wp;
        a.wscale = wscale;
        a.out = reinterpret_cast<char*>(out) + (f32out ? (size_t)m0 : xrow0) * ldo * esz;
        a.M = std::min(pass, M - m0);
        a.N = N;
        a.K = K;
        a.ldo = ldo;
        a.split_rows = G;
        a.ssq_in = use_rstd ? v.ssq + (size_t)m0 * np : nullptr;
        if (next_norm_w) {
            a.ssq_out = v.ssq + (size_t)m0 * np;
            a.xg_w = next_norm_w;
            a.xg_out = v.xg_dec + xrow0 * N;
        }
        a.npart = np;
        a.eps = m->c.rms_eps;
        // the launcher may split K over several workgroups per tile (per-wave rings: o_proj / down only; the workgroup-shared
        // form: any matrix)
        a.sk_scratch = v.sk_scratch;
        a.sk_counters = v.sk_counters;
        a.sk_scratch_floats = sk_floats(m->c);
        a.sk_counters_n = sk_counters_n(m->c);
        a.stamp = next_stamp(v);
        if (G) a.Wp_lo = lo_plane(m, Wp);   // split mode on an inexact checkpoint
        launch_gemv(a, epi, v.st);
    }
}

// the GEMVs of one decode step over the first M rows; `between(l)` runs after the qkv projection of layer l (the attention)
template <class F, class G>
void decode_linears(vc_model* m, const LoopView& v, int M, F&& between, G&& after_layer) {
    const vc_model_cfg& c = m->c;
    const int D = c.hidden, Fd = c.ffn;
    const bf16_t* xg = v.xg_dec;
    for (int l = 0; l < c.layers; ++l) {
        const LlmLayer& L = m->llm[l];
        const float* next_in = l + 1 < c.layers ? m->llm[l + 1].in_norm : m->final_norm;
        gemv(m, v, xg, L.qkv_p, L.qkv_s, v.qkv_dec, M, 3 * D, D, 3 * D, v.split_G ? GEMV_F32 : GEMV_BF16, true);  // K11+K12
        between(l);                                                                                          // K13-K15
        gemv(m, v, v.attn_dec, L.o_p, L.o_s, v.x_dec, M, D, D, D, GEMV_RESID_F32, false, L.post_norm);          // K16
        gemv(m, v, xg, L.gu_p, L.gu_s, v.h_dec, M, 2 * Fd, D, Fd, GEMV_SWIGLU, true);                          // K11+K17
        gemv(m, v, v.h_dec, L.down_p, L.down_s, v.x_dec, M, D, Fd, D, GEMV_RESID_F32, false, next_in);          // K17
        after_layer(l);
    }
    gemv(m, v, xg, m->lm_head_p, nullptr, v.logits, M, c.vocab, D, c.vocab, GEMV_F32, true);                    // K11+K18
}
template <class F>
void decode_linears(vc_model* m, const LoopView& v, int M, F&& between) {
    decode_linears(m, v, M, between, [](int) {});
}

// ------------------------------------------------------------------------------------------------
// ViT: pixels of all modalities batched into ONE tower pass (the tower weights are shared).  A modality may carry
// any number of images (the reference's list / 5-D image form gives a sample several images, vcoder_ds_llava_arch.py:
// 135-143); the common 4-D form has one image per sample and modality.
struct PixSet {
    const float* p[3];  // IMAGE, SEG, DEPTH pixel blocks (nullptr = modality absent)
    int n[3];           // images in each block
};

// CLIPVisionTower.forward (clip_encoder.py:39-51) up to hidden_states[select_layer]: leaves the fp32 residual stream of
// all N images in v_x [N*Tv, Dv]; returns N and the block order
int run_vit_tower(vc_model* m, const PixSet& in, int pixels_on_device, int order[3], int first_img[3]) {
    const vc_model_cfg& c = m->c;
    const int Dv = c.vit_hidden, Fv = c.vit_ffn, H = c.vit_heads, Tv = m->Tv, P = m->P;
    int nmod = 0, N = 0;
    for (int k = 0; k < 3; ++k) {
        first_img[k] = 0;
        if (in.p[k] && in.n[k] > 0) {
            order[nmod++] = k;
            first_img[k] = N;
            N += in.n[k];
        }
    }
    REQUIRE(N > 0, VC_ERR_INVALID, "no images");
    const size_t img_elems = (size_t)3 * c.vit_image * c.vit_image;
    m->v_pixels.ensure((size_t)N * img_elems * 4);
    for (int i = 0; i < nmod; ++i) {
        const int k = order[i];
        HIPCHK(hipMemcpyAsync(m->v_pixels.as<float>() + (size_t)first_img[k] * img_elems, in.p[k],
                              (size_t)in.n[k] * img_elems * 4,
                              pixels_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, m->st));
    }
    const int M = N * Tv, Mp = N * P;
    const int Ts = (int)rup(Tv, 64);
    m->v_cols.ensure((size_t)Mp * m->Kpad * 2);
    m->v_patches.ensure((size_t)Mp * Dv * 4);
    m->v_x.ensure((size_t)M * Dv * 4);
    m->v_xn.ensure((size_t)M * Dv * 2);
    m->v_qkv.ensure((size_t)M * 3 * Dv * 2);
    m->v_q.ensure((size_t)N * H * Ts * m->vhd * 2, true);
    m->v_k.ensure((size_t)N * H * Ts * m->vhd * 2, true);
    m->v_vt.ensure((size_t)N * H * Ts * m->vhd * 2, true);
    m->v_attn.ensure((size_t)M * Dv * 2);
    m->v_h.ensure((size_t)M * Fv * 2);
    launch_im2col(m->v_pixels.as<float>(), m->v_cols.as<bf16_t>(), N, c.vit_image, c.vit_patch, m->Kpad, m->st);
    gemm(m, m->v_cols.as<bf16_t>(), m->vit_patch_w, nullptr, m->v_patches.p, Mp, Dv, m->Kpad, Dv, EPI_F32);
    launch_vit_embed_ln(m->v_patches.as<float>(), m->vit_cls, m->vit_pos, m->vit_pre_w, m->vit_pre_b, m->v_x.as<float>(),
                        N, Tv, Dv, c.vit_ln_eps, m->st);
    for (int j = 0; j < c.vit_layers_used; ++j) {
        const VitLayer& L = m->vit[j];
        launch_layernorm(m->v_x.as<float>(), L.ln1_w, L.ln1_b, m->v_xn.as<bf16_t>(), M, Dv, c.vit_ln_eps, m->st);
        gemm(m, m->v_xn.as<bf16_t>(), L.qkv_w, L.qkv_b, m->v_qkv.p, M, 3 * Dv, Dv, 3 * Dv, EPI_BF16);
        QkvSplitArgs qa{m->v_qkv.as<bf16_t>(), m->v_q.as<bf16_t>(), m->v_k.as<bf16_t>(), m->v_vt.as<bf16_t>(), N, Tv, H,
                        m->vhd, Ts, Ts, nullptr, nullptr, nullptr};
        launch_qkv_split(qa, m->st);
        AttnArgs aa{m->v_q.as<bf16_t>(), m->v_k.as<bf16_t>(), m->v_vt.as<bf16_t>(), m->v_attn.as<bf16_t>(), N, H, Tv,
                    m->vhd, Ts, Ts, 0, 1.0f / sqrtf((float)m->vhd)};
        launch_attention(aa, m->st);
        gemm(m, m->v_attn.as<bf16_t>(), L.out_w, L.out_b, m->v_x.p, M, Dv, Dv, Dv, EPI_RESID_F32);
        launch_layernorm(m->v_x.as<float>(), L.ln2_w, L.ln2_b, m->v_xn.as<bf16_t>(), M, Dv, c.vit_ln_eps, m->st);
        gemm(m, m->v_xn.as<bf16_t>(), L.fc1_w, L.fc1_b, m->v_h.p, M, Fv, Dv, Fv, EPI_BF16_QGELU);
        gemm(m, m->v_h.as<bf16_t>(), L.fc2_w, L.fc2_b, m->v_x.p, M, Dv, Fv, Dv, EPI_RESID_F32);
    }
    // feature_select (clip_encoder.py:29-37): hidden_states[select_layer] is the last layer evaluated; drop CLS for 'patch'
    const int skip = c.vit_keep_cls ? 0 : 1;
    const int R = Tv - skip;
    m->v_sel.ensure((size_t)N * R * Dv * 2);
    launch_select_rows_bf16(m->v_x.as<float>(), m->v_sel.as<bf16_t>(), N, Tv, skip, Dv, m->st);
    return N;
}

void run_vit_and_adapters(vc_model* m, const PixSet& in, int pixels_on_device) {
    const vc_model_cfg& c = m->c;
    const int Dv = c.vit_hidden, D = c.hidden;
    int order[3], first_img[3];
    const int N = run_vit_tower(m, in, pixels_on_device, order, first_img);
    const int R = m->Tv - (c.vit_keep_cls ? 0 : 1);  // feature rows per image
    m->feats.ensure((size_t)N * R * D * 2);
    m->v_mid.ensure((size_t)N * R * D * 2);
    for (int k = 0; k < 3; ++k) m->feat_rows[k] = 0;
    for (int mod = 0; mod < 3; ++mod) {
        if (!(in.p[mod] && in.n[mod] > 0)) continue;
        // images -> mm_projector; seg AND depth -> seg_mm_projector (quirk 1, vcoder_ds_llava_arch.py:111-114);
        // mm2_projector is unreachable (quirk 2, :137,145)
        const Projector& pj = mod == VC_MOD_IMAGE ? m->mm : m->seg;
        const int rows = in.n[mod] * R;
        const bf16_t* src = m->v_sel.as<bf16_t>() + (size_t)first_img[mod] * R * Dv;
        bf16_t* out = m->feats.as<bf16_t>() + (size_t)first_img[mod] * R * D;
        m->feat_off[mod] = first_img[mod] * R;
        m->feat_rows[mod] = rows;
        if (pj.depth == 0) {
            REQUIRE(Dv == D, VC_ERR_INVALID, "identity projector needs mm_hidden_size == hidden_size");
            HIPCHK(hipMemcpyAsync(out, src, (size_t)rows * D * 2, hipMemcpyDeviceToDevice, m->st));
            continue;
        }
        const bf16_t* cur = src;
        int K = Dv;
        for (int l = 0; l < pj.depth; ++l) {
            const bool last = l == pj.depth - 1;
            if (!last && l % 2 == 1) m->v_h.ensure((size_t)rows * D * 2);
            bf16_t* dst = last ? out : (l % 2 == 0 ? m->v_mid.as<bf16_t>() : m->v_h.as<bf16_t>());
            gemm(m, cur, pj.w[l], pj.b[l], dst, rows, D, K, D, last ? EPI_BF16 : EPI_BF16_GELU);
            cur = dst;
            K = D;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// SPLIT path (precision 2): the fast path's kernels and sequence of operations with every MFMA operand carried as bf16
// hi + lo (x = hi + lo to ~16 mantissa bits; weights are exactly bf16): GEMMs contract the [hi | lo] rows against the weight
// twice, attention uses three MFMAs per product, and everything between two MFMAs stays fp32.  Leaves the fp32 residual
// stream of all N images in v_x [N*Tv, Dv].
int run_vit_tower_split(vc_model* m, const PixSet& in, int pixels_on_device, int order[3], int first_img[3]) {
    const vc_model_cfg& c = m->c;
    const int Dv = c.vit_hidden, Fv = c.vit_ffn, H = c.vit_heads, Tv = m->Tv, P = m->P;
    int nmod = 0, N = 0;
    for (int k = 0; k < 3; ++k) {
        first_img[k] = 0;
        if (in.p[k] && in.n[k] > 0) {
            order[nmod++] = k;
            first_img[k] = N;
            N += in.n[k];
        }
    }
    REQUIRE(N > 0, VC_ERR_INVALID, "no images");
    const size_t img_elems = (size_t)3 * c.vit_image * c.vit_image;
    m->v_pixels.ensure((size_t)N * img_elems * 4);
    for (int i = 0; i < nmod; ++i) {
        const int k = order[i];
        HIPCHK(hipMemcpyAsync(m->v_pixels.as<float>() + (size_t)first_img[k] * img_elems, in.p[k],
                              (size_t)in.n[k] * img_elems * 4,
                              pixels_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, m->st));
    }
    const int M = N * Tv, Mp = N * P;
    const int Ts = (int)rup(Tv, 64);
    const int ldx = split_ld(Dv), ldh = split_ld(Fv);
    m->v_cols.ensure((size_t)Mp * 2 * m->Kpad * 2);
    m->v_patches.ensure((size_t)Mp * Dv * 4);
    m->v_x.ensure((size_t)M * Dv * 4);
    m->v_xn.ensure((size_t)M * ldx * 2);
    m->s_vqkv.ensure((size_t)M * 3 * Dv * 4);
    const size_t plane = (size_t)N * H * Ts * m->vhd;   // one bf16 plane of Q / K / V^T
    m->v_q.ensure(2 * plane * 2, true);
    m->v_k.ensure(2 * plane * 2, true);
    m->v_vt.ensure(2 * plane * 2, true);
    m->v_attn.ensure((size_t)M * ldx * 2);
    m->v_h.ensure((size_t)M * ldh * 2);
    launch_im2col(m->v_pixels.as<float>(), m->v_cols.as<bf16_t>(), N, c.vit_image, c.vit_patch, m->Kpad, m->st, true);
    gemm_split(m, m->v_cols.as<bf16_t>(), m->vit_patch_w, nullptr, m->v_patches.p, Mp, Dv, m->Kpad, Dv, EPI_F32, 2 * m->Kpad);
    launch_vit_embed_ln(m->v_patches.as<float>(), m->vit_cls, m->vit_pos, m->vit_pre_w, m->vit_pre_b, m->v_x.as<float>(),
                        N, Tv, Dv, c.vit_ln_eps, m->st);
    bf16_t *qh = m->v_q.as<bf16_t>(), *kh = m->v_k.as<bf16_t>(), *vh = m->v_vt.as<bf16_t>();
    for (int j = 0; j < c.vit_layers_used; ++j) {
        const VitLayer& L = m->vit[j];
        launch_layernorm_split(m->v_x.as<float>(), L.ln1_w, L.ln1_b, m->v_xn.as<bf16_t>(), M, Dv, c.vit_ln_eps, ldx, Dv, m->st);
        gemm_split(m, m->v_xn.as<bf16_t>(), L.qkv_w, L.qkv_b, m->s_vqkv.p, M, 3 * Dv, Dv, 3 * Dv, EPI_F32, ldx);
        QkvSplit32Args qa{m->s_vqkv.as<float>(), qh, qh + plane, kh, kh + plane, vh, vh + plane, nullptr, nullptr,
                          N, Tv, H, m->vhd, Ts, Ts, Ts, 0, nullptr, nullptr};
        launch_qkv_split32(qa, m->st);
        AttnArgs aa{qh, kh, vh, m->v_attn.as<bf16_t>(), N, H, Tv, m->vhd, Ts, Ts, 0, 1.0f / sqrtf((float)m->vhd), 0,
                    qh + plane, kh + plane, vh + plane, ldx, Dv};
        launch_attention(aa, m->st);
        gemm_split(m, m->v_attn.as<bf16_t>(), L.out_w, L.out_b, m->v_x.p, M, Dv, Dv, Dv, EPI_RESID_F32, ldx);
        launch_layernorm_split(m->v_x.as<float>(), L.ln2_w, L.ln2_b, m->v_xn.as<bf16_t>(), M, Dv, c.vit_ln_eps, ldx, Dv, m->st);
        gemm_split(m, m->v_xn.as<bf16_t>(), L.fc1_w, L.fc1_b, m->v_h.p, M, Fv, Dv, ldh, EPI_BF16_QGELU, ldx, Fv);
        gemm_split(m, m->v_h.as<bf16_t>(), L.fc2_w, L.fc2_b, m->v_x.p, M, Dv, Fv, Dv, EPI_RESID_F32, ldh);
    }
    return N;
}

// feature_select + adapters of the split path: projected features fp32 in s_feats (what the fp32 splice reads)
void run_vit_and_adapters_split(vc_model* m, const PixSet& in, int pixels_on_device) {
    const vc_model_cfg& c = m->c;
    const int Dv = c.vit_hidden, D = c.hidden;
    int order[3], first_img[3];
    const int N = run_vit_tower_split(m, in, pixels_on_device, order, first_img);
    const int skip = c.vit_keep_cls ? 0 : 1;
    const int R = m->Tv - skip;  // feature rows per image
    const int ldv = 2 * Dv, ldd = split_ld(D);
    m->v_sel.ensure((size_t)N * R * ldv * 2);
    launch_select_rows_bf16(m->v_x.as<float>(), m->v_sel.as<bf16_t>(), N, m->Tv, skip, Dv, m->st, true);
    m->s_feats.ensure((size_t)N * R * D * 4);
    m->v_mid.ensure((size_t)N * R * ldd * 2);
    for (int k = 0; k < 3; ++k) m->feat_rows[k] = 0;
    for (int mod = 0; mod < 3; ++mod) {
        if (!(in.p[mod] && in.n[mod] > 0)) continue;
        const Projector& pj = mod == VC_MOD_IMAGE ? m->mm : m->seg;  // quirk 1: depth -> seg_mm_projector
        const int rows = in.n[mod] * R;
        float* out = m->s_feats.as<float>() + (size_t)first_img[mod] * R * D;
        m->feat_off[mod] = first_img[mod] * R;
        m->feat_rows[mod] = rows;
        if (pj.depth == 0) {  // identity: the fp32 rows of hidden_states[select_layer]
            REQUIRE(Dv == D, VC_ERR_INVALID, "identity projector needs mm_hidden_size == hidden_size");
            m->s_sel.ensure((size_t)N * R * Dv * 4);
            launch_select_rows_f32(m->v_x.as<float>(), m->s_sel.as<float>(), N, m->Tv, skip, Dv, m->st);
            HIPCHK(hipMemcpyAsync(out, m->s_sel.as<float>() + (size_t)first_img[mod] * R * Dv, (size_t)rows * D * 4,
                                  hipMemcpyDeviceToDevice, m->st));
            continue;
        }
        const bf16_t* cur = m->v_sel.as<bf16_t>() + (size_t)first_img[mod] * R * ldv;
        int K = Dv, lda = ldv;
        for (int l = 0; l < pj.depth; ++l) {
            const bool last = l == pj.depth - 1;
            if (!last && l % 2 == 1) m->v_h.ensure((size_t)rows * ldd * 2);
            if (last) {
                gemm_split(m, cur, pj.w[l], pj.b[l], out, rows, D, K, D, EPI_F32, lda);
            } else {
                bf16_t* dst = l % 2 == 0 ? m->v_mid.as<bf16_t>() : m->v_h.as<bf16_t>();
                gemm_split(m, cur, pj.w[l], pj.b[l], dst, rows, D, K, ldd, EPI_BF16_GELU, lda, D);
                cur = dst;
                K = D;
                lda = ldd;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// STRICT (fp32-faithful) path: same sequence of operations, fp32 activations, strict.hip kernels, no graph.
void gemm32(vc_model* m, const float* A, const bf16_t* W, const float* bias, float* out, int M, int N, int K, int lda,
            int ldw, int ldo, int epi) {
    GemmF32Args a{A, W, bias, out, M, N, K, lda, ldw, ldo};
    a.W_lo = lo_plane(m, W);
    launch_gemm_f32(a, epi, m->st);
}

int run_vit_tower_strict(vc_model* m, const PixSet& in, int pixels_on_device, int order[3], int first_img[3]) {
    const vc_model_cfg& c = m->c;
    const int Dv = c.vit_hidden, Fv = c.vit_ffn, H = c.vit_heads, Tv = m->Tv, P = m->P, D = c.hidden;
    int nmod = 0, N = 0;
    for (int k = 0; k < 3; ++k) {
        first_img[k] = 0;
        if (in.p[k] && in.n[k] > 0) {
            order[nmod++] = k;
            first_img[k] = N;
            N += in.n[k];
        }
    }
    REQUIRE(N > 0, VC_ERR_INVALID, "no images");
    const size_t img_elems = (size_t)3 * c.vit_image * c.vit_image;
    m->v_pixels.ensure((size_t)N * img_elems * 4);
    for (int i = 0; i < nmod; ++i) {
        const int k = order[i];
        HIPCHK(hipMemcpyAsync(m->v_pixels.as<float>() + (size_t)first_img[k] * img_elems, in.p[k],
                              (size_t)in.n[k] * img_elems * 4,
                              pixels_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, m->st));
    }
    const int M = N * Tv, Mp = N * P;
    m->s_cols.ensure((size_t)Mp * m->Kpatch * 4);
    m->s_patches.ensure((size_t)Mp * Dv * 4);
    m->s_vx.ensure((size_t)M * Dv * 4);
    m->s_vxn.ensure((size_t)M * Dv * 4);
    m->s_vqkv.ensure((size_t)M * 3 * Dv * 4);
    m->s_vq.ensure((size_t)M * Dv * 4);
    m->s_vk.ensure((size_t)M * Dv * 4);
    m->s_vv.ensure((size_t)M * Dv * 4);
    m->s_vattn.ensure((size_t)M * Dv * 4);
    m->s_vh.ensure((size_t)M * std::max(Fv, D) * 4);
    float *x = m->s_vx.as<float>(), *xn = m->s_vxn.as<float>();
    launch_im2col_f32(m->v_pixels.as<float>(), m->s_cols.as<float>(), N, c.vit_image, c.vit_patch, m->st);
    gemm32(m, m->s_cols.as<float>(), m->vit_patch_w, nullptr, m->s_patches.as<float>(), Mp, Dv, m->Kpatch, m->Kpatch, m->Kpad,
           Dv, EPI_F32);
    launch_vit_embed_ln(m->s_patches.as<float>(), m->vit_cls, m->vit_pos, m->vit_pre_w, m->vit_pre_b, x, N, Tv, Dv,
                        c.vit_ln_eps, m->st);
    for (int j = 0; j < c.vit_layers_used; ++j) {
        const VitLayer& L = m->vit[j];
        launch_layernorm_f32(x, L.ln1_w, L.ln1_b, xn, M, Dv, c.vit_ln_eps, m->st);
        gemm32(m, xn, L.qkv_w, L.qkv_b, m->s_vqkv.as<float>(), M, 3 * Dv, Dv, Dv, Dv, 3 * Dv, EPI_F32);
        QkvF32Args qa{m->s_vqkv.as<float>(), m->s_vq.as<float>(), m->s_vk.as<float>(), m->s_vv.as<float>(), N, Tv, H, m->vhd,
                      Tv, Tv, nullptr, nullptr, nullptr};
        launch_qkv_rope_f32(qa, m->st);
        AttnF32Args aa{m->s_vq.as<float>(), m->s_vk.as<float>(), m->s_vv.as<float>(), m->s_vattn.as<float>(), N, H, Tv, m->vhd,
                       Tv, Tv, 0, Tv, nullptr, 1.0f / sqrtf((float)m->vhd)};
        launch_attention_f32(aa, m->st);
        gemm32(m, m->s_vattn.as<float>(), L.out_w, L.out_b, x, M, Dv, Dv, Dv, Dv, Dv, EPI_RESID_F32);
        launch_layernorm_f32(x, L.ln2_w, L.ln2_b, xn, M, Dv, c.vit_ln_eps, m->st);
        gemm32(m, xn, L.fc1_w, L.fc1_b, m->s_vh.as<float>(), M, Fv, Dv, Dv, Dv, Fv, EPI_BF16_QGELU);
        gemm32(m, m->s_vh.as<float>(), L.fc2_w, L.fc2_b, x, M, Dv, Fv, Fv, Fv, Dv, EPI_RESID_F32);
    }
    const int skip = c.vit_keep_cls ? 0 : 1;
    const int R = Tv - skip;
    m->s_sel.ensure((size_t)N * R * Dv * 4);
    launch_select_rows_f32(x, m->s_sel.as<float>(), N, Tv, skip, Dv, m->st);
    return N;
}

void run_vit_and_adapters_strict(vc_model* m, const PixSet& in, int pixels_on_device) {
    const vc_model_cfg& c = m->c;
    const int Dv = c.vit_hidden, D = c.hidden;
    int order[3], first_img[3];
    const int N = run_vit_tower_strict(m, in, pixels_on_device, order, first_img);
    const int R = m->Tv - (c.vit_keep_cls ? 0 : 1);
    m->s_feats.ensure((size_t)N * R * D * 4);
    m->s_mid.ensure((size_t)N * R * D * 4);
    for (int k = 0; k < 3; ++k) m->feat_rows[k] = 0;
    for (int mod = 0; mod < 3; ++mod) {
        if (!(in.p[mod] && in.n[mod] > 0)) continue;
        const Projector& pj = mod == VC_MOD_IMAGE ? m->mm : m->seg;  // quirk 1: depth -> seg_mm_projector
        const int rows = in.n[mod] * R;
        const float* src = m->s_sel.as<float>() + (size_t)first_img[mod] * R * Dv;
        float* out = m->s_feats.as<float>() + (size_t)first_img[mod] * R * D;
        m->feat_off[mod] = first_img[mod] * R;
        m->feat_rows[mod] = rows;
        if (pj.depth == 0) {
            REQUIRE(Dv == D, VC_ERR_INVALID, "identity projector needs mm_hidden_size == hidden_size");
            HIPCHK(hipMemcpyAsync(out, src, (size_t)rows * D * 4, hipMemcpyDeviceToDevice, m->st));
            continue;
        }
        const float* cur = src;
        int K = Dv;
        for (int l = 0; l < pj.depth; ++l) {
            const bool last = l == pj.depth - 1;
            float* dst = last ? out : (l % 2 == 0 ? m->s_mid.as<float>() : m->s_vh.as<float>());
            gemm32(m, cur, pj.w[l], pj.b[l], dst, rows, D, K, K, K, D, last ? EPI_F32 : EPI_BF16_GELU);
            cur = dst;
            K = D;
        }
    }
}

void ensure_strict(vc_model* m, int B, int Scap) {
    const vc_model_cfg& c = m->c;
    const int D = c.hidden, F = c.ffn, H = c.heads;
    if (B != m->s_capB || Scap > m->s_capS) {
        const size_t per_layer = (size_t)B * H * Scap * m->hd;
        m->s_kc.release();
        m->s_vc.release();
        m->s_kc.ensure(per_layer * c.layers * 4, true);
        m->s_vc.ensure(per_layer * c.layers * 4, true);
        m->s_capB = B;
        m->s_capS = Scap;
    }
    const size_t Mr = (size_t)B * Scap;
    m->s_xn.ensure(Mr * D * 4);
    m->s_qkv.ensure(Mr * 3 * D * 4);
    m->s_q.ensure(Mr * D * 4);
    m->s_attn.ensure(Mr * D * 4);
    m->s_h.ensure(Mr * F * 4);
    m->s_xl.ensure((size_t)rup(B, 16) * D * 4);
}
float* s_kcache(vc_model* m, int l) { return m->s_kc.as<float>() + (size_t)l * m->s_capB * m->c.heads * m->s_capS * m->hd; }
float* s_vcache(vc_model* m, int l) { return m->s_vc.as<float>() + (size_t)l * m->s_capB * m->c.heads * m->s_capS * m->hd; }

// one decoder stack pass over `T` new tokens per sample starting at the device-scalar position (prefill: pos 0).
// is_prefill says which of the two the pass is — a text-only prefill of ONE token has T == 1 too
void run_llm_layers_strict(vc_model* m, float* x, int B, int T, const int* pos_dev, bool is_prefill) {
    const vc_model_cfg& c = m->c;
    const int D = c.hidden, F = c.ffn, H = c.heads, M = B * T;
    float *xn = m->s_xn.as<float>(), *qkv = m->s_qkv.as<float>(), *q = m->s_q.as<float>(), *at = m->s_attn.as<float>(),
          *h = m->s_h.as<float>();
    const int nl = (m->layer_limit > 0 && is_prefill) ? std::min(m->layer_limit, c.layers) : c.layers;
    for (int l = 0; l < nl; ++l) {
        const LlmLayer& L = m->llm[l];
        launch_rmsnorm_f32(x, nullptr, L.in_norm, xn, M, D, c.rms_eps, m->st);
        gemm32(m, xn, L.qkv_w, nullptr, qkv, M, 3 * D, D, D, D, 3 * D, EPI_F32);
        QkvF32Args qa{qkv, q, s_kcache(m, l), s_vcache(m, l), B, T, H, m->hd, T, m->s_capS, pos_dev, m->rope_cos, m->rope_sin};
        launch_qkv_rope_f32(qa, m->st);
        AttnF32Args aa{q, s_kcache(m, l), s_vcache(m, l), at, B, H, T, m->hd, T, m->s_capS, 1, 0, pos_dev,
                       1.0f / sqrtf((float)m->hd)};
        if (is_prefill ? m->has_kmask : m->kmask_in_decode) {
            aa.key_mask = m->kmask.as<uint8_t>();
            aa.mask_stride = c.max_positions;
        }
        launch_attention_f32(aa, m->st);
        const bool step = !is_prefill && x == m->x_dec.as<float>();   // a session's cached decode step (m->cur_pos = its position)
        const bool pre = is_prefill && x == m->x.as<float>();
        if (pre || step) {
            AttnProbsArgs pa{};
            pa.q32 = q;
            pa.k32 = s_kcache(m, l);
            pa.q_stride = T;
            pa.kv_stride = m->s_capS;
            if (step) emit_attentions(m, l, B, 1, pa, m->cur_pos + 1, m->cur_pos);
            else emit_attentions(m, l, B, T, pa);
        }
        gemm32(m, at, L.o_w, nullptr, x, M, D, D, D, D, D, EPI_RESID_F32);
        launch_rmsnorm_f32(x, nullptr, L.post_norm, xn, M, D, c.rms_eps, m->st);
        gemm32(m, xn, L.gu_w, nullptr, h, M, 2 * F, D, D, D, F, EPI_SWIGLU);
        gemm32(m, h, L.down_w, nullptr, x, M, D, F, F, F, D, EPI_RESID_F32);
        if (pre) emit_hidden(m, l + 1, B, T);
        else if (step) emit_hidden(m, l + 1, B, 1, x);
    }
}

void logits_strict(vc_model* m, const float* x, const int* row_idx, int rows) {
    launch_rmsnorm_f32(x, row_idx, m->final_norm, m->s_xl.as<float>(), rows, m->c.hidden, m->c.rms_eps, m->st);
    gemm32(m, m->s_xl.as<float>(), m->lm_head, nullptr, m->logits.as<float>(), rows, m->c.vocab, m->c.hidden, m->c.hidden,
           m->c.hidden, m->c.vocab, EPI_F32);
}

// ------------------------------------------------------------------------------------------------
// splice planner (host).  One (kind, src) pair per destination row of inputs_embeds.
struct RowSrc { int kind, src; };

void plan_rows(vc_model* m, const int64_t* ids, int B, int T, bool has_seg, const std::vector<bool>* depth_zero,
               int R, std::vector<std::vector<RowSrc>>& out) {
    const int variant = m->c.variant;
    int img_i = 0, seg_i = 0, dep_i = 0;
    auto text = [&](std::vector<RowSrc>& rows, const int64_t* p, int n) {
        for (int i = 0; i < n; ++i) {
            REQUIRE(p[i] >= 0 && p[i] < m->c.vocab, VC_ERR_INDEX,
                    "index out of range in self (id %lld reached the embedding lookup)", (long long)p[i]);
            rows.push_back({0, (int)p[i]});
        }
    };
    // features[idx] of the reference: the block of sample idx — R rows per image, all images of the sample flattened
    // (`[x.flatten(0, 1) for x in image_features]`, vcoder_ds_llava_arch.py:143); one image per sample in the 4-D form
    auto feat = [&](std::vector<RowSrc>& rows, int mod, int idx, bool emit) {
        const std::vector<int>& first = m->img_first[mod];
        const int nblocks = first.empty() ? m->feat_rows[mod] / R : (int)first.size() - 1;
        REQUIRE(idx < nblocks, VC_ERR_INDEX, "index %d is out of bounds for dimension 0 with size %d", idx, nblocks);
        const int i0 = first.empty() ? idx : first[idx], i1 = first.empty() ? idx + 1 : first[idx + 1];
        for (int r = i0 * R; emit && r < i1 * R; ++r) rows.push_back({1, m->feat_off[mod] + r});
    };
    auto find = [](const int64_t* p, int n, int tok) {
        for (int i = 0; i < n; ++i)
            if (p[i] == tok) return i;
        return -1;
    };
    out.assign(B, {});
    for (int b = 0; b < B; ++b) {
        const int64_t* cur = ids + (size_t)b * T;
        int n = T;
        std::vector<RowSrc>& rows = out[b];
        int n_img = 0, n_seg = 0;
        for (int i = 0; i < T; ++i) {
            n_img += cur[i] == IMAGE_TOKEN_INDEX;
            n_seg += cur[i] == SEG_TOKEN_INDEX;
        }
        // "not multimodal" guard: llava_arch.py:118, vcoder_llava_arch.py:187 (`or`), vcoder_ds_llava_arch.py:181 (`and`)
        const bool plain = variant == VC_VARIANT_LLAVA ? n_img == 0
                           : variant == VC_VARIANT_VCODER ? (n_img == 0 || n_seg == 0)
                                                          : (n_img == 0 && n_seg == 0);
        if (plain) {
            feat(rows, VC_MOD_IMAGE, img_i, false);  // the reference indexes the features before embedding the ids
            if (variant != VC_VARIANT_LLAVA && has_seg) feat(rows, VC_MOD_SEG, seg_i, false);
            text(rows, cur, n);
            ++img_i; ++seg_i; ++dep_i;
            continue;
        }
        for (int at; (at = find(cur, n, IMAGE_TOKEN_INDEX)) >= 0;) {
            feat(rows, VC_MOD_IMAGE, img_i, false);
            text(rows, cur, at);
            feat(rows, VC_MOD_IMAGE, img_i++, true);
            cur += at + 1;
            n -= at + 1;
        }
        if (variant != VC_VARIANT_LLAVA && has_seg) {
            for (int at; (at = find(cur, n, SEG_TOKEN_INDEX)) >= 0;) {
                feat(rows, VC_MOD_SEG, seg_i, false);
                if (variant == VC_VARIANT_VCODER) text(rows, cur, at);  // vcoder_llava_arch.py:236 keeps the text,
                feat(rows, VC_MOD_SEG, seg_i++, true);                  // vcoder_ds_llava_arch.py:238 drops it
                cur += at + 1;
                n -= at + 1;
            }
        }
        if (variant == VC_VARIANT_VCODER_DS) {
            // a row with several <depth> placeholders advances the depth index more than once: later rows then index past
            // the per-sample list, where the reference raises IndexError (vcoder_ds_llava_arch.py:246)
            REQUIRE(!depth_zero || dep_i < (int)depth_zero->size(), VC_ERR_INDEX, "list index out of range (depth index %d of %zu)",
                    dep_i, depth_zero ? depth_zero->size() : (size_t)0);
            const bool dz = depth_zero ? (*depth_zero)[dep_i] : true;
            if (!dz) {
                for (int at; (at = find(cur, n, DEPTH_TOKEN_INDEX)) >= 0;) {
                    feat(rows, VC_MOD_DEPTH, dep_i, false);
                    text(rows, cur, at);
                    feat(rows, VC_MOD_DEPTH, dep_i++, true);
                    cur += at + 1;
                    n -= at + 1;
                }
            } else {
                ++dep_i;
            }
        }
        if (n > 0) text(rows, cur, n);
    }
}

// workspaces of a prefill of B sequences of up to Scap rows (independent of where the KV cache lives)
void ensure_prefill_ws(vc_model* m, int B, int Scap) {
    const vc_model_cfg& c = m->c;
    const int D = c.hidden, F = c.ffn;
    const size_t Mrows = (size_t)B * Scap;
    m->x.ensure(Mrows * D * 4);
    m->xn.ensure(Mrows * (D + XN_PAD) * 2);
    m->qkv.ensure(Mrows * 3 * D * 2);
    m->q.ensure(Mrows * D * 2, true);
    m->attn.ensure(Mrows * D * 2);
    m->h.ensure(Mrows * F * 2);
    if (m->weight_format == 2) {
        m->a8.ensure(Mrows * std::max(D, F));
        m->a8_scale.ensure(Mrows * 4);
    }
    m->row_src.ensure(Mrows * 8);
    m->p_ssq.ensure(Mrows * (size_t)m->npart * 4);   // folded RMSNorm of the prefill: sum-of-squares partials + row scales
    m->p_rstd.ensure(Mrows * 4);
    const int Bp = (int)rup(B, 16);
    m->last_idx.ensure(Bp * 4);
    m->xl.ensure((size_t)Bp * D * 2, true);
    m->logits.ensure((size_t)Bp * c.vocab * 4, true);
}

void drop_graph(vc_model* m) {
    if (m->graph) { (void)hipGraphExecDestroy(m->graph); m->graph = nullptr; }
}

// the session's own decode loop: KV cache for B sequences of S_total positions + the per-step buffers
void ensure_llm(vc_model* m, int B, int S_total) {
    const vc_model_cfg& c = m->c;
    const int D = c.hidden, F = c.ffn, H = c.heads;
    const int Scap = (int)rup(S_total, 64);
    REQUIRE(B <= VC_MAX_ROWS, VC_ERR_INVALID, "batch %d: at most %d sequences per GPU replica (shard larger batches over ranks)",
            B, VC_MAX_ROWS);
    REQUIRE(Scap <= c.max_positions, VC_ERR_INVALID, "sequence %d exceeds max_position_embeddings=%d", Scap, c.max_positions);
    // split mode keeps fp24 (or fp32) keys / values; the fp8 weight format e4m3 ones
    const int es = m->precision == 2 ? split_kv_es() : (m->precision == 0 ? step_kv_es(m) : 2);
    if (B != m->capB || Scap > m->capS || es != m->kv_es) {
        const int newS = std::max(Scap, (m->capB == B && es == m->kv_es) ? m->capS : 0);
        const size_t per_layer = (size_t)B * H * newS * m->hd;
        m->kc.release();
        m->vc.release();
        m->kc.ensure(per_layer * c.layers * es, true);
        m->vc.ensure(per_layer * c.layers * es, true);
        m->capB = B;
        m->capS = newS;
        m->kv_es = es;
        drop_graph(m);
    }
    const void* before[] = {m->x_dec.p, m->xg_dec.p, m->qkv_dec.p, m->attn_dec.p, m->h_dec.p, m->next_tok.p, m->rows.p, m->ssq.p,
                            m->logits.p};
    ensure_prefill_ws(m, B, Scap);
    const int Bp = (int)rup(B, 16);
    // (sized for the split step as well: hi + lo row groups double the bf16 operands, qkv_dec is fp32 there)
    m->x_dec.ensure((size_t)Bp * D * 4, true);
    m->xg_dec.ensure((size_t)2 * Bp * D * 2, true);
    m->qkv_dec.ensure((size_t)Bp * 3 * D * 4, true);
    m->attn_dec.ensure((size_t)2 * Bp * D * 2, true);
    m->h_dec.ensure((size_t)2 * Bp * F * 2, true);
    m->next_tok.ensure(Bp * 4, true);
    m->rows.ensure((size_t)Bp * RS_STRIDE * 4, true);
    m->ssq.ensure((size_t)Bp * m->npart * 4, true);
    m->sk_scratch.ensure(sk_floats(c) * 4);   // [K-slices][tiles][2 row groups][256]
    m->sk_counters.ensure((size_t)sk_counters_n(c) * 4, true);
    const void* after[] = {m->x_dec.p, m->xg_dec.p, m->qkv_dec.p, m->attn_dec.p, m->h_dec.p, m->next_tok.p, m->rows.p, m->ssq.p,
                           m->logits.p};
    for (size_t i = 0; i < sizeof(before) / sizeof(before[0]); ++i)
        if (before[i] != after[i]) drop_graph(m);  // the decode graph bakes these pointers in
}

LoopView session_view(vc_model* m) {
    LoopView v{};
    v.st = m->st;
    v.kc = m->kc.as<bf16_t>();
    v.vc = m->vc.as<bf16_t>();
    v.es = m->kv_es;
    v.split_G = m->precision == 2 ? (m->capB <= 8 ? 8 : 16) : 0;
    v.capR = m->capB;
    v.capS = m->capS;
    v.rows = m->rows.as<int>();
    v.kmask = m->kmask_in_decode ? m->kmask.as<uint8_t>() : nullptr;
    v.kmask_stride = m->c.max_positions;
    v.x_dec = m->x_dec.as<float>();
    v.xg_dec = m->xg_dec.as<bf16_t>();
    v.qkv_dec = m->qkv_dec.as<bf16_t>();
    v.attn_dec = m->attn_dec.as<bf16_t>();
    v.h_dec = m->h_dec.as<bf16_t>();
    v.logits = m->logits.as<float>();
    v.next_tok = m->next_tok.as<int>();
    v.out_ids = m->out_ids.as<int>();
    v.ssq = m->ssq.as<float>();
    v.sk_scratch = m->sk_scratch.as<float>();
    v.sk_counters = m->sk_counters.as<unsigned>();
    return v;
}

// where a prefill writes its keys / values: rows [row0, row0 + B) of a cache with capR rows of capS positions
struct KvTarget {
    bf16_t *kc, *vc;
    int capR, capS, row0;
    int es;  // bytes per element: 2 (bf16); precision mode "split": 3 (fp24) or 4 (fp32)
};
// layer l of a cache whose elements are `es` bytes (the pointer type is nominal for es == 4)
bf16_t* kv_layer(bf16_t* base, int es, size_t elems) { return reinterpret_cast<bf16_t*>(reinterpret_cast<char*>(base) + elems * es); }
bf16_t* kcache(const vc_model* m, const KvTarget& t, int l) {
    return kv_layer(t.kc, t.es, ((size_t)l * t.capR + t.row0) * m->c.heads * t.capS * m->hd);
}
bf16_t* vcache(const vc_model* m, const KvTarget& t, int l) {
    return kv_layer(t.vc, t.es, ((size_t)l * t.capR + t.row0) * m->c.heads * t.capS * m->hd);
}
KvTarget session_kv(vc_model* m) { return KvTarget{m->kc.as<bf16_t>(), m->vc.as<bf16_t>(), m->capB, m->capS, 0, m->kv_es}; }
bf16_t* kcache(const LoopView& v, const vc_model* m, int l) {
    return kv_layer(v.kc, v.es, (size_t)l * v.capR * m->c.heads * v.capS * m->hd);
}
bf16_t* vcache(const LoopView& v, const vc_model* m, int l) {
    return kv_layer(v.vc, v.es, (size_t)l * v.capR * m->c.heads * v.capS * m->hd);
}

// The decode loop outran the cache (a host-driven vc_decode_step loop past the reserve of its prefill): re-allocate with
// room for `need` positions and move the live prefix — K and V rows are contiguous per (layer, sample, head), so each is
// one strided 2-D copy.  The decode graph bakes the cache pointers in and is re-captured.
void grow_kv(vc_model* m, int need) {
    const vc_model_cfg& c = m->c;
    const int oldS = m->capS, live = m->cur_pos;
    int newS = (int)rup(std::max(need, std::min(2 * oldS, c.max_positions)), 64);
    newS = std::min(newS, c.max_positions / 64 * 64);
    REQUIRE(newS >= need, VC_ERR_STATE, "KV cache full: position %d exceeds max_position_embeddings=%d", need, c.max_positions);
    const size_t heads = (size_t)c.layers * m->capB * c.heads;
    Buf nk, nv;
    const size_t es = (size_t)m->kv_es;
    nk.ensure(heads * newS * m->hd * es, true);
    nv.ensure(heads * newS * m->hd * es, true);
    HIPCHK(hipMemcpy2DAsync(nk.p, (size_t)newS * m->hd * es, m->kc.p, (size_t)oldS * m->hd * es, (size_t)live * m->hd * es, heads,
                            hipMemcpyDeviceToDevice, m->st));
    HIPCHK(hipMemcpy2DAsync(nv.p, (size_t)newS * m->hd * es, m->vc.p, (size_t)oldS * m->hd * es, (size_t)live * m->hd * es, heads,
                            hipMemcpyDeviceToDevice, m->st));
    HIPCHK(hipStreamSynchronize(m->st));
    m->kc.release();
    m->vc.release();
    m->kc = nk;
    m->vc = nv;
    if (m->precision == 1 && m->s_capS == oldS && m->s_capB == m->capB) {  // the strict path's fp32 caches (K and V key-major)
        Buf sk, sv;
        sk.ensure(heads * newS * m->hd * 4, true);
        sv.ensure(heads * newS * m->hd * 4, true);
        HIPCHK(hipMemcpy2DAsync(sk.p, (size_t)newS * m->hd * 4, m->s_kc.p, (size_t)oldS * m->hd * 4, (size_t)live * m->hd * 4,
                                heads, hipMemcpyDeviceToDevice, m->st));
        HIPCHK(hipMemcpy2DAsync(sv.p, (size_t)newS * m->hd * 4, m->s_vc.p, (size_t)oldS * m->hd * 4, (size_t)live * m->hd * 4,
                                heads, hipMemcpyDeviceToDevice, m->st));
        HIPCHK(hipStreamSynchronize(m->st));
        m->s_kc.release();
        m->s_vc.release();
        m->s_kc = sk;
        m->s_vc = sv;
        m->s_capS = newS;
    }
    m->capS = newS;
    drop_graph(m);
}

// output_hidden_states ([HF] LlamaModel.forward: the tuple (inputs_embeds, layer 1 output, ..., layer L-1 output,
// norm(layer L output))): entry `idx` of the caller's host buffer <- the fp32 residual stream (idx == layers: after the final
// RMSNorm).  No-op unless requested for this prefill.
void emit_hidden(vc_model* m, int idx, int B, int S, const float* x_src) {
    if (!m->hidden_out) return;
    const vc_model_cfg& c = m->c;
    const size_t n = (size_t)B * S * c.hidden;
    REQUIRE((size_t)(idx + 1) * n <= m->hidden_cap, VC_ERR_INVALID, "hidden-state buffer too small: %zu floats for entry %d of %zu",
            m->hidden_cap, idx, n);
    const float* src = x_src ? x_src : m->x.as<float>();
    if (idx == c.layers) {
        m->hidden_tmp.ensure(n * 4);
        launch_rmsnorm_f32(src, nullptr, m->final_norm, m->hidden_tmp.as<float>(), B * S, c.hidden, c.rms_eps, m->st);
        src = m->hidden_tmp.as<float>();
    }
    HIPCHK(hipMemcpyAsync(m->hidden_out + (size_t)idx * n, src, n * 4, hipMemcpyDeviceToHost, m->st));
}

// decoder layers [l0, l1) of a prefill (l1 < 0: all, or the first layer_limit)
void run_prefill_layers(vc_model* m, const KvTarget& kv, int B, int S, int l0 = 0, int l1 = -1) {
    const vc_model_cfg& c = m->c;
    const int D = c.hidden, F = c.ffn, H = c.heads, M = B * S;
    const int nl = l1 >= 0 ? l1 : (m->layer_limit > 0 ? std::min(m->layer_limit, c.layers) : c.layers);
    const int Sr = (int)rup(S, 64);
    m->vt_pre.ensure((size_t)B * H * m->hd * Sr * 2, true);
    if (kv.es == 1) m->k_pre.ensure((size_t)B * H * Sr * m->hd * 2, true);
    const bool f8 = m->weight_format == 2;
    // VC_PREFILL_FOLD=1 (opt-in; read per call so that tests can switch it): RMSNorm never runs as a pass behind the first
    // layer — the GEMM that writes a residual row (o_proj, down) also writes xg = bf16(x * g) for the next GEMM and the row's
    // sum-of-squares partials, a one-wave-per-row launch turns them into 1/rms, and the consuming GEMM (gate/up, the next
    // layer's QKV) scales its accumulator by it: the decode steps' form.  Measured on MI355X (7b, B = 8,
    // profiles/r04_d_*): 112.9 ms per prefill against 111.6 with the 63 passes — the residual GEMMs' epilogue (8-byte bf16
    // stores in 32-byte segments, partials) costs more than the 45-us passes it removes — so the passes stay the default.
    const bool fold = prefill_fold_on() && !f8;
    float* rstd = m->p_rstd.as<float>();
    NormFold prod{nullptr, m->xn.as<bf16_t>(), nullptr, m->p_ssq.as<float>(), D + XN_PAD, 0, m->npart};
    const NormFold cons{rstd};
    bool have_xg = false;   // xn holds bf16(x * g) of the CURRENT x for the norm about to be consumed, rstd its row scales
    // the fused QKV epilogue: hd 128, two heads per 256-row weight tile, a problem the 256 x 256 kernel serves anyway
    const int qf = root_of(m)->qkv_fused;
    const bool qkv_fused = qf && m->hd == 128 && D % 256 == 0 && (qf > 1 || (long)B * rup(S, 32) >= 1024);
    for (int l = l0; l < nl; ++l) {
        const LlmLayer& L = m->llm[l];
        // K and V rows go to the cache (key-major: what the decode steps stream); the V^T tiles of this layer's flash
        // attention live in a per-call scratch [B,H,hd,Sr]
        // (kv.es == 1, the e4m3 cache of the fp8 format: the flash kernel of THIS prefill reads bf16 K rows from a per-call
        // scratch, the cache receives e4m3 rows)
        const bool kv8 = kv.es == 1;
        bf16_t* kflash = kv8 ? m->k_pre.as<bf16_t>() : kcache(m, kv, l);
        const int kflash_stride = kv8 ? Sr : kv.capS;
        // round 6: RoPE + head split + KV write in the QKV GEMM's epilogue — the fused [M, 3D] rows are never
        // written and qkv_split_kernel's pass (105 us per 7b layer at B = 8, 263 us at 13b B = 16) is gone.  The token rows of that
        // GEMM are the samples padded to a multiple of 32 (EPI_QKV, kernels.h); same bits as the two launches it replaces.
        const int Sp = (int)rup(S, 32);
        if (qkv_fused) {
            QkvEpiArgs qe{m->q.as<bf16_t>(), kflash, kv8 ? nullptr : vcache(m, kv, l), m->vt_pre.as<bf16_t>(),
                          kv8 ? reinterpret_cast<uint8_t*>(kcache(m, kv, l)) : nullptr,
                          kv8 ? reinterpret_cast<uint8_t*>(vcache(m, kv, l)) : nullptr, m->rope_cos, m->rope_sin, B, S, Sp, H, S,
                          kflash_stride, Sr, kv.capS};
            if (f8) {
                launch_rmsnorm_q8(m->x.as<float>(), L.in_norm, m->a8.as<uint8_t>(), m->a8_scale.as<float>(), M, D, c.rms_eps, m->st);
                gemm_f8(m, nullptr, L.qkv_q, L.qkv_s, nullptr, B * Sp, 3 * D, D, 0, EPI_QKV, &qe);
            } else {
                if (!have_xg) launch_rmsnorm(m->x.as<float>(), L.in_norm, m->xn.as<bf16_t>(), M, D, c.rms_eps, m->st, D + XN_PAD);
                gemm(m, m->xn.as<bf16_t>(), L.qkv_w, nullptr, nullptr, B * Sp, 3 * D, D, 0, EPI_QKV, D + XN_PAD, have_xg ? &cons : nullptr, &qe);
            }
        } else {
            if (f8) {  // RMSNorm writes the e4m3 operand of the QKV GEMM directly
                launch_rmsnorm_q8(m->x.as<float>(), L.in_norm, m->a8.as<uint8_t>(), m->a8_scale.as<float>(), M, D, c.rms_eps, m->st);
                gemm_f8(m, nullptr, L.qkv_q, L.qkv_s, m->qkv.p, M, 3 * D, D, 3 * D, EPI_BF16);
            } else {
                if (!have_xg) launch_rmsnorm(m->x.as<float>(), L.in_norm, m->xn.as<bf16_t>(), M, D, c.rms_eps, m->st, D + XN_PAD);
                gemm(m, m->xn.as<bf16_t>(), L.qkv_w, nullptr, m->qkv.p, M, 3 * D, D, 3 * D, EPI_BF16, D + XN_PAD, have_xg ? &cons : nullptr);
            }
            QkvSplitArgs qa{m->qkv.as<bf16_t>(), m->q.as<bf16_t>(), kflash, m->vt_pre.as<bf16_t>(), B, S, H, m->hd, S, kflash_stride,
                            nullptr, m->rope_cos, m->rope_sin, kv8 ? nullptr : vcache(m, kv, l), Sr,
                            kv8 ? reinterpret_cast<uint8_t*>(kcache(m, kv, l)) : nullptr,
                            kv8 ? reinterpret_cast<uint8_t*>(vcache(m, kv, l)) : nullptr, kv.capS};
            launch_qkv_split(qa, m->st);
        }
        AttnArgs aa{m->q.as<bf16_t>(), kflash, m->vt_pre.as<bf16_t>(), m->attn.as<bf16_t>(), B, H, S, m->hd, S, kflash_stride, 1,
                    1.0f / sqrtf((float)m->hd), Sr};
        if (m->has_kmask) {
            aa.key_mask = m->kmask.as<uint8_t>();
            aa.mask_stride = c.max_positions;
        }
        launch_attention(aa, m->st);
        if (m->attn_out) {
            AttnProbsArgs pa{};
            pa.q_hi = m->q.as<bf16_t>();
            pa.k_hi = kflash;
            pa.q_stride = S;
            pa.kv_stride = kflash_stride;
            emit_attentions(m, l, B, S, pa);
        }
        if (f8) {
            gemm_f8(m, m->attn.as<bf16_t>(), L.o_q, L.o_s, m->x.p, M, D, D, D, EPI_RESID_F32);
            launch_rmsnorm_q8(m->x.as<float>(), L.post_norm, m->a8.as<uint8_t>(), m->a8_scale.as<float>(), M, D, c.rms_eps, m->st);
            gemm_f8(m, nullptr, L.gu_q, L.gu_s, m->h.p, M, 2 * F, D, F, EPI_SWIGLU);
            gemm_f8(m, m->h.as<bf16_t>(), L.down_q, L.down_s, m->x.p, M, D, F, D, EPI_RESID_F32);
        } else if (fold) {
            prod.xg_w = L.post_norm;
            gemm(m, m->attn.as<bf16_t>(), L.o_w, nullptr, m->x.p, M, D, D, D, EPI_RESID_F32, 0, &prod);
            launch_rstd_from_partials(prod.ssq_out, m->npart, D / 16, rstd, M, D, c.rms_eps, m->st);
            gemm(m, m->xn.as<bf16_t>(), L.gu_w, nullptr, m->h.p, M, 2 * F, D, F, EPI_SWIGLU, D + XN_PAD, &cons);
            have_xg = l + 1 < nl;   // the last layer's output meets the final norm on its gathered rows only
            prod.xg_w = have_xg ? m->llm[l + 1].in_norm : nullptr;
            gemm(m, m->h.as<bf16_t>(), L.down_w, nullptr, m->x.p, M, D, F, D, EPI_RESID_F32, 0, have_xg ? &prod : nullptr);
            if (have_xg) launch_rstd_from_partials(prod.ssq_out, m->npart, D / 16, rstd, M, D, c.rms_eps, m->st);
        } else {
            gemm(m, m->attn.as<bf16_t>(), L.o_w, nullptr, m->x.p, M, D, D, D, EPI_RESID_F32);
            launch_rmsnorm(m->x.as<float>(), L.post_norm, m->xn.as<bf16_t>(), M, D, c.rms_eps, m->st, D + XN_PAD);
            gemm(m, m->xn.as<bf16_t>(), L.gu_w, nullptr, m->h.p, M, 2 * F, D, F, EPI_SWIGLU, D + XN_PAD);
            gemm(m, m->h.as<bf16_t>(), L.down_w, nullptr, m->x.p, M, D, F, D, EPI_RESID_F32);
        }
        emit_hidden(m, l + 1, B, S);
    }
}

// the decoder stack of a prefill in precision mode "split": fp32 keys / values go to `kv` (es == 4), the bf16 hi / lo
// planes the flash kernel needs live in per-call scratch
void run_prefill_layers_split(vc_model* m, const KvTarget& kv, int B, int S, int l0 = 0, int l1 = -1) {
    const vc_model_cfg& c = m->c;
    const int D = c.hidden, F = c.ffn, H = c.heads, M = B * S;
    const int nl = l1 >= 0 ? l1 : (m->layer_limit > 0 ? std::min(m->layer_limit, c.layers) : c.layers);
    const int Sr = (int)rup(S, 64);
    const int ldx = split_ld(D), ldh = split_ld(F);
    REQUIRE(kv.es == 3 || kv.es == 4, VC_ERR_STATE, "split mode needs an fp24 / fp32 KV cache");
    const size_t qplane = (size_t)B * H * S * m->hd, kplane = (size_t)B * H * Sr * m->hd;
    m->xn.ensure((size_t)M * ldx * 2);
    m->s_qkv.ensure((size_t)M * 3 * D * 4);
    m->q.ensure(2 * qplane * 2, true);
    m->vt_pre.ensure(4 * kplane * 2, true);   // K hi | K lo | V^T hi | V^T lo
    m->attn.ensure((size_t)M * ldx * 2);
    m->h.ensure((size_t)M * ldh * 2);
    bf16_t *qh = m->q.as<bf16_t>(), *kh = m->vt_pre.as<bf16_t>(), *vh = kh + 2 * kplane;
    // folded RMSNorm as in run_prefill_layers (opt-in): the producer writes both planes of xg ([hi | lo], the lo plane D columns right)
    const bool fold_on = prefill_fold_on();
    float* rstd = m->p_rstd.as<float>();
    NormFold prod{nullptr, m->xn.as<bf16_t>(), nullptr, m->p_ssq.as<float>(), ldx, D, m->npart};
    const NormFold cons{rstd};
    bool have_xg = false;
    for (int l = l0; l < nl; ++l) {
        const LlmLayer& L = m->llm[l];
        if (!have_xg) launch_rmsnorm_split(m->x.as<float>(), nullptr, L.in_norm, m->xn.as<bf16_t>(), M, D, c.rms_eps, ldx, D, m->st);
        gemm_split(m, m->xn.as<bf16_t>(), L.qkv_w, nullptr, m->s_qkv.p, M, 3 * D, D, 3 * D, EPI_F32, ldx, 0, have_xg ? &cons : nullptr);
        QkvSplit32Args qa{m->s_qkv.as<float>(), qh, qh + qplane, kh, kh + kplane, vh, vh + kplane,
                          reinterpret_cast<float*>(kcache(m, kv, l)), reinterpret_cast<float*>(vcache(m, kv, l)),
                          B, S, H, m->hd, S, Sr, Sr, kv.capS, m->rope_cos, m->rope_sin, kv.es == 3};
        launch_qkv_split32(qa, m->st);
        AttnArgs aa{qh, kh, vh, m->attn.as<bf16_t>(), B, H, S, m->hd, S, Sr, 1, 1.0f / sqrtf((float)m->hd), Sr,
                    qh + qplane, kh + kplane, vh + kplane, ldx, D};
        if (m->has_kmask) {
            aa.key_mask = m->kmask.as<uint8_t>();
            aa.mask_stride = c.max_positions;
        }
        launch_attention(aa, m->st);
        if (m->attn_out) {
            AttnProbsArgs pa{};
            pa.q_hi = qh;
            pa.q_lo = qh + qplane;
            pa.k_hi = kh;
            pa.k_lo = kh + kplane;
            pa.q_stride = S;
            pa.kv_stride = Sr;
            emit_attentions(m, l, B, S, pa);
        }
        if (fold_on) {
            prod.xg_w = L.post_norm;
            gemm_split(m, m->attn.as<bf16_t>(), L.o_w, nullptr, m->x.p, M, D, D, D, EPI_RESID_F32, ldx, 0, &prod);
            launch_rstd_from_partials(prod.ssq_out, m->npart, D / 16, rstd, M, D, c.rms_eps, m->st);
            gemm_split(m, m->xn.as<bf16_t>(), L.gu_w, nullptr, m->h.p, M, 2 * F, D, ldh, EPI_SWIGLU, ldx, F, &cons);
            have_xg = l + 1 < nl;
            prod.xg_w = have_xg ? m->llm[l + 1].in_norm : nullptr;
            gemm_split(m, m->h.as<bf16_t>(), L.down_w, nullptr, m->x.p, M, D, F, D, EPI_RESID_F32, ldh, 0, have_xg ? &prod : nullptr);
            if (have_xg) launch_rstd_from_partials(prod.ssq_out, m->npart, D / 16, rstd, M, D, c.rms_eps, m->st);
        } else {
            gemm_split(m, m->attn.as<bf16_t>(), L.o_w, nullptr, m->x.p, M, D, D, D, EPI_RESID_F32, ldx);
            launch_rmsnorm_split(m->x.as<float>(), nullptr, L.post_norm, m->xn.as<bf16_t>(), M, D, c.rms_eps, ldx, D, m->st);
            gemm_split(m, m->xn.as<bf16_t>(), L.gu_w, nullptr, m->h.p, M, 2 * F, D, ldh, EPI_SWIGLU, ldx, F);
            gemm_split(m, m->h.as<bf16_t>(), L.down_w, nullptr, m->x.p, M, D, F, D, EPI_RESID_F32, ldh);
        }
        emit_hidden(m, l + 1, B, S);
    }
}

// ---- one cached decode step over the first `nrows` rows of a loop (captured into a hipGraph): 5 launches per layer + 2.
// x_dec (fp32 residual rows of the new tokens), their sum-of-squares partials and xg are prepared by the previous step's
// select kernel (or by embed_tokens_ssq when the host supplies the tokens).  Positions, step counts and every
// generation parameter are read from the rows' RowState records.
SelectArgs select_args(vc_model* m, const LoopView& v, const float* logits, int nrows, int advance) {
    SelectArgs a{};
    a.logits = logits;
    a.ldl = m->c.vocab;
    a.rows = v.rows;
    a.next_tok = v.next_tok;
    a.out_ids = v.out_ids;
    a.embed = m->embed;
    a.embed_lo = (m->precision || v.split_G) ? lo_plane(m, m->embed) : nullptr;   // strict / split on an inexact checkpoint: x = hi + lo
    a.x = v.x_dec;
    a.ssq = v.ssq;
    a.xg_w = m->llm[0].in_norm;
    a.xg = v.xg_dec;
    a.D = m->c.hidden;
    a.npart = m->npart;
    a.V = m->c.vocab;
    a.nrows = nrows;
    a.advance = advance;
    a.xg_G = v.split_G;
    return a;
}

void enqueue_decode_step(vc_model* m, const LoopView& v, int nrows) {
    decode_linears(m, v, nrows, [&](int l) {
        AttnDecodeFusedArgs da{v.qkv_dec, kcache(v, m, l), vcache(v, m, l), v.attn_dec, nrows, m->c.heads, m->hd, v.capS,
                               v.rows + RS_POS, m->rope_cos, m->rope_sin, 1.0f / sqrtf((float)m->hd), RS_STRIDE,
                               v.rows + RS_ACTIVE, v.split_G ? (v.es == 3 ? 2 : 1) : (v.es == 1 ? 3 : 0), v.split_G, v.kmask, v.kmask_stride};
        da.stamp = next_stamp(v);
        launch_attention_decode_fused(da, v.st);
    });
    launch_select_embed(select_args(m, v, v.logits, nrows, 3), v.st);                                        // K19/K20+K10
    // in-situ timing: fold the step's slots (5 per layer: qkv, attention, o, gate/up, down; then lm_head) into the span's sums
    if (v.stamps && v.stamp_next && v.prof_acc)
        launch_stamp_accumulate(v.stamps, *v.stamp_next, m->c.layers, v.prof_acc, v.stamp_scratch, v.st);
}

// The same step run eagerly with the output_hidden_states / output_attentions hooks of a cached decode step
// (vc_request_hidden_states / vc_request_attentions before vc_decode_step): inputs_embeds row, every layer's residual row, the
// final norm; and per layer the probabilities of the new token's query over the pos + 1 keys, recomputed from the step's own
// roped q (rounded as the fused kernel rounds it) and the K cache — the fused decode attention keeps only unnormalised scores.
void enqueue_decode_step_diag(vc_model* m, const LoopView& v, int nrows, int pos) {
    const vc_model_cfg& c = m->c;
    emit_hidden(m, 0, nrows, 1, v.x_dec);
    decode_linears(
        m, v, nrows,
        [&](int l) {
            AttnDecodeFusedArgs da{v.qkv_dec, kcache(v, m, l), vcache(v, m, l), v.attn_dec, nrows, c.heads, m->hd, v.capS,
                                   v.rows + RS_POS, m->rope_cos, m->rope_sin, 1.0f / sqrtf((float)m->hd), RS_STRIDE,
                                   v.rows + RS_ACTIVE, v.split_G ? (v.es == 3 ? 2 : 1) : (v.es == 1 ? 3 : 0), v.split_G, v.kmask, v.kmask_stride};
            launch_attention_decode_fused(da, v.st);
            if (m->attn_out) {
                m->attn_q.ensure((size_t)nrows * c.hidden * 4);
                launch_rope_q_decode(v.qkv_dec, v.split_G != 0, m->attn_q.as<float>(), nrows, c.heads, m->hd, pos, m->rope_cos, m->rope_sin,
                                     v.split_G == 0, v.st);
                AttnProbsArgs pa{};
                pa.q32 = m->attn_q.as<float>();
                if (v.split_G && v.es == 3) pa.k24 = kcache(v, m, l);
                else if (v.split_G) pa.k32 = reinterpret_cast<const float*>(kcache(v, m, l));
                else if (v.es == 1) pa.k8 = reinterpret_cast<const uint8_t*>(kcache(v, m, l));   // the fp8 weight format's e4m3 rows
                else pa.k_hi = kcache(v, m, l);
                pa.q_stride = 1;
                pa.kv_stride = v.capS;
                emit_attentions(m, l, nrows, 1, pa, pos + 1, pos);
            }
        },
        [&](int l) { emit_hidden(m, l + 1, nrows, 1, v.x_dec); });
    launch_select_embed(select_args(m, v, v.logits, nrows, 3), v.st);
}

// strict mode: the rows of a session advance in lockstep, so row 0's position serves every row of the fp32 kernels
void enqueue_decode_step_strict(vc_model* m, int B) {
    const LoopView v = session_view(m);
    emit_hidden(m, 0, B, 1, m->x_dec.as<float>());
    run_llm_layers_strict(m, m->x_dec.as<float>(), B, 1, v.rows + RS_POS, false);
    logits_strict(m, m->x_dec.as<float>(), nullptr, B);
    launch_select_embed(select_args(m, v, v.logits, B, 3), v.st);
}

hipGraphExec_t capture_step(vc_model* m, const LoopView& v, int nrows) {
    hipGraph_t g = nullptr;
    hipGraphExec_t exec = nullptr;
    // thread-local mode: other sessions (host threads) may allocate / copy while this thread captures
    HIPCHK(hipStreamBeginCapture(v.st, hipStreamCaptureModeThreadLocal));
    try {
        enqueue_decode_step(m, v, nrows);
    } catch (...) {
        (void)hipStreamEndCapture(v.st, &g);
        throw;
    }
    HIPCHK(hipStreamEndCapture(v.st, &g));
    HIPCHK(hipGraphInstantiate(&exec, g, nullptr, nullptr, 0));
    HIPCHK(hipGraphDestroy(g));
    return exec;
}

void ensure_graph(vc_model* m, int B) {
    if (m->graph && m->graph_rows == B && m->graph_masked == m->kmask_in_decode) return;
    drop_graph(m);
    m->graph = capture_step(m, session_view(m), B);
    m->graph_rows = B;
    m->graph_masked = m->kmask_in_decode;
}

void ensure_out_ids(vc_model* m, int B, int max_new) {
    const int stride = std::max(max_new, 1);
    if (stride > m->out_stride || m->out_ids.cap < (size_t)rup(B, 16) * stride * 4) {
        m->out_stride = std::max(stride, m->out_stride);
        m->out_ids.ensure((size_t)rup(B, 16) * m->out_stride * 4);
        drop_graph(m);  // pointer baked into the graph
    }
}

// ---- host-side RowState records ---------------------------------------------------------------------------------
struct GenParams {  // what a generate() call asks for (HF GenerationMixin subset)
    int max_new = 0, eos = -1, pad = 0;
    int do_sample = 0, top_k = 0;
    float temperature = 1.f, top_p = 1.f;
    uint64_t seed = 0;
    int n_stop = 0;
    int stop[VC_MAX_STOP][1 + VC_MAX_STOP_LEN] = {};
};

uint32_t mix_seed(uint64_t seed, uint32_t row, uint32_t salt) {
    uint64_t z = seed + 0x9E3779B97F4A7C15ull * (uint64_t)(row + 1) + salt;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return (uint32_t)(z ^ (z >> 31));
}

// records of the B rows of one request (row b of the request = record b); `tail` = last prompt ids per row, right-aligned
void fill_rows(int* rec, int B, const GenParams& g, int pos, const int* tail /*[B][VC_MAX_STOP_LEN-1] or null*/, int out_off0,
               int out_stride) {
    memset(rec, 0, (size_t)B * RS_STRIDE * sizeof(int));
    const float inv_t = 1.0f / g.temperature;
    for (int b = 0; b < B; ++b) {
        int* r = rec + (size_t)b * RS_STRIDE;
        r[RS_ACTIVE] = 1;
        r[RS_POS] = pos;
        r[RS_MAXNEW] = g.max_new;
        r[RS_EOS] = g.eos;
        r[RS_PAD] = g.pad;
        r[RS_NSTOP] = g.n_stop;
        r[RS_SAMPLE] = g.do_sample;
        memcpy(&r[RS_INVTEMP], &inv_t, 4);
        r[RS_TOPK] = g.top_k;
        memcpy(&r[RS_TOPP], &g.top_p, 4);
        r[RS_SEED_LO] = (int)mix_seed(g.seed, (uint32_t)b, 0x51u);  // per-row streams: independent of the row's slot
        r[RS_SEED_HI] = (int)mix_seed(g.seed, (uint32_t)b, 0xA7u);
        r[RS_OUT_OFF] = out_off0 + b * out_stride;
        for (int j = 0; j < VC_MAX_STOP_LEN - 1; ++j) r[RS_TAIL + j] = tail ? tail[b * (VC_MAX_STOP_LEN - 1) + j] : INT32_MIN;
        for (int q = 0; q < g.n_stop; ++q)
            for (int j = 0; j < 1 + VC_MAX_STOP_LEN; ++j) r[RS_STOP + q * (1 + VC_MAX_STOP_LEN) + j] = g.stop[q][j];
    }
}

// prefill through the last-row logits; leaves logits [B,V] on device
// encode + splice: inputs_embeds of the batch in m->x.  `own_kv`: size the session's own KV cache / decode loop for the
// spliced length plus reserve_new positions (reserve_new < 0: a hint, see below); otherwise only the prefill workspaces
// (the keys go to a pool's cache and the caller checks the capacity).
void do_prefill(vc_model* m, const int64_t* ids, int B, int T, const float* img, const float* seg, const float* depth,
                int on_dev, int has_mask, int reserve_new, bool own_kv, int* S_out) {
    const vc_model_cfg& c = m->c;
    REQUIRE(m->finalized, VC_ERR_STATE, "vc_model_finalize() has not been called");
    REQUIRE(B >= 1 && T >= 1 && ids, VC_ERR_INVALID, "bad ids/B/T");
    REQUIRE(B <= VC_MAX_ROWS, VC_ERR_INVALID, "batch %d: at most %d sequences per prefill (larger batches run in pieces)", B,
            VC_MAX_ROWS);
    // images == NULL: the reference's prepare_inputs_labels_for_multimodal returns early (vcoder_ds_llava_arch.py:129-133) and
    // the call is a plain LlamaForCausalLM forward over the text ids (every id must be a vocabulary id)
    const bool text_only = img == nullptr;
    if (text_only || c.variant == VC_VARIANT_LLAVA) seg = depth = nullptr;
    if (c.variant != VC_VARIANT_VCODER_DS) depth = nullptr;
    PixSet pix{{img, seg, depth}, {0, 0, 0}};
    for (int k = 0; k < 3; ++k) {
        m->img_first[k].clear();
        if (!pix.p[k]) continue;
        if (m->img_counts[k].empty()) {
            pix.n[k] = B;
        } else {
            REQUIRE((int)m->img_counts[k].size() == B, VC_ERR_INVALID, "image counts given for %zu samples, batch is %d",
                    m->img_counts[k].size(), B);
            m->img_first[k].push_back(0);
            for (int b = 0; b < B; ++b) m->img_first[k].push_back(m->img_first[k].back() + m->img_counts[k][b]);
            pix.n[k] = m->img_first[k].back();
        }
    }
    for (auto& v : m->img_counts) v.clear();  // one-shot
    if (m->ev[0]) HIPCHK(hipEventRecord(m->ev[0], m->st));
    if (text_only) {
        for (int k = 0; k < 3; ++k) m->feat_rows[k] = 0;
        if (m->precision) m->s_feats.ensure(256);   // the fp32 splice takes a feature base pointer (no row refers to it)
        else m->feats.ensure(256);
    } else if (m->plan_only) {
        // the plan needs the feature-row COUNTS of every modality and the depth pixels (is_depth_zero), not the features
        const int Rp = m->Tv - (c.vit_keep_cls ? 0 : 1);
        const size_t img_elems = (size_t)3 * c.vit_image * c.vit_image;
        int first = 0;
        for (int k = 0; k < 3; ++k) {
            m->feat_off[k] = first * Rp;
            m->feat_rows[k] = pix.p[k] ? pix.n[k] * Rp : 0;
            if (pix.p[k]) first += pix.n[k];
        }
        REQUIRE(first > 0, VC_ERR_INVALID, "no images");
        m->v_pixels.ensure((size_t)first * img_elems * 4);
        if (depth)
            HIPCHK(hipMemcpyAsync(m->v_pixels.as<float>() + (size_t)(first - pix.n[VC_MOD_DEPTH]) * img_elems, depth,
                                  (size_t)pix.n[VC_MOD_DEPTH] * img_elems * 4, on_dev ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice,
                                  m->st));
    } else if (m->precision == 1) run_vit_and_adapters_strict(m, pix, on_dev);
    else if (m->precision == 2) run_vit_and_adapters_split(m, pix, on_dev);
    else run_vit_and_adapters(m, pix, on_dev);
    const int R = m->Tv - (c.vit_keep_cls ? 0 : 1);
    std::vector<bool> dz;
    if (depth) {  // is_depth_zero = [mean(d) == 0 for d in depth_images]  (vcoder_ds_llava_arch.py:161) — one host sync, as the reference
        const int nd = pix.n[VC_MOD_DEPTH];
        m->dsum.ensure((size_t)rup(nd, 16) * ROW_SUM_PARTS * 4);
        int di = 0;
        for (int k = 0; k < 2; ++k) di += pix.n[k];
        const size_t img_elems = (size_t)3 * c.vit_image * c.vit_image;
        launch_row_sum(m->v_pixels.as<float>() + (size_t)di * img_elems, img_elems, nd, m->dsum.as<float>(), m->st);
        std::vector<float> hp((size_t)nd * ROW_SUM_PARTS), hs(nd, 0.f);
        HIPCHK(hipMemcpyAsync(hp.data(), m->dsum.p, hp.size() * 4, hipMemcpyDeviceToHost, m->st));
        HIPCHK(hipStreamSynchronize(m->st));
        for (int i = 0; i < nd; ++i)
            for (int q = 0; q < ROW_SUM_PARTS; ++q) hs[i] += hp[(size_t)i * ROW_SUM_PARTS + q];
        const std::vector<int>& first = m->img_first[VC_MOD_DEPTH];
        for (int b = 0; b < B; ++b) {
            const int i0 = first.empty() ? b : first[b], i1 = first.empty() ? b + 1 : first[b + 1];
            float tot = 0.f;
            for (int i = i0; i < i1; ++i) tot += hs[i];
            dz.push_back(i1 > i0 && tot / ((float)(i1 - i0) * (float)img_elems) == 0.0f);
        }
    }
    if (m->ev[1]) HIPCHK(hipEventRecord(m->ev[1], m->st));
    std::vector<std::vector<RowSrc>> rows;
    if (text_only) {
        rows.assign(B, {});
        for (int b = 0; b < B; ++b)
            for (int t = 0; t < T; ++t) {
                const int64_t id = ids[(size_t)b * T + t];
                REQUIRE(id >= 0 && id < c.vocab, VC_ERR_INDEX, "index out of range in self (id %lld reached the embedding lookup)",
                        (long long)id);
                rows[b].push_back({0, (int)id});
            }
    } else {
        plan_rows(m, ids, B, T, seg != nullptr, depth ? &dz : nullptr, R, rows);
    }
    size_t S = 0;
    bool unequal = false;
    for (auto& r : rows) {
        S = std::max(S, r.size());
        unequal |= r.size() != rows[0].size();
    }
    // quirk 6: unequal spliced lengths with an attention_mask and no labels die at vcoder_ds_llava_arch.py:295-297
    REQUIRE(!(unequal && has_mask), VC_ERR_UNEQUAL, "local variable '_new_labels' referenced before assignment");
    REQUIRE(S >= 1, VC_ERR_INVALID, "empty sequence");
    if (m->plan_only) {
        if (S_out) *S_out = (int)S;
        return;
    }
    // The caller's attention_mask [B, T] is LEFT-extended with "visible" over the S - T rows the splice added — by position,
    // whatever the rows hold (vcoder_ds_llava_arch.py:305-311) — and hides its zero positions as KEYS from every query of the
    // sequence in this prefill.
    std::vector<uint8_t> kmask_host;
    m->has_kmask = false;
    if (!m->mask_next.empty()) {
        std::vector<uint8_t> mk;
        mk.swap(m->mask_next);  // one-shot
        REQUIRE(m->mask_B == B && m->mask_T == T, VC_ERR_INVALID, "attention_mask is [%d, %d], input_ids [%d, %d]", m->mask_B,
                m->mask_T, B, T);
        REQUIRE((int)S >= T, VC_ERR_INVALID, "spliced length %zu shorter than the prompt %d", S, T);
        const int stride = c.max_positions, lead = (int)S - T;
        kmask_host.assign((size_t)VC_MAX_ROWS * stride, 1);
        bool any = false;
        for (int b = 0; b < B; ++b)
            for (int t = 0; t < T; ++t)
                if (!mk[(size_t)b * T + t]) {
                    kmask_host[(size_t)b * stride + lead + t] = 0;
                    any = true;
                }
        for (int b = 0; b < B && any; ++b)
            REQUIRE(kmask_host[(size_t)b * stride] != 0, VC_ERR_INVALID,
                    "attention_mask hides position 0 of sequence %d: its first queries would attend to nothing", b);
        m->has_kmask = any;
    }
    {   // reserve_new < 0: a hint (vc_prefill): as many of -reserve_new decode slots as max_position_embeddings allows;
        // reserve_new >= 0: required (generate) — exceeding max_position_embeddings is an error
        int want = (int)S + std::max(reserve_new < 0 ? -reserve_new : reserve_new, 1);
        if (reserve_new < 0) want = std::max((int)S + 1, std::min(want, c.max_positions / 64 * 64));
        if (own_kv) ensure_llm(m, B, want);
        else ensure_prefill_ws(m, B, (int)rup(S, 64));
    }
    std::vector<int> flat((size_t)B * S * 2);
    for (int b = 0; b < B; ++b)
        for (size_t s = 0; s < S; ++s) {
            const RowSrc r = s < rows[b].size() ? rows[b][s] : RowSrc{2, 0};  // zero right-padding (:283)
            flat[((size_t)b * S + s) * 2] = r.kind;
            flat[((size_t)b * S + s) * 2 + 1] = r.src;
        }
    HIPCHK(hipMemcpyAsync(m->row_src.p, flat.data(), flat.size() * 4, hipMemcpyHostToDevice, m->st));
    if (m->has_kmask) {
        m->kmask.ensure(kmask_host.size());
        HIPCHK(hipMemcpyAsync(m->kmask.p, kmask_host.data(), kmask_host.size(), hipMemcpyHostToDevice, m->st));
    }
    m->kmask_in_decode = false;   // a decode mask never outlives its prefill (vc_prefill re-arms it)
    if (m->precision) {  // strict and split: the projected features are fp32
        if (m->precision == 1) {
            REQUIRE(own_kv, VC_ERR_STATE, "strict mode runs on the session's own decode loop");
            ensure_strict(m, B, m->capS);
        }
        launch_splice_f32(m->row_src.as<int>(), (int)(B * S), m->embed, m->s_feats.as<float>(), m->x.as<float>(), c.hidden,
                          m->st, lo_plane(m, m->embed));
    } else {
        launch_splice(m->row_src.as<int>(), (int)(B * S), m->embed, m->feats.as<bf16_t>(), m->x.as<float>(), c.hidden, m->st);
    }
    HIPCHK(hipStreamSynchronize(m->st));  // `flat` is host memory
    m->curB = B;
    m->curS = (int)S;
    if (S_out) *S_out = (int)S;
}

// decoder stack over the spliced batch + last-row logits (m->logits [B,V]); keys / values go to `kv`
void finish_prefill(vc_model* m, const KvTarget& kv, float* logits_all_host) {
    const vc_model_cfg& c = m->c;
    const int B = m->curB, S = m->curS, D = c.hidden;
    std::vector<int> idx(B);
    for (int b = 0; b < B; ++b) idx[b] = b * S + S - 1;
    HIPCHK(hipMemcpyAsync(m->last_idx.p, idx.data(), B * 4, hipMemcpyHostToDevice, m->st));
    emit_hidden(m, 0, B, S);   // inputs_embeds
    if (m->precision == 1) {
        run_llm_layers_strict(m, m->x.as<float>(), B, S, nullptr, true);
        logits_strict(m, m->x.as<float>(), m->last_idx.as<int>(), B);
    } else if (m->precision == 2) {
        run_prefill_layers_split(m, kv, B, S);
        // final norm of the last rows as one stacked hi / lo group of the split GEMV, then lm_head
        const int G = B <= 8 ? 8 : 16;
        m->xl.ensure((size_t)2 * 16 * D * 2, true);
        launch_rmsnorm_split(m->x.as<float>(), m->last_idx.as<int>(), m->final_norm, m->xl.as<bf16_t>(), B, D, c.rms_eps, D,
                             (size_t)G * D, m->st);
        LoopView lv{};
        lv.st = m->st;
        lv.split_G = G;
        gemv(m, lv, m->xl.as<bf16_t>(), m->lm_head_p, nullptr, m->logits.p, B, c.vocab, D, c.vocab, GEMV_F32);
    } else {
        run_prefill_layers(m, kv, B, S);
        launch_rmsnorm_rows(m->x.as<float>(), m->last_idx.as<int>(), m->final_norm, m->xl.as<bf16_t>(), B, D, c.rms_eps, m->st);
        LoopView lv{};  // the lm_head GEMV over the last rows only needs a stream
        lv.st = m->st;
        gemv(m, lv, m->xl.as<bf16_t>(), m->lm_head_p, nullptr, m->logits.p, B, c.vocab, D, c.vocab, GEMV_F32);
    }
    if (logits_all_host) {  // lm_head over ALL S positions, as the reference's forward returns (:93)
        const size_t Mr = (size_t)B * S;
        m->logits_all.ensure(Mr * c.vocab * 4);
        if (m->precision == 1) {
            launch_rmsnorm_f32(m->x.as<float>(), nullptr, m->final_norm, m->s_xn.as<float>(), (int)Mr, D, c.rms_eps, m->st);
            gemm32(m, m->s_xn.as<float>(), m->lm_head, nullptr, m->logits_all.as<float>(), (int)Mr, c.vocab, D, D, D, c.vocab,
                   EPI_F32);
        } else if (m->precision == 2) {
            const int ldx = split_ld(D);
            m->xn.ensure(Mr * ldx * 2);
            launch_rmsnorm_split(m->x.as<float>(), nullptr, m->final_norm, m->xn.as<bf16_t>(), (int)Mr, D, c.rms_eps, ldx, D, m->st);
            gemm_split(m, m->xn.as<bf16_t>(), m->lm_head, nullptr, m->logits_all.p, (int)Mr, c.vocab, D, c.vocab, EPI_F32, ldx);
        } else {
            launch_rmsnorm(m->x.as<float>(), m->final_norm, m->xn.as<bf16_t>(), (int)Mr, D, c.rms_eps, m->st);
            gemm(m, m->xn.as<bf16_t>(), m->lm_head, nullptr, m->logits_all.p, (int)Mr, c.vocab, D, c.vocab, EPI_F32);
        }
        HIPCHK(hipMemcpyAsync(logits_all_host, m->logits_all.p, Mr * c.vocab * 4, hipMemcpyDeviceToHost, m->st));
    }
    HIPCHK(hipStreamSynchronize(m->st));  // `idx` is host memory
    m->hidden_out = nullptr;               // one-shot
    m->attn_out = nullptr;
}

// arm the session's own loop for the B rows just prefilled: every row at position S, step 0
void arm_session_rows(vc_model* m, const GenParams& g, const int* tail) {
    const int B = m->curB, S = m->curS;
    std::vector<int> rec((size_t)B * RS_STRIDE);
    fill_rows(rec.data(), B, g, S, tail, 0, m->out_stride);
    HIPCHK(hipMemsetAsync(m->rows.p, 0, m->rows.cap, m->st));
    HIPCHK(hipMemcpyAsync(m->rows.p, rec.data(), rec.size() * 4, hipMemcpyHostToDevice, m->st));
    HIPCHK(hipStreamSynchronize(m->st));
    m->cur_pos = S;
}

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

// =================================================================================================
// C ABI
// =================================================================================================
#define GUARD_BEGIN try {
// the HIP current device is per host thread: sessions may be driven from any thread
#define USE_DEVICE(ctxp)                       \
    do {                                       \
        HIPCHK(hipSetDevice((ctxp)->device)); \
        t_stream = (ctxp)->stream;             \
    } while (0)
// The HIP runtime keeps a per-thread "last error" that a LATER caller of hipGetLastError() — torch checks it after its own
// launches — would inherit from a query of ours that legitimately returned non-success (hipEventQuery: not ready, ...).
// Every entry point leaves it clean; VC_DEBUG_HIP=1 reports what it found.
#define CLEAR_HIP_LAST_ERROR(where)                                                                  \
    do {                                                                                             \
        hipError_t le_ = hipGetLastError();                                                          \
        if (le_ != hipSuccess && getenv("VC_DEBUG_HIP"))                                             \
            fprintf(stderr, "[vcoder_amd] %s left HIP last-error %d (%s)\n", where, (int)le_, hipGetErrorString(le_)); \
    } while (0)
#define GUARD_END(ctxp)                                   \
    }                                                     \
    catch (const Fail& f) {                               \
        if (ctxp) (ctxp)->err = f.msg;                    \
        CLEAR_HIP_LAST_ERROR(__func__);                   \
        return f.code;                                    \
    }                                                     \
    catch (const std::exception& e) {                     \
        if (ctxp) (ctxp)->err = e.what();                 \
        CLEAR_HIP_LAST_ERROR(__func__);                   \
        return VC_ERR_INVALID;                            \
    }                                                     \
    CLEAR_HIP_LAST_ERROR(__func__);                       \
    return VC_OK;

VC_API int vc_init(int device_id, vc_ctx** out) {
    if (!out) return VC_ERR_INVALID;
    *out = nullptr;
    vc_ctx* ctx = new vc_ctx();
    GUARD_BEGIN
    int n = 0;
    HIPCHK(hipGetDeviceCount(&n));
    REQUIRE(n > 0 && device_id >= 0 && device_id < n, VC_ERR_HIP, "no HIP device %d (found %d)", device_id, n);
    HIPCHK(hipSetDevice(device_id));
    ctx->device = device_id;
    ctx->stream = make_stream();
    *out = ctx;
    GUARD_END(ctx)
}
VC_API void vc_shutdown(vc_ctx* ctx) {
    if (!ctx) return;
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}
VC_API const char* vc_last_error(vc_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }
VC_API int vc_synchronize(vc_ctx* ctx) {
    GUARD_BEGIN
    USE_DEVICE(ctx);
    HIPCHK(hipStreamSynchronize(ctx->stream));
    GUARD_END(ctx)
}
VC_API void* vc_stream(vc_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }

VC_API int vc_model_create(vc_ctx* ctx, const vc_model_cfg* cfg, vc_model** out) {
    if (!ctx || !cfg || !out) return VC_ERR_INVALID;
    *out = nullptr;
    vc_model* m = nullptr;
    GUARD_BEGIN
    USE_DEVICE(ctx);
    const vc_model_cfg& c = *cfg;
    REQUIRE(c.variant >= 0 && c.variant <= 2, VC_ERR_INVALID, "bad variant %d", c.variant);
    REQUIRE(c.hidden % c.heads == 0 && c.vit_hidden % c.vit_heads == 0, VC_ERR_INVALID, "hidden %% heads != 0");
    const int hd = c.hidden / c.heads, vhd = c.vit_hidden / c.vit_heads;
    REQUIRE((hd == 128 || hd == 64) && (vhd == 64 || vhd == 128), VC_ERR_INVALID,
            "head dims (%d LLM, %d ViT) must be 64 or 128", hd, vhd);
    REQUIRE(c.hidden % 64 == 0 && c.ffn % 64 == 0 && c.vit_hidden % 64 == 0 && c.vit_ffn % 64 == 0 && c.vocab % 16 == 0,
            VC_ERR_INVALID, "hidden/ffn sizes must be multiples of 64 and vocab of 16");
    REQUIRE(c.vit_layers_used >= 0 && c.vit_layers_used <= c.vit_layers, VC_ERR_INVALID, "bad vit_layers_used");
    REQUIRE(c.vit_image % c.vit_patch == 0, VC_ERR_INVALID, "image size must be a multiple of the patch size");
    REQUIRE(c.mm_proj_depth >= 0 && c.seg_proj_depth >= 0, VC_ERR_INVALID, "bad projector depth");
    REQUIRE(c.max_positions >= 64 && c.max_positions <= 4096, VC_ERR_INVALID,
            "max_positions %d: the decode attention keeps a row's scores in LDS (<= 4096 keys)", c.max_positions);
    m = new vc_model();
    m->ctx = ctx;
    m->c = c;
    m->st = ctx->stream;
    m->hd = hd;
    m->vhd = vhd;
    const int g = c.vit_image / c.vit_patch;
    m->P = g * g;
    m->Tv = m->P + 1;
    m->Kpatch = 3 * c.vit_patch * c.vit_patch;
    m->Kpad = (int)rup(m->Kpatch, 64);
    m->npart = (int)rup(c.hidden / 16, 16);
    const int D = c.hidden, F = c.ffn, V = c.vocab, Dv = c.vit_hidden, Fv = c.vit_ffn;
    m->embed = walloc<bf16_t>(m, (size_t)V * D);
    m->lm_head = walloc<bf16_t>(m, (size_t)V * D);
    m->final_norm = walloc<float>(m, D);
    m->llm.resize(c.layers);
    for (auto& L : m->llm) {
        L.in_norm = walloc<float>(m, D);
        L.post_norm = walloc<float>(m, D);
        L.qkv_w = walloc<bf16_t>(m, (size_t)3 * D * D);
        L.o_w = walloc<bf16_t>(m, (size_t)D * D);
        L.gate_tmp = walloc<bf16_t>(m, (size_t)F * D);
        L.up_tmp = walloc<bf16_t>(m, (size_t)F * D);
        L.gu_w = walloc<bf16_t>(m, (size_t)2 * F * D);
        L.down_w = walloc<bf16_t>(m, (size_t)D * F);
        L.qkv_p = L.o_p = L.gu_p = L.down_p = nullptr;
    }
    auto mkproj = [&](Projector& pj, int depth) {
        pj.depth = depth;
        for (int l = 0; l < depth; ++l) {
            pj.w.push_back(walloc<bf16_t>(m, (size_t)D * (l == 0 ? Dv : D)));
            pj.b.push_back(walloc<float>(m, D));
        }
    };
    mkproj(m->mm, c.mm_proj_depth);
    if (c.variant != VC_VARIANT_LLAVA) mkproj(m->seg, c.seg_proj_depth);
    m->vit_cls = walloc<float>(m, Dv);
    m->vit_pos = walloc<float>(m, (size_t)m->Tv * Dv);
    m->vit_pre_w = walloc<float>(m, Dv);
    m->vit_pre_b = walloc<float>(m, Dv);
    m->vit_patch_w = walloc<bf16_t>(m, (size_t)Dv * m->Kpad, true);
    m->vit.resize(c.vit_layers_used);
    for (auto& L : m->vit) {
        L.ln1_w = walloc<float>(m, Dv); L.ln1_b = walloc<float>(m, Dv);
        L.ln2_w = walloc<float>(m, Dv); L.ln2_b = walloc<float>(m, Dv);
        L.qkv_w = walloc<bf16_t>(m, (size_t)3 * Dv * Dv); L.qkv_b = walloc<float>(m, 3 * Dv);
        L.out_w = walloc<bf16_t>(m, (size_t)Dv * Dv); L.out_b = walloc<float>(m, Dv);
        L.fc1_w = walloc<bf16_t>(m, (size_t)Fv * Dv); L.fc1_b = walloc<float>(m, Fv);
        L.fc2_w = walloc<bf16_t>(m, (size_t)Dv * Fv); L.fc2_b = walloc<float>(m, Dv);
    }
    mark_needed(m);
    for (auto& e : m->ev) HIPCHK(hipEventCreate(&e));
    *out = m;
    GUARD_END(ctx)
}

/* A second SESSION on the same weights: own stream (the new ctx), own workspaces / KV cache / hipGraph; weight tensors are
 * shared read-only with `parent` (which must be finalized and must outlive the session).  Several sessions let the GPU
 * overlap one batch's MFMA-bound prefill and per-launch ramps with another batch's HBM-bound decode. */
VC_API int vc_model_create_shared(vc_ctx* ctx, vc_model* parent, vc_model** out) {
    if (!ctx || !parent || !out) return VC_ERR_INVALID;
    *out = nullptr;
    GUARD_BEGIN
    USE_DEVICE(ctx);
    REQUIRE(parent->finalized, VC_ERR_STATE, "parent model is not finalized");
    vc_model* m = new vc_model();
    m->ctx = ctx;
    m->c = parent->c;
    m->st = ctx->stream;
    m->finalized = true;
    m->owns_weights = false;
    m->root = parent->root ? parent->root : parent;
    m->weight_format = parent->weight_format;
    m->P = parent->P; m->Tv = parent->Tv; m->Kpatch = parent->Kpatch; m->Kpad = parent->Kpad;
    m->hd = parent->hd; m->vhd = parent->vhd; m->npart = parent->npart;
    m->vit_cls = parent->vit_cls; m->vit_pos = parent->vit_pos; m->vit_pre_w = parent->vit_pre_w;
    m->vit_pre_b = parent->vit_pre_b; m->vit_patch_w = parent->vit_patch_w;
    m->vit = parent->vit; m->llm = parent->llm; m->mm = parent->mm; m->seg = parent->seg;
    m->embed = parent->embed; m->lm_head = parent->lm_head; m->lm_head_p = parent->lm_head_p;
    m->final_norm = parent->final_norm; m->rope_cos = parent->rope_cos; m->rope_sin = parent->rope_sin;
    for (auto& e : m->ev) HIPCHK(hipEventCreate(&e));
    *out = m;
    GUARD_END(ctx)
}

VC_API void vc_model_destroy(vc_model* m) {
    if (!m) return;
    (void)hipSetDevice(m->ctx->device);
    (void)hipStreamSynchronize(m->st);
    if (m->pool) {  // the root's decode pool (sessions must be gone: they prefill into its rows)
        pool_destroy(m->pool);
        m->pool = nullptr;
    }
    if (m->graph) (void)hipGraphExecDestroy(m->graph);
    m->graph = nullptr;
    if (m->owns_weights)
        for (void* p : m->owned) (void)hipFree(p);
    for (Buf* b : {&m->stage, &m->stage2, &m->v_pixels, &m->v_cols, &m->v_patches, &m->v_x, &m->v_xn, &m->v_qkv, &m->v_q,
                   &m->v_k, &m->v_vt, &m->v_attn, &m->v_h, &m->v_sel, &m->v_mid, &m->feats, &m->x, &m->xn, &m->qkv, &m->q,
                   &m->attn, &m->h, &m->kc, &m->vc, &m->vt_pre, &m->row_src, &m->last_idx, &m->xl, &m->logits_all, &m->x_dec,
                   &m->xg_dec, &m->qkv_dec, &m->attn_dec, &m->h_dec, &m->logits, &m->next_tok, &m->rows,
                   &m->out_ids, &m->dsum, &m->ssq, &m->sk_scratch, &m->sk_counters, &m->gemm_ws, &m->s_cols, &m->s_patches, &m->s_vx, &m->s_vxn, &m->s_vqkv,
                   &m->s_vq, &m->s_vk, &m->s_vv, &m->s_vattn, &m->s_vh, &m->s_sel, &m->s_mid, &m->s_feats, &m->s_xn, &m->s_qkv,
                   &m->s_q, &m->s_attn, &m->s_h, &m->s_kc, &m->s_vc, &m->s_xl, &m->pp_src, &m->pp_sq, &m->pp_tmp, &m->pp_out,
                   &m->pp_tab, &m->pp_f32, &m->kmask, &m->hidden_tmp, &m->attn_q, &m->a8, &m->a8_scale, &m->p_ssq, &m->p_rstd, &m->k_pre})
        b->release();
    for (auto& e : m->ev)
        if (e) (void)hipEventDestroy(e);
    delete m;
    CLEAR_HIP_LAST_ERROR("vc_model_destroy");
}

VC_API int vc_model_load_tensor(vc_model* m, const char* hf_key, const void* host_ptr, int dtype, const int64_t* shape,
                                int ndim) {
    if (!m) return VC_ERR_INVALID;
    int rc = VC_OK;
    GUARD_BEGIN
    USE_DEVICE(m->ctx);
    REQUIRE(hf_key && host_ptr && shape && ndim >= 1 && ndim <= 8, VC_ERR_INVALID, "bad load_tensor arguments");
    REQUIRE(dtype == VC_F32 || dtype == VC_BF16, VC_ERR_INVALID, "dtype must be VC_F32 or VC_BF16");
    REQUIRE(!m->finalized, VC_ERR_STATE, "model already finalized");
    size_t numel = 1;
    for (int i = 0; i < ndim; ++i) numel *= (size_t)shape[i];
    const size_t bytes = numel * (dtype == VC_F32 ? 4 : 2);
    m->stage.ensure(bytes);
    HIPCHK(hipMemcpyAsync(m->stage.p, host_ptr, bytes, hipMemcpyHostToDevice, m->st));
    rc = place_tensor(m, hf_key, m->stage.p, dtype, shape, ndim);
    HIPCHK(hipStreamSynchronize(m->st));
    if (rc != VC_OK) return rc;
    GUARD_END(m->ctx)
}

VC_API int vc_model_synth_tensor(vc_model* m, const char* hf_key, const int64_t* shape, int ndim, uint32_t tensor_seed,
                                 float offset, float halfwidth) {
    if (!m) return VC_ERR_INVALID;
    int rc = VC_OK;
    GUARD_BEGIN
    USE_DEVICE(m->ctx);
    REQUIRE(hf_key && shape && ndim >= 1 && ndim <= 8, VC_ERR_INVALID, "bad synth_tensor arguments");
    REQUIRE(!m->finalized, VC_ERR_STATE, "model already finalized");
    size_t numel = 1;
    for (int i = 0; i < ndim; ++i) numel *= (size_t)shape[i];
    REQUIRE(numel < ((size_t)1 << 32), VC_ERR_INVALID, "tensor too large for the 32-bit generator");
#if VC_OPERAND_FP16
    // the fp16-operand build: the generator's values (on the bfloat16 grid, as in every build) are staged as fp32 and take the fp32
    // load path — VC_BF16 sources are bfloat16 BITS, which is not what launch_synth_bf16 writes here
    m->stage.ensure(numel * 4);
    launch_synth_f32(m->stage.as<float>(), numel, tensor_seed, offset, halfwidth, m->st);
    rc = place_tensor(m, hf_key, m->stage.p, VC_F32, shape, ndim);
#else
    m->stage.ensure(numel * 2);
    launch_synth_bf16(m->stage.as<bf16_t>(), numel, tensor_seed, offset, halfwidth, m->st);
    rc = place_tensor(m, hf_key, m->stage.p, VC_BF16, shape, ndim);
#endif
    HIPCHK(hipStreamSynchronize(m->st));
    if (rc != VC_OK) return rc;
    GUARD_END(m->ctx)
}

/* The same generator with the value classes of the reference's own checkpoints: rounding 0 = bf16 (vc_model_synth_tensor), 1 = the
 * value an fp16 checkpoint holds, 2 = unrounded fp32 (vcoder_amd/synth.py synth_tensor(rounding=...)).  1 and 2 go through the fp32
 * load path, i.e. keep weight lo planes (vc_model_inexact_tensors). */
VC_API int vc_model_synth_tensor_rounded(vc_model* m, const char* hf_key, const int64_t* shape, int ndim, uint32_t tensor_seed,
                                         float offset, float halfwidth, int rounding) {
    if (!m) return VC_ERR_INVALID;
    if (rounding == 0) return vc_model_synth_tensor(m, hf_key, shape, ndim, tensor_seed, offset, halfwidth);
    int rc = VC_OK;
    GUARD_BEGIN
    USE_DEVICE(m->ctx);
    REQUIRE(hf_key && shape && ndim >= 1 && ndim <= 8 && (rounding == 1 || rounding == 2), VC_ERR_INVALID, "bad synth_tensor arguments");
    REQUIRE(!m->finalized, VC_ERR_STATE, "model already finalized");
    size_t numel = 1;
    for (int i = 0; i < ndim; ++i) numel *= (size_t)shape[i];
    REQUIRE(numel < ((size_t)1 << 32), VC_ERR_INVALID, "tensor too large for the 32-bit generator");
    m->stage.ensure(numel * 4);
    launch_synth_f32_rounded(m->stage.as<float>(), numel, tensor_seed, offset, halfwidth, rounding, m->st);
    rc = place_tensor(m, hf_key, m->stage.p, VC_F32, shape, ndim);
    HIPCHK(hipStreamSynchronize(m->st));
    if (rc != VC_OK) return rc;
    GUARD_END(m->ctx)
}

/* 0: bf16 MFMA path (default, benchmarked); 1: strict fp32 path (fp32 activations + fp32 MFMA, ~1e-6 from the fp32 CPU
 * reference; slow).  Takes effect at the next prefill. */
VC_API int vc_model_set_precision(vc_model* m, int mode) {
    if (!m || mode < 0 || mode > 2) return VC_ERR_INVALID;
    if (mode == 2) {
        // split mode on a checkpoint with weight lo planes: its decode GEMV is the workgroup-shared form only — refuse HERE what
        // would otherwise surface as an exception out of a launch (possibly inside a graph capture)
        const vc_model* root = m->root ? m->root : m;
        // (the e4m3 weight formats re-round the decoder linears and drop their lo planes: only K = hidden matrices keep one then)
        if (!root->lo_of.empty() && (!gemv_wg_enabled() || root->c.hidden % 64 != 0 || (root->weight_format == 0 && root->c.ffn % 64 != 0))) {
            m->ctx->err = std::string("precision mode split on a checkpoint with weight lo planes (fp16 / fp32 values) needs the workgroup-shared "
                              "decode GEMV (vck_set_gemv_variant != 0, hidden_size and intermediate_size multiples of 64)");
            return VC_ERR_STATE;
        }
    }
    if (mode != m->precision) {  // the decode graph bakes the step's kernels and buffers in
        (void)hipSetDevice(m->ctx->device);
        (void)hipStreamSynchronize(m->st);
        drop_graph(m);
    }
    m->precision = mode;
    m->cur_pos = -1;
    return VC_OK;
}

/* 0: bf16 weights (default).  1: W8A16 — q/k/v/o/gate/up/down of every decoder layer are quantised at finalize to OCP
 * e4m3 with a per-output-row power-of-two scale; embeddings, lm_head, norms, the CLIP tower and the adapters stay
 * bf16/fp32.  2: fp8 (BASELINE config C5, "CDNA4 fp8 MFMA") — the same weights; the cached decode steps stream the bytes
 * as in 1 (bf16 activations), and the prefill's decoder linears additionally quantise their activation rows to e4m3
 * (per-token power-of-two scale) and run e4m3 x e4m3 on v_mfma_scale_f32_16x16x128_f8f6f4 at twice the bf16 MFMA rate.
 * Must be called before vc_model_finalize. */
VC_API int vc_model_set_weight_format(vc_model* m, int fmt) {
    if (!m || fmt < 0 || fmt > 2) return VC_ERR_INVALID;
    if (m->finalized) return VC_ERR_STATE;
    m->weight_format = fmt;
    return VC_OK;
}

/* Weight format 2 ("fp8") keeps the KV cache of its decode steps in e4m3 as well (1 byte per element, unscaled, saturating at 448;
 * default on).  on = 0: bf16 rows, as the other formats.  Before vc_model_finalize. */
VC_API int vc_model_set_fp8_kv(vc_model* m, int on) {
    if (!m) return VC_ERR_INVALID;
    if (m->finalized) return VC_ERR_STATE;
    m->fp8_kv = on != 0;
    return VC_OK;
}

/* Batch invariance (the reference's rows do not depend on the batch size; §4 test 4: the gathered stream of
 * N ranks == the single-GPU stream of the same global batch).  By default a prefill GEMM whose last round of output tiles is short
 * cuts that round into K-slices, and the number of slices follows from the tile count — so a sample's low-order bits can differ
 * between a batch of 4 and two batches of 2 (1e-6-level; visible only through near-tied greedy choices).  on = 1: no remainder
 * split — every output row is summed in one fixed k order whatever shares its launch (the decode steps, attention and row kernels
 * already are), at the price of a short last round of tiles per GEMM (~2-4 % of a prefill).  Applies to all sessions of the model. */
VC_API int vc_model_set_batch_invariant(vc_model* m, int on) {
    if (!m) return VC_ERR_INVALID;
    (m->root ? m->root : m)->batch_invariant = on != 0;
    return VC_OK;
}

/* The prefill's QKV projection with RoPE + head split + KV-cache write in the GEMM's epilogue (default on): on = 0 runs the
 * two launches it replaces (EPI_BF16 GEMM + qkv_split_kernel) — same bits when the padded and the plain token count give the GEMM the
 * same tile rounds, A/B and regression switch; on = 2 takes the fused form for every problem size (default 1: from 1024 token rows,
 * where the 256 x 256 kernel is used anyway).  Applies to all sessions of the model. */
VC_API int vc_model_set_qkv_fused(vc_model* m, int on) {
    if (!m) return VC_ERR_INVALID;
    if (on < 0 || on > 2) return VC_ERR_INVALID;
    (m->root ? m->root : m)->qkv_fused = on;
    return VC_OK;
}

/* Parity diagnostic: the next prefills evaluate only the first n decoder layers (0 = all) and apply the final norm +
 * lm_head to that hidden state — the logits of the same checkpoint cut to n layers.  Lets a test chart how the bf16
 * path's deviation from the fp32 oracle grows with depth on ONE loaded model.  Decode steps always use every layer. */
VC_API int vc_model_set_layer_limit(vc_model* m, int n_layers) {
    if (!m || n_layers < 0) return VC_ERR_INVALID;
    m->layer_limit = n_layers;
    m->cur_pos = -1;
    return VC_OK;
}

/* 16-bit operand format this library was built for: 0 = bfloat16 (libvcoder_hip.so, the benchmarked path), 1 = IEEE fp16
 * (libvcoder_hip_f16.so: the same kernels built with -DVC_F16 — v_mfma_f32_16x16x32_f16, saturating fp16 conversions; the
 * precision of the reference's own GPU path, vcoder_llava/model/builder.py:39,142) */
VC_API int vc_operand_format(void) { return VC_OPERAND_FP16; }

/* Number of loaded tensors whose fp32 source held values bf16 cannot represent (an fp16 / fp32 checkpoint; 0 for a bf16 one).
 * Each keeps a bf16 lo plane: precision modes "strict" and "split" compute with hi + lo (the checkpoint's values to ~16 mantissa
 * bits, exact for fp16), the bf16 fast path with the bf16-rounded weights alone. */
VC_API int vc_model_inexact_tensors(vc_model* m) {
    if (!m) return VC_ERR_INVALID;
    const vc_model* r = m->root ? m->root : m;
    return (int)r->inexact_regions.size();
}

VC_API int vc_model_finalize(vc_model* m) {
    if (!m) return VC_ERR_INVALID;
    GUARD_BEGIN
    USE_DEVICE(m->ctx);
    if (m->finalized) return VC_OK;
    for (auto& kv : m->need) REQUIRE(kv.second, VC_ERR_STATE, "missing tensor '%s'", kv.first.c_str());
    const vc_model_cfg& c = m->c;
    const int D = c.hidden, F = c.ffn, V = c.vocab;
    // decode copy of one decoder linear: MFMA-packed bf16, or (W8A16) e4m3 super-tiles + per-row scales — the quantiser
    // also rewrites the row-major bf16 matrix with the dequantised values so prefill and decode share one set of weights
    auto decode_copy = [&](bf16_t* W, int N, int K, bf16_t*& Wp, float*& Ws, uint8_t*& Wq) {
        if (m->weight_format >= 1) {
            Wp = reinterpret_cast<bf16_t*>(walloc<uint8_t>(m, (size_t)N * K));
            Ws = walloc<float>(m, N);
            Wq = m->weight_format == 2 ? walloc<uint8_t>(m, (size_t)N * K) : nullptr;
            launch_quantize_fp8(W, reinterpret_cast<uint8_t*>(Wp), Ws, N, K, m->st, Wq);
        } else {
            Wp = walloc<bf16_t>(m, (size_t)N * K);
            launch_pack_weight(W, Wp, N, K, m->st);
        }
    };
    if (m->weight_format >= 1)
        REQUIRE(D % 64 == 0 && F % 64 == 0, VC_ERR_INVALID, "W8A16 needs hidden and ffn sizes divisible by 64");
    if (m->weight_format == 2)
        REQUIRE(D % 128 == 0 && F % 128 == 0, VC_ERR_INVALID, "the fp8 prefill GEMM needs hidden and ffn sizes divisible by 128");
    // lo planes of an inexact checkpoint (m->lo_of): interleaved / packed like their hi planes; the e4m3 weight formats re-round
    // the decoder linears anyway, so those drop theirs
    auto free_owned = [&](void* p) {
        for (auto& o : m->owned)
            if (o == p) { (void)hipFree(o); o = nullptr; }
    };
    auto drop_lo = [&](const void* hi) {
        auto it = m->lo_of.find(hi);
        if (it == m->lo_of.end()) return;
        free_owned(it->second);
        m->lo_of.erase(it);
    };
    auto pack_lo = [&](const bf16_t* W, const bf16_t* Wp, int N, int K) {   // the decode steps' packed copy of W's lo plane
        auto it = m->lo_of.find(W);
        if (it == m->lo_of.end()) return;
        bf16_t* lp = walloc<bf16_t>(m, (size_t)N * K);
        launch_pack_weight(it->second, lp, N, K, m->st);
        m->lo_of[Wp] = lp;
    };
    for (auto& L : m->llm) {
        launch_interleave_rows(L.gate_tmp, L.up_tmp, L.gu_w, F, D, m->st);
        if (m->lo_of.count(L.gate_tmp) || m->lo_of.count(L.up_tmp)) {
            for (bf16_t* t : {L.gate_tmp, L.up_tmp})
                if (!m->lo_of.count(t)) m->lo_of[t] = walloc<bf16_t>(m, (size_t)F * D, true);
            bf16_t* gu_lo = walloc<bf16_t>(m, (size_t)2 * F * D);
            launch_interleave_rows(m->lo_of[L.gate_tmp], m->lo_of[L.up_tmp], gu_lo, F, D, m->st);
            m->lo_of[L.gu_w] = gu_lo;
        }
        if (m->weight_format >= 1)
            for (const bf16_t* w : {L.qkv_w, L.o_w, L.gu_w, L.down_w}) drop_lo(w);
        decode_copy(L.qkv_w, 3 * D, D, L.qkv_p, L.qkv_s, L.qkv_q);
        decode_copy(L.o_w, D, D, L.o_p, L.o_s, L.o_q);
        decode_copy(L.gu_w, 2 * F, D, L.gu_p, L.gu_s, L.gu_q);
        decode_copy(L.down_w, D, F, L.down_p, L.down_s, L.down_q);
        if (m->weight_format == 0) {
            pack_lo(L.qkv_w, L.qkv_p, 3 * D, D);
            pack_lo(L.o_w, L.o_p, D, D);
            pack_lo(L.gu_w, L.gu_p, 2 * F, D);
            pack_lo(L.down_w, L.down_p, D, F);
        }
    }
    m->lm_head_p = walloc<bf16_t>(m, (size_t)V * D);
    launch_pack_weight(m->lm_head, m->lm_head_p, V, D, m->st);
    pack_lo(m->lm_head, m->lm_head_p, V, D);
    HIPCHK(hipStreamSynchronize(m->st));
    // gate/up staging copies (and their lo planes) are no longer needed
    for (auto& L : m->llm) {
        for (bf16_t** t : {&L.gate_tmp, &L.up_tmp}) {
            drop_lo(*t);
            free_owned(*t);
            *t = nullptr;
        }
    }
    // rope tables (LlamaRotaryEmbedding, [HF] llama/modeling_llama.py:73-127): inv_freq = theta^(-2i/hd) in fp32,
    // angle = pos * inv_freq in fp32, cos/sin in fp32
    const int half = m->hd / 2;
    std::vector<float> hc((size_t)c.max_positions * half), hs((size_t)c.max_positions * half);
    for (int i = 0; i < half; ++i) {
        const float inv = (float)(1.0 / pow((double)c.rope_theta, (double)(2 * i) / (double)m->hd));
        for (int p = 0; p < c.max_positions; ++p) {
            const float ang = (float)p * inv;
            hc[(size_t)p * half + i] = (float)cos((double)ang);
            hs[(size_t)p * half + i] = (float)sin((double)ang);
        }
    }
    m->rope_cos = walloc<float>(m, hc.size());
    m->rope_sin = walloc<float>(m, hs.size());
    HIPCHK(hipMemcpyAsync(m->rope_cos, hc.data(), hc.size() * 4, hipMemcpyHostToDevice, m->st));
    HIPCHK(hipMemcpyAsync(m->rope_sin, hs.data(), hs.size() * 4, hipMemcpyHostToDevice, m->st));
    HIPCHK(hipStreamSynchronize(m->st));
    m->stage.release();
    m->stage2.release();
    m->finalized = true;
    GUARD_END(m->ctx)
}

VC_API int vc_encode(vc_model* m, int modality, const float* pixels, int pixels_on_device, int B, float* out) {
    if (!m) return VC_ERR_INVALID;
    GUARD_BEGIN
    USE_DEVICE(m->ctx);
    REQUIRE(m->finalized, VC_ERR_STATE, "vc_model_finalize() has not been called");
    REQUIRE(pixels && B >= 1 && modality >= 0 && modality <= 2, VC_ERR_INVALID, "bad encode arguments");
    REQUIRE(modality == VC_MOD_IMAGE || m->c.variant != VC_VARIANT_LLAVA, VC_ERR_INVALID, "llava has no seg/depth encoder");
    PixSet pix{{nullptr, nullptr, nullptr}, {0, 0, 0}};
    pix.p[modality] = pixels;
    pix.n[modality] = B;
    if (m->precision == 1) run_vit_and_adapters_strict(m, pix, pixels_on_device);
    else if (m->precision == 2) run_vit_and_adapters_split(m, pix, pixels_on_device);
    else run_vit_and_adapters(m, pix, pixels_on_device);
    if (out) {
        const size_t n = (size_t)m->feat_rows[modality] * m->c.hidden;
        if (m->precision) {
            HIPCHK(hipMemcpyAsync(out, m->s_feats.as<float>() + (size_t)m->feat_off[modality] * m->c.hidden, n * 4,
                                  hipMemcpyDeviceToHost, m->st));
        } else {
            m->v_patches.ensure(n * 4);
            launch_bf16_to_f32(m->feats.as<bf16_t>() + (size_t)m->feat_off[modality] * m->c.hidden, m->v_patches.as<float>(),
                               n, m->st);
            HIPCHK(hipMemcpyAsync(out, m->v_patches.p, n * 4, hipMemcpyDeviceToHost, m->st));
        }
    }
    HIPCHK(hipStreamSynchronize(m->st));
    GUARD_END(m->ctx)
}

/* CLIPVisionTower.forward + feature_select (multimodal_encoder/clip_encoder.py:29-51): the UN-projected tower output the
 * reference's encode_* functions hand to the adapters — hidden_states[select_layer] of N images, CLS dropped for 'patch'.
 * out [N, R, vit_hidden] fp32 on the host (R = patches, +1 with 'cls_patch'). */
VC_API int vc_vision_tower_forward(vc_model* m, const float* pixels, int pixels_on_device, int N, float* out) {
    if (!m) return VC_ERR_INVALID;
    GUARD_BEGIN
    USE_DEVICE(m->ctx);
    REQUIRE(m->finalized, VC_ERR_STATE, "vc_model_finalize() has not been called");
    REQUIRE(pixels && out && N >= 1, VC_ERR_INVALID, "bad vision tower arguments");
    PixSet pix{{pixels, nullptr, nullptr}, {N, 0, 0}};
    int order[3], first[3];
    const int R = m->Tv - (m->c.vit_keep_cls ? 0 : 1);
    const size_t n = (size_t)N * R * m->c.vit_hidden;
    if (m->precision == 1) {
        run_vit_tower_strict(m, pix, pixels_on_device, order, first);
        HIPCHK(hipMemcpyAsync(out, m->s_sel.p, n * 4, hipMemcpyDeviceToHost, m->st));
    } else if (m->precision == 2) {
        run_vit_tower_split(m, pix, pixels_on_device, order, first);
        m->s_sel.ensure(n * 4);
        launch_select_rows_f32(m->v_x.as<float>(), m->s_sel.as<float>(), N, m->Tv, m->c.vit_keep_cls ? 0 : 1, m->c.vit_hidden, m->st);
        HIPCHK(hipMemcpyAsync(out, m->s_sel.p, n * 4, hipMemcpyDeviceToHost, m->st));
    } else {
        run_vit_tower(m, pix, pixels_on_device, order, first);
        m->v_patches.ensure(n * 4);
        launch_bf16_to_f32(m->v_sel.as<bf16_t>(), m->v_patches.as<float>(), n, m->st);
        HIPCHK(hipMemcpyAsync(out, m->v_patches.p, n * 4, hipMemcpyDeviceToHost, m->st));
    }
    HIPCHK(hipStreamSynchronize(m->st));
    GUARD_END(m->ctx)
}

/* images per sample for the NEXT vc_prefill* / vc_generate* call (one-shot): the reference's list / 5-D image form
 * (vcoder_ds_llava_arch.py:135-169), where sample b owns counts[b] images whose feature rows are spliced as ONE block at
 * its placeholder.  The pixel pointers of that call then hold sum(counts) images.  NULL = one image per sample. */
VC_API int vc_set_image_counts(vc_model* m, const int32_t* img, const int32_t* seg, const int32_t* depth, int B) {
    if (!m || B < 1) return VC_ERR_INVALID;
    const int32_t* src[3] = {img, seg, depth};
    for (int k = 0; k < 3; ++k) {
        m->img_counts[k].clear();
        if (!src[k]) continue;
        for (int b = 0; b < B; ++b) {
            if (src[k][b] < 1) {
                for (auto& v : m->img_counts) v.clear();
                m->ctx->err = "every sample needs at least one image per modality";
                return VC_ERR_INVALID;
            }
            m->img_counts[k].push_back(src[k][b]);
        }
    }
    return VC_OK;
}

/* output_hidden_states for the NEXT vc_prefill (one-shot): `out` (host, cap_floats floats) receives [(layers + 1), B, S, hidden]
 * fp32 — inputs_embeds, the residual stream behind every decoder layer, the last one after the final RMSNorm
 * ([HF] LlamaModel.forward all_hidden_states).  The buffer must stay valid until that vc_prefill returns. */
VC_API int vc_request_hidden_states(vc_model* m, float* out, size_t cap_floats) {
    if (!m) return VC_ERR_INVALID;
    m->hidden_out = out;
    m->hidden_cap = out ? cap_floats : 0;
    return VC_OK;
}

/* output_attentions for the NEXT vc_prefill (one-shot): `out` (host, cap_floats floats) receives [layers, B, heads, S, S] fp32,
 * the attention probabilities of every decoder layer (zeros above the diagonal and at hidden keys) */
VC_API int vc_request_attentions(vc_model* m, float* out, size_t cap_floats) {
    if (!m) return VC_ERR_INVALID;
    m->attn_out = out;
    m->attn_cap = out ? cap_floats : 0;
    return VC_OK;
}

/* attention_mask [B, T] (bytes, 0 = hidden) of the NEXT vc_prefill* / vc_generate* call (one-shot).  See include/vcoder_hip.h. */
VC_API int vc_set_attention_mask(vc_model* m, const uint8_t* mask, int B, int T) {
    if (!m || !mask || B < 1 || T < 1) return VC_ERR_INVALID;
    m->mask_next.assign(mask, mask + (size_t)B * T);
    m->mask_B = B;
    m->mask_T = T;
    return VC_OK;
}
/* the cached decode steps behind the current vc_prefill see every key again (what the reference's multimodal decode path does:
 * vcoder_ds_llava_arch.py:130-133 replaces the mask with ones) */
VC_API int vc_clear_attention_mask(vc_model* m) {
    if (!m) return VC_ERR_INVALID;
    m->kmask_in_decode = false;
    m->mask_next.clear();
    return VC_OK;
}

/* KV-cache slots the next vc_prefill keeps free behind the prompt for vc_decode_step (default 64; clamped to
 * max_position_embeddings).  A decode loop that outruns the reserve still works — the cache grows, at the cost of a copy. */
VC_API int vc_model_reserve_decode(vc_model* m, int max_new_tokens) {
    if (!m || max_new_tokens < 0) return VC_ERR_INVALID;
    m->reserve_new = max_new_tokens;
    return VC_OK;
}

VC_API int vc_prefill(vc_model* m, const int64_t* ids, int B, int T, const float* img, const float* seg,
                      const float* depth, int pixels_on_device, int has_attention_mask, float* logits_last,
                      float* logits_all, int* S_out) {
    if (!m) return VC_ERR_INVALID;
    OneShotReset one_shot{m};
    GUARD_BEGIN
    USE_DEVICE(m->ctx);
    m->cur_pos = -1;
    do_prefill(m, ids, B, T, img, seg, depth, pixels_on_device, has_attention_mask, -std::max(m->reserve_new, 1), true, S_out);
    ensure_out_ids(m, B, 1);
    finish_prefill(m, session_kv(m), logits_all);
    // a vc_decode_step loop behind this prefill keeps the hidden keys hidden (a caller that carries its attention_mask through
    // the steps: forward() without images); vc_clear_attention_mask() gives the steps the all-ones mask the reference's
    // multimodal decode path builds (vcoder_ds_llava_arch.py:130-133)
    m->kmask_in_decode = m->has_kmask;
    // greedy choice of the prefill logits, so that vc_decode_step(tok = NULL) continues the sequence: nothing is recorded
    // (max_new 0), no EOS bookkeeping, the position stays at S
    GenParams g;
    arm_session_rows(m, g, nullptr);
    const LoopView v = session_view(m);
    launch_select_embed(select_args(m, v, v.logits, B, 0), m->st);
    HIPCHK(hipStreamSynchronize(m->st));
    if (logits_last) {
        HIPCHK(hipMemcpyAsync(logits_last, m->logits.p, (size_t)B * m->c.vocab * 4, hipMemcpyDeviceToHost, m->st));
        HIPCHK(hipStreamSynchronize(m->st));
    }
    GUARD_END(m->ctx)
}

/* splice only (no decoder layers): inputs_embeds [B,S,hidden] fp32 to host.  Used by the parity tests of a7. */
VC_API int vc_prefill_embeds_only(vc_model* m, const int64_t* ids, int B, int T, const float* img, const float* seg,
                                  const float* depth, int pixels_on_device, int has_attention_mask, float* out_host,
                                  int* S_out) {
    if (!m) return VC_ERR_INVALID;
    OneShotReset one_shot{m};
    GUARD_BEGIN
    USE_DEVICE(m->ctx);
    m->cur_pos = -1;
    int S = 0;
    do_prefill(m, ids, B, T, img, seg, depth, pixels_on_device, has_attention_mask, -1, true, &S);
    if (S_out) *S_out = S;
    if (out_host) {
        HIPCHK(hipMemcpyAsync(out_host, m->x.p, (size_t)B * S * m->c.hidden * 4, hipMemcpyDeviceToHost, m->st));
        HIPCHK(hipStreamSynchronize(m->st));
    }
    GUARD_END(m->ctx)
}

/* The spliced length S a vc_prefill* / vc_generate* call with these arguments would produce — the splice plan alone (ids,
 * which modalities are present, the per-sample image counts of vc_set_image_counts, is_depth_zero of the depth pixels): no
 * tower pass, no cache or mask state touched.  Same errors as the real call's plan (IndexError / quirk 6). */
VC_API int vc_plan_spliced_len(vc_model* m, const int64_t* ids, int B, int T, const float* img, const float* seg,
                               const float* depth, int pixels_on_device, int has_attention_mask, int* S_out) {
    if (!m) return VC_ERR_INVALID;
    struct Reset {
        vc_model* m;
        ~Reset() {
            m->plan_only = false;
            for (auto& v : m->img_counts) v.clear();
        }
    } reset{m};
    GUARD_BEGIN
    USE_DEVICE(m->ctx);
    m->plan_only = true;
    do_prefill(m, ids, B, T, img, seg, depth, pixels_on_device, has_attention_mask, -1, false, S_out);
    GUARD_END(m->ctx)
}

/* Parity diagnostic (tests/: per-layer teacher forcing): decoder layers [l0, l1) of a PREFILL applied to a caller-supplied
 * residual stream x_in [B, S, hidden] (fp32, host) at positions 0..S-1, in the model's weight format and precision mode;
 * x_out receives the residual stream behind layer l1 - 1.  Feeding every layer the ORACLE's input isolates that layer's
 * arithmetic: quantisation noise of the layers in front cannot compound.  Uses the session's own KV cache. */
VC_API int vc_debug_prefill_layers(vc_model* m, int l0, int l1, const float* x_in, int B, int S, float* x_out) {
    if (!m) return VC_ERR_INVALID;
    GUARD_BEGIN
    USE_DEVICE(m->ctx);
    REQUIRE(m->finalized, VC_ERR_STATE, "vc_model_finalize() has not been called");
    REQUIRE(x_in && x_out && B >= 1 && B <= VC_MAX_ROWS && S >= 1 && l0 >= 0 && l1 > l0 && l1 <= m->c.layers, VC_ERR_INVALID,
            "bad layer range / shape");
    m->cur_pos = -1;
    m->has_kmask = m->kmask_in_decode = false;   // a diagnostic pass over caller-supplied rows: no key mask of an earlier prefill
    ensure_llm(m, B, S + 1);
    const size_t n = (size_t)B * S * m->c.hidden;
    HIPCHK(hipMemcpyAsync(m->x.p, x_in, n * 4, hipMemcpyHostToDevice, m->st));
    if (m->precision == 1) {
        ensure_strict(m, B, m->capS);
        REQUIRE(l0 == 0 && l1 == m->c.layers, VC_ERR_INVALID, "strict mode runs the whole stack");
        run_llm_layers_strict(m, m->x.as<float>(), B, S, nullptr, true);
    } else if (m->precision == 2) {
        run_prefill_layers_split(m, session_kv(m), B, S, l0, l1);
    } else {
        run_prefill_layers(m, session_kv(m), B, S, l0, l1);
    }
    HIPCHK(hipMemcpyAsync(x_out, m->x.p, n * 4, hipMemcpyDeviceToHost, m->st));
    HIPCHK(hipStreamSynchronize(m->st));
    GUARD_END(m->ctx)
}

/* Beam search support: the KV rows of the session's current loop are permuted, row r <- old row src_rows[r] (live prefix only) —
 * `past_key_values` reordered by beam_idx ([HF] generation/utils.py: _reorder_cache after every beam step). */
VC_API int vc_reorder_cache(vc_model* m, const int32_t* src_rows, int B) {
    if (!m) return VC_ERR_INVALID;
    GUARD_BEGIN
    USE_DEVICE(m->ctx);
    REQUIRE(m->cur_pos >= 0, VC_ERR_STATE, "vc_reorder_cache before vc_prefill");
    REQUIRE(src_rows && B == m->curB, VC_ERR_INVALID, "beam_idx must have one entry per row of the current batch (%d)", m->curB);
    bool identity = true;
    for (int b = 0; b < B; ++b) {
        REQUIRE(src_rows[b] >= 0 && src_rows[b] < B, VC_ERR_INDEX, "beam index %d out of range", src_rows[b]);
        identity = identity && src_rows[b] == b;
    }
    if (!identity) {
        const vc_model_cfg& c = m->c;
        const int live = m->cur_pos, H = c.heads;
        const bool strict = m->precision == 1;
        const size_t es = strict ? 4 : (size_t)m->kv_es;
        const int capS = strict ? m->s_capS : m->capS, capB = strict ? m->s_capB : m->capB;
        const size_t cap_row = (size_t)capS * m->hd * es, live_row = (size_t)live * m->hd * es;
        m->stage.ensure((size_t)B * H * live_row);
        m->stage2.ensure((size_t)B * 4);
        HIPCHK(hipMemcpyAsync(m->stage2.p, src_rows, (size_t)B * 4, hipMemcpyHostToDevice, m->st));
        char* kb = reinterpret_cast<char*>(strict ? m->s_kc.p : m->kc.p);
        char* vb = reinterpret_cast<char*>(strict ? m->s_vc.p : m->vc.p);
        for (int l = 0; l < c.layers; ++l) {
            const size_t off = (size_t)l * capB * H * cap_row;
            launch_kv_permute(kb + off, m->stage.p, m->stage2.as<int>(), B, H, cap_row, live_row, m->st);
            launch_kv_permute(vb + off, m->stage.p, m->stage2.as<int>(), B, H, cap_row, live_row, m->st);
        }
        HIPCHK(hipStreamSynchronize(m->st));
    }
    GUARD_END(m->ctx)
}

VC_API int vc_decode_step(vc_model* m, const int32_t* tok, float* logits, int32_t* next_tok) {
    if (!m) return VC_ERR_INVALID;
    GUARD_BEGIN
    USE_DEVICE(m->ctx);
    REQUIRE(m->cur_pos >= 0, VC_ERR_STATE, "vc_decode_step before vc_prefill");
    if (m->cur_pos + 1 > m->capS) grow_kv(m, m->cur_pos + 1);
    const int B = m->curB;
    if (tok) {
        for (int b = 0; b < B; ++b)
            REQUIRE(tok[b] >= 0 && tok[b] < m->c.vocab, VC_ERR_INDEX, "index out of range in self (token id %d)", tok[b]);
        HIPCHK(hipMemcpyAsync(m->next_tok.p, tok, B * 4, hipMemcpyHostToDevice, m->st));
        launch_embed_tokens_ssq(m->next_tok.as<int>(), m->embed, m->x_dec.as<float>(), m->ssq.as<float>(), m->llm[0].in_norm,
                                m->xg_dec.as<bf16_t>(), B, m->c.hidden, m->npart, m->st, session_view(m).split_G,
                                m->precision ? lo_plane(m, m->embed) : nullptr);
    }
    struct StepRequests {   // one-shot, also when the step fails
        vc_model* m;
        ~StepRequests() {
            m->hidden_out = nullptr;
            m->attn_out = nullptr;
            m->hidden_cap = m->attn_cap = 0;
        }
    } step_requests{m};
    if (m->precision == 1) {
        enqueue_decode_step_strict(m, B);
    } else if (m->hidden_out || m->attn_out) {
        enqueue_decode_step_diag(m, session_view(m), B, m->cur_pos);
    } else {
        ensure_graph(m, B);
        HIPCHK(hipGraphLaunch(m->graph, m->st));
    }
    m->cur_pos += 1;
    if (logits) HIPCHK(hipMemcpyAsync(logits, m->logits.p, (size_t)B * m->c.vocab * 4, hipMemcpyDeviceToHost, m->st));
    if (next_tok) HIPCHK(hipMemcpyAsync(next_tok, m->next_tok.p, B * 4, hipMemcpyDeviceToHost, m->st));
    HIPCHK(hipStreamSynchronize(m->st));
    GUARD_END(m->ctx)
}

namespace {

// first column count at which a row of `produced` ids is stopped by one of the stop sequences (host restatement of the
// select kernel's suffix match; used to trim the returned columns exactly where HF's loop would have stopped)
int row_stop_end(const GenParams& g, const int32_t* row, const int* tail_b, int produced) {
    constexpr int TL = VC_MAX_STOP_LEN - 1;
    for (int st = 0; st < produced; ++st)
        for (int i = 0; i < g.n_stop; ++i) {
            const int L = g.stop[i][0];
            bool ok = L > 0;
            for (int j = 0; j < L && ok; ++j) {
                const int back = L - 1 - j;
                const int v = st - back >= 0 ? row[st - back] : tail_b[TL + (st - back)];
                ok = v == g.stop[i][1 + j];
            }
            if (ok) return st + 1;
        }
    return produced;
}

// columns HF's loop would have returned: it stops right after the first step at which every row has produced EOS / met a stop
int trim_columns(const GenParams& g, const int32_t* out_ids, int ld, const int* tail, int B, int produced) {
    if (!(g.eos >= 0 || g.n_stop > 0)) return produced;
    int last = 0;
    for (int b = 0; b < B; ++b) {
        int e = produced;
        if (g.eos >= 0)
            for (int s_ = 0; s_ < produced; ++s_)
                if (out_ids[(size_t)b * ld + s_] == g.eos) { e = s_ + 1; break; }
        if (g.n_stop > 0) e = std::min(e, row_stop_end(g, out_ids + (size_t)b * ld, tail + (size_t)b * (VC_MAX_STOP_LEN - 1), produced));
        last = std::max(last, e);
    }
    return std::min(produced, last);
}

// one encode + prefill at a time PER PROCESS (two MFMA-bound phases side by side only slow each other down).  The deployment is
// one process per GPU, so per process == per device; a process driving several devices would serialise their
// prefills' ENQUEUEING here (not their execution) and should then gate per device instead.
std::mutex g_prefill_gate;

// =================================================================================================
// Decode pool: the cached decode steps of CONCURRENT generate() calls share one step.
//
// A decode step streams every decoder weight once whatever the number of rows in it (<= 32: two MFMA token-slot groups
// per weight tile), so k requests decoding side by side on private loops read the weights k times per token where one
// pooled step reads them once — the HBM leg of the composite roofline per image drops from  W + KV  to  W / k + KV
// (DESIGN.md section 2b).  The reference's callers issue independent generate() calls (serve/cli.py:122, the eval
// loaders' per-sample loop, one process per GPU in scripts/v1_5/eval/cost_depth.sh); the pool is what lets several of them
// in flight on one GPU behave like one larger batch during decode without changing what any of them computes: rows are
// independent in every kernel of the step, a row's arithmetic does not depend on which other rows are present, and the
// ids a request gets are bit-identical to the ones its own loop would produce (tests: pooled == session loop).
//
//   request thread (vc_generate)                         driver thread (one per pool)
//   ---------------------------------------------       ---------------------------------------------------------------
//   take B free rows of the pool                         loop:
//   encode + prefill on ITS stream, keys / values          admit pending requests between two steps: wait for their prefill
//     written straight into the pool's KV rows               event on the pool stream, write their RowState records, select
//   record prefill-done event, queue the request             token 0 from the request's prefill logits (select kernel)
//   sleep until done (streaming: wake per report)          replay the step graph over the first 16 / all 32 rows
//   copy its out_ids rows, free the rows                    count steps per request; retire finished ones (rows inactive)
//
// Only the driver touches the pool's decode state, always on the pool's stream, so joins and retirements are ordered
// between steps without any host synchronisation of the GPU; the only host waits are a bounded run-ahead (two steps) and
// the completion events the request threads sleep on.
struct PoolRequest {
    vc_model* sess = nullptr;
    int row0 = 0, B = 0;
    GenParams g;
    std::vector<int> tail;
    int* rec = nullptr;               // pinned host copy of the B RowState records
    hipEvent_t prefill_done = nullptr, join_ev = nullptr, done_ev = nullptr, report_ev = nullptr;
    int steps_left = 0;               // pool steps still to run for it
    int produced = 0;                 // columns of out_ids written so far (driver's count)
    bool can_finish = false;
    // early finish poll (EOS / stops): FINISHED words copied out asynchronously
    int* fin_host = nullptr;          // pinned [B]
    hipEvent_t fin_ev = nullptr;
    bool fin_pending = false;
    int fin_at = 0;                   // `produced` at the time the poll was issued
    // hand-off to the request thread
    bool done = false, failed = false;
    std::string err;
    int avail = 0;                    // columns the request thread may read (streaming)
    bool report_taken = true;
    vc_token_cb cb = nullptr;
    int cb_every = 1;
    std::condition_variable cv;
};

}  // namespace

struct vc_pool {
    vc_model* root = nullptr;
    int device = 0;
    hipStream_t st = nullptr;
    int R = VC_POOL_ROWS, capS = 0, out_stride = 0;   // R: rows of THIS pool (root->pool_rows when it was built)
    bool split = false;               // built for precision mode "split": fp32 KV, stacked hi / lo step operands
    int kv_es = 2;                    // bytes per cache element: 2 bf16, 1 e4m3 (fp8 weight format), 3 / 4 fp24 / fp32 (split)
    int split_G = 16;                 // rows per stacked hi / lo group: 16 (per-wave-ring GEMV: two weight passes per 32-row step)
                                      // or 32 (workgroup-shared GEMV: one)
    Buf kc, vc, rows, x_dec, xg_dec, qkv_dec, attn_dec, h_dec, logits, next_tok, out_ids, ssq, sk_scratch, sk_counters;
    hipGraphExec_t graph[VC_POOL_ROWS_MAX / 8] = {};    // one decode step over rows [0, 8 * (i + 1))
    std::mutex mu;
    std::condition_variable cv_driver, cv_rows;
    std::deque<PoolRequest*> pending;
    std::vector<PoolRequest*> active;
    bool used[VC_POOL_ROWS_MAX] = {};
    int users = 0;                    // generate() calls inside the pool (rows held or waited for)
    bool stop = false;
    std::thread driver;
    hipEvent_t step_ev[4] = {nullptr, nullptr, nullptr, nullptr};
    unsigned long steps_run = 0;
    unsigned long long steps_by_span[VC_POOL_ROWS_MAX / 8] = {};  // steps launched over 8 / 16 / 24 / 32 (... 64) rows (vc_pool_step_counts)
    // generate() calls that hold rows and have not yet handed their request to the driver (encode + prefill being enqueued): with
    // the hold policy (vc_pool_set_hold) the driver waits for them instead of stepping the rows it has — a step that runs beside a
    // prefill's GEMMs takes 2-7x its time (profiles/r05_d_decode_kernels_alone_vs_corun.md) and the joiner needs a full set of
    // steps of its own anyway, so nothing is gained by starting without it
    int prefilling = 0;
    // in-situ timing (vc_pool_profile): the step graphs were captured with stamp slots; prof_acc[span][kind] = {exec ticks,
    // period ticks, launches}
    bool prof = false;
    Buf stamps, prof_acc, stamp_scratch;
};

namespace {

LoopView pool_view(vc_pool* p) {
    LoopView v{};
    v.st = p->st;
    v.kc = p->kc.as<bf16_t>();
    v.vc = p->vc.as<bf16_t>();
    v.es = p->kv_es;
    // one layout whatever rows a step spans: a row keeps its slot between steps.  The workgroup-shared GEMV takes all 32 rows
    // (hi + lo planes) in ONE weight pass; the per-wave-ring form two passes of 16
    v.split_G = p->split ? p->split_G : 0;
    v.capR = p->R;
    v.capS = p->capS;
    v.rows = p->rows.as<int>();
    v.x_dec = p->x_dec.as<float>();
    v.xg_dec = p->xg_dec.as<bf16_t>();
    v.qkv_dec = p->qkv_dec.as<bf16_t>();
    v.attn_dec = p->attn_dec.as<bf16_t>();
    v.h_dec = p->h_dec.as<bf16_t>();
    v.logits = p->logits.as<float>();
    v.next_tok = p->next_tok.as<int>();
    v.out_ids = p->out_ids.as<int>();
    v.ssq = p->ssq.as<float>();
    v.sk_scratch = p->sk_scratch.as<float>();
    v.sk_counters = p->sk_counters.as<unsigned>();
    v.stamps = p->prof ? p->stamps.as<unsigned>() : nullptr;
    v.stamp_scratch = p->prof ? p->stamp_scratch.as<unsigned>() : nullptr;
    return v;
}

void pool_retire(vc_pool* p, PoolRequest* rq, bool failed, const std::string& err) {  // p->mu held
    rq->done = true;
    rq->failed = failed;
    rq->err = err;
    rq->avail = rq->produced;
    rq->cv.notify_all();
}

void pool_fail_all(vc_pool* p, const std::string& msg) {  // p->mu held
    for (PoolRequest* rq : p->active) pool_retire(p, rq, true, msg);
    for (PoolRequest* rq : p->pending) pool_retire(p, rq, true, msg);
    p->active.clear();
    p->pending.clear();
    p->stop = true;
}

void pool_driver(vc_pool* p) {
    (void)hipSetDevice(p->device);
    t_stream = p->st;
    const LoopView v = pool_view(p);
    vc_model* m = p->root;
    std::unique_lock<std::mutex> lk(p->mu);
    try {
        for (;;) {
            p->cv_driver.wait(lk, [&] { return p->stop || !p->pending.empty() || !p->active.empty(); });
            if (p->stop) break;
            // ---- admit: everything is enqueued on the pool stream between two steps
            while (!p->pending.empty()) {
                PoolRequest* rq = p->pending.front();
                p->pending.pop_front();
                HIPCHK(hipStreamWaitEvent(p->st, rq->prefill_done, 0));
                HIPCHK(hipMemcpyAsync(v.rows + (size_t)rq->row0 * RS_STRIDE, rq->rec, (size_t)rq->B * RS_STRIDE * 4,
                                      hipMemcpyHostToDevice, p->st));
                HIPCHK(hipEventRecord(rq->join_ev, p->st));
                SelectArgs sa = select_args(m, v, rq->sess->logits.as<float>(), rq->B, 1);  // token 0: step 0 -> 1, position stays
                sa.row0 = rq->row0;
                launch_select_embed(sa, p->st);
                rq->produced = 1;
                rq->steps_left = rq->g.max_new - 1;
                p->active.push_back(rq);
            }
            // ---- retire requests that need no (further) step, before and after stepping
            auto retire_finished = [&]() {
                for (size_t i = 0; i < p->active.size();) {
                    PoolRequest* rq = p->active[i];
                    bool over = rq->steps_left <= 0;
                    if (!over && rq->fin_pending && hipEventQuery(rq->fin_ev) == hipSuccess) {
                        rq->fin_pending = false;
                        bool all = true;
                        for (int b = 0; b < rq->B; ++b) all = all && rq->fin_host[b] != 0;
                        if (all) {  // every row had finished when the poll was taken: later columns are pads
                            rq->produced = std::min(rq->produced, std::max(rq->fin_at, 1));
                            over = true;
                        }
                    }
                    if (!over) { ++i; continue; }
                    // rows go inactive (RS_ACTIVE is word 0 of each record), stream-ordered after the request's last step
                    HIPCHK(hipMemset2DAsync(v.rows + (size_t)rq->row0 * RS_STRIDE, (size_t)RS_STRIDE * 4, 0, 4, rq->B, p->st));
                    HIPCHK(hipEventRecord(rq->done_ev, p->st));
                    p->active.erase(p->active.begin() + i);
                    pool_retire(p, rq, false, "");
                }
            };
            retire_finished();
            if (p->active.empty()) continue;
            if (p->root->pool_hold.load() && p->prefilling > 0) {
                // somebody is about to join: wait for its request (or for it to give up) rather than step without it
                // (a request cancelled by its caller — steps_left forced to 0, generate_on_pool's failure path — ends the wait too:
                // the loop's next pass retires it instead of keeping its caller blocked for somebody else's whole prefill)
                p->cv_driver.wait(lk, [&] {
                    bool cancelled = false;
                    for (PoolRequest* rq : p->active) cancelled = cancelled || rq->steps_left <= 0;
                    return p->stop || !p->pending.empty() || p->prefilling == 0 || !p->root->pool_hold.load() || cancelled;
                });
                continue;
            }
            int top = 0;
            for (PoolRequest* rq : p->active) top = std::max(top, rq->row0 + rq->B);
            const int gi = (top + 7) / 8 - 1;  // the step covers rows [0, top) rounded up to 8: free rows above cost nothing
            // bounded run-ahead: at most two steps queued beyond the one executing (a joining request waits that long)
            const unsigned long n = p->steps_run;
            if (n >= 2) {
                hipEvent_t e = p->step_ev[(n - 2) % 4];
                lk.unlock();
                HIPCHK(hipEventSynchronize(e));
                lk.lock();
            }
            HIPCHK(hipGraphLaunch(p->graph[gi], p->st));
            HIPCHK(hipEventRecord(p->step_ev[n % 4], p->st));
            p->steps_run = n + 1;
            p->steps_by_span[gi] += 1;
            for (PoolRequest* rq : p->active) {
                rq->steps_left -= 1;
                rq->produced += 1;
                // streamer: hand the request thread an event every cb_every columns (skipped while it is still busy)
                if (rq->cb && rq->report_taken && rq->produced - rq->avail >= rq->cb_every && rq->steps_left > 0) {
                    HIPCHK(hipEventRecord(rq->report_ev, p->st));
                    rq->avail = rq->produced;
                    rq->report_taken = false;
                    rq->cv.notify_all();
                }
                // EOS / stop: look at the rows' FINISHED words every 8 columns without stalling the stream
                if (rq->can_finish && !rq->fin_pending && rq->produced % 8 == 0 && rq->steps_left > 0) {
                    HIPCHK(hipMemcpy2DAsync(rq->fin_host, 4, v.rows + (size_t)rq->row0 * RS_STRIDE + RS_FINISHED,
                                            (size_t)RS_STRIDE * 4, 4, rq->B, hipMemcpyDeviceToHost, p->st));
                    HIPCHK(hipEventRecord(rq->fin_ev, p->st));
                    rq->fin_pending = true;
                    rq->fin_at = rq->produced;
                }
            }
            retire_finished();
        }
    } catch (const Fail& f) {
        pool_fail_all(p, f.msg);
    } catch (const std::exception& e) {  // a refused kernel launch
        pool_fail_all(p, e.what());
    }
}

void pool_destroy(vc_pool* p) {
    if (!p) return;
    {
        std::lock_guard<std::mutex> lk(p->mu);
        p->stop = true;
    }
    p->cv_driver.notify_all();
    if (p->driver.joinable()) p->driver.join();
    (void)hipSetDevice(p->device);
    if (p->st) (void)hipStreamSynchronize(p->st);
    for (auto& g : p->graph)
        if (g) (void)hipGraphExecDestroy(g);
    for (Buf* b : {&p->kc, &p->vc, &p->rows, &p->x_dec, &p->xg_dec, &p->qkv_dec, &p->attn_dec, &p->h_dec, &p->logits,
                   &p->next_tok, &p->out_ids, &p->ssq, &p->sk_scratch, &p->sk_counters, &p->stamps, &p->prof_acc, &p->stamp_scratch})
        b->release();
    for (auto& e : p->step_ev)
        if (e) (void)hipEventDestroy(e);
    if (p->st) (void)hipStreamDestroy(p->st);
    delete p;
}

std::mutex g_pool_create;

// rows per stacked hi / lo group of a split pool: 32 when the workgroup-shared GEMV (one weight pass over the 32 rows' two planes)
// can serve EVERY matrix of the model, else the per-wave-ring form's two passes of 16
int pool_split_G(const vc_model* root) {
    const vc_model_cfg& c = root->c;
    const bool fp8w = root->weight_format != 0;
    return (gemv_wg_enabled() && gemv_wg_applies(c.hidden, fp8w) && gemv_wg_applies(c.ffn, fp8w)) ? 32 : 16;
}

// the root model's pool with room for `need_S` positions and `need_out` ids per row; an idle pool that is too small (or that
// stopped after an error) is rebuilt, a busy one makes the caller wait for it to drain.  The pool is returned ACQUIRED:
// p->users was incremented while g_pool_create was still held, so no concurrent pool_for can find it idle and destroy it
// before the caller has registered (callers release with pool_release).
// whether a pool built now for `root` carries the in-situ timing slots: asked for, a bf16-step pool (a split step's GEMVs may take
// two passes: the slot layout assumes one — as do the two weight passes of a 64-row pool), and within the fold kernel's 512 slots.
// ONE definition for the rebuild test and the build (with the slot limit missing from the test, a > 102-layer model
// with profiling on would have rebuilt its pool on every generate()).
bool pool_wants_prof(const vc_model* root, bool want_split) {
    return root->pool_profile && !want_split && root->pool_rows <= VC_POOL_ROWS && 5 * root->c.layers + 1 <= 512;
}

vc_pool* pool_for(vc_model* m, int need_S, int need_out) {
    vc_model* root = m->root ? m->root : m;
    const vc_model_cfg& c = root->c;
    std::unique_lock<std::mutex> create(g_pool_create);
    vc_pool* p = root->pool;
    const bool want_split = m->precision == 2;
    if (p) {
        std::unique_lock<std::mutex> lk(p->mu);
        // (a split pool laid out for the other GEMV form — set_gemv_variant switched since it was built — is rebuilt as well)
        if (p->capS < need_S || p->out_stride < need_out || p->stop || p->split != want_split ||
            (want_split && p->split_G != pool_split_G(root)) || p->prof != pool_wants_prof(root, want_split) || p->R != root->pool_rows) {
            p->cv_rows.wait(lk, [&] { return p->users == 0; });
            lk.unlock();
            pool_destroy(p);
            root->pool = p = nullptr;
        } else {
            p->users += 1;
            return p;
        }
    }
    p = new vc_pool();
    try {
        p->root = root;
        p->device = root->ctx->device;
        p->split = want_split;
        p->split_G = pool_split_G(root);
        p->R = root->pool_rows;
        REQUIRE(!(want_split && p->R > VC_POOL_ROWS), VC_ERR_STATE, "a %d-row pool serves the bf16 step only", p->R);
        const int D = c.hidden, F = c.ffn, H = c.heads, R = p->R;
        const size_t es = want_split ? (size_t)split_kv_es() : (size_t)step_kv_es(root), two = want_split ? 2 : 1;
        p->kv_es = (int)es;
        p->capS = std::min((int)rup(std::max(need_S, 2048), 64), c.max_positions / 64 * 64);
        p->out_stride = std::max(need_out, p->capS);
        REQUIRE(p->capS >= need_S, VC_ERR_INVALID, "sequence %d exceeds max_position_embeddings=%d", need_S, c.max_positions);
        p->st = make_stream();
        t_stream = p->st;  // zero-fills of the new buffers
        const size_t kvb = (size_t)c.layers * R * H * p->capS * root->hd * es;
        p->kc.ensure(kvb, true);
        p->vc.ensure(kvb, true);
        p->rows.ensure((size_t)R * RS_STRIDE * 4, true);
        p->x_dec.ensure((size_t)R * D * 4, true);
        p->xg_dec.ensure(two * R * D * 2, true);
        p->qkv_dec.ensure((size_t)R * 3 * D * (want_split ? 4 : 2), true);   // the split step's projection rows are fp32
        p->attn_dec.ensure(two * R * D * 2, true);
        p->h_dec.ensure(two * R * F * 2, true);
        p->logits.ensure((size_t)R * c.vocab * 4, true);
        p->next_tok.ensure(R * 4, true);
        p->out_ids.ensure((size_t)R * p->out_stride * 4, true);
        p->ssq.ensure((size_t)R * root->npart * 4, true);
        p->sk_scratch.ensure(sk_floats(c) * 4);   // [K-slices][tiles][2 row groups][256]
        p->sk_counters.ensure((size_t)sk_counters_n(c) * 4, true);
        t_stream = m->st;
        for (auto& e : p->step_ev) HIPCHK(hipEventCreate(&e));
        // (a split step's GEMVs may take two passes: the slot layout assumes one; the fold kernel walks at most 512 slots)
        p->prof = pool_wants_prof(root, want_split);
        if (p->prof) {
            const size_t nslots = (size_t)5 * c.layers + 1;
            p->stamps.ensure(nslots * STAMP_SLOT_WORDS * 4, true);
            p->prof_acc.ensure((size_t)(R / 8) * PROF_KINDS * 3 * 8, true);
            p->stamp_scratch.ensure((1 + 2 * nslots) * 4, true);
            HIPCHK(hipStreamSynchronize(m->st));
        }
        LoopView v = pool_view(p);
        for (int i = 0; i < R / 8; ++i) {
            int slot = 0;
            if (p->prof) {
                v.stamp_next = &slot;
                v.prof_acc = p->prof_acc.as<unsigned long long>() + (size_t)i * PROF_KINDS * 3;
            }
            p->graph[i] = capture_step(root, v, 8 * (i + 1));
        }
        p->driver = std::thread(pool_driver, p);
    } catch (...) {  // a failed allocation / capture must not leak the half-built pool or leave t_stream on its stream
        t_stream = m->st;
        pool_destroy(p);
        throw;
    }
    p->users = 1;
    root->pool = p;
    return p;
}
void pool_release(vc_pool* p) {
    std::lock_guard<std::mutex> lk(p->mu);
    p->users -= 1;
    p->cv_rows.notify_all();
}

// generate() through the pool: prefill on the session's stream into pool rows, decode steps shared with whoever else is in
#define DBG_HIP(tag)                                                                                         \
    do {                                                                                                     \
        if (g_dbg_hip) {                                                                                     \
            hipError_t e_ = hipGetLastError();                                                               \
            if (e_ != hipSuccess) fprintf(stderr, "[vcoder_amd] last-error %d (%s) after %s\n", (int)e_, hipGetErrorString(e_), tag); \
        }                                                                                                    \
    } while (0)
const bool g_dbg_hip = getenv("VC_DEBUG_HIP") != nullptr;

void generate_on_pool(vc_model* m, const int64_t* ids, int B, int T, const float* img, const float* seg, const float* depth,
                      int on_dev, const GenParams& g, const std::vector<int>& tail, vc_token_cb cb, void* cb_user,
                      int cb_every, int32_t* out_ids, int* n_generated) {
    const vc_model_cfg& c = m->c;
    const int max_new = g.max_new;
    // the spliced length is only known after the splice plan; an upper bound sizes the pool: the text rows plus one
    // feature block per image the sample owns (vc_set_image_counts: the list / 5-D form gives a sample several images per
    // modality; otherwise one per modality), over the sample with the most images
    int n_img_max = 3;
    for (int b = 0; b < B; ++b) {
        int n_b = 0;
        for (int k = 0; k < 3; ++k) n_b += m->img_counts[k].empty() ? 1 : ((int)m->img_counts[k].size() == B ? m->img_counts[k][b] : 1);
        n_img_max = std::max(n_img_max, n_b);
    }
    const int S_bound = std::min(c.max_positions, T + n_img_max * m->Tv);
    vc_pool* p = pool_for(m, std::min(S_bound + max_new, c.max_positions / 64 * 64), max_new);  // acquired: users counted
    PoolRequest rq;
    rq.sess = m;
    rq.B = B;
    rq.g = g;
    rq.tail = tail;
    rq.cb = cb;
    rq.cb_every = std::max(cb_every, 1);
    rq.can_finish = g.eos >= 0 || g.n_stop > 0;
    // ---- rows: first fit of B contiguous free rows; blocks while the pool is full
    {
        std::unique_lock<std::mutex> lk(p->mu);
        int row0 = -1;
        p->cv_rows.wait(lk, [&] {
            for (int r0 = 0; r0 + B <= p->R; ++r0) {
                bool free_ = true;
                for (int r = r0; r < r0 + B && free_; ++r) free_ = !p->used[r];
                if (free_) { row0 = r0; return true; }
            }
            return false;
        });
        for (int r = row0; r < row0 + B; ++r) p->used[r] = true;
        rq.row0 = row0;
        p->prefilling += 1;
    }
    bool counted_prefilling = true;
    auto done_prefilling = [&]() {   // p->mu held
        if (!counted_prefilling) return;
        counted_prefilling = false;
        p->prefilling -= 1;
        p->cv_driver.notify_all();
    };
    auto release_rows = [&]() {
        std::lock_guard<std::mutex> lk(p->mu);
        for (int r = rq.row0; r < rq.row0 + B; ++r) p->used[r] = false;
        p->users -= 1;
        p->cv_rows.notify_all();
    };
    hipEvent_t* evs[] = {&rq.prefill_done, &rq.join_ev, &rq.done_ev, &rq.report_ev, &rq.fin_ev};
    auto cleanup = [&]() {
        for (hipEvent_t* e : evs)
            if (*e) (void)hipEventDestroy(*e);
        if (rq.rec) (void)hipHostFree(rq.rec);
        if (rq.fin_host) (void)hipHostFree(rq.fin_host);
    };
    try {
        for (hipEvent_t* e : evs) HIPCHK(hipEventCreate(e));
        HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&rq.rec), (size_t)B * RS_STRIDE * 4, 0));
        HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&rq.fin_host), (size_t)B * 4, 0));
        DBG_HIP("pool request setup");
        // ---- encode + prefill, keys / values straight into the pool's rows
        m->cur_pos = -1;
        int S = 0;
        std::unique_lock<std::mutex> gate(g_prefill_gate);
        do_prefill(m, ids, B, T, img, seg, depth, on_dev, 1, max_new, false, &S);
        DBG_HIP("do_prefill");
        m->last_S = S;
        REQUIRE(S + max_new <= p->capS, VC_ERR_INVALID,
                "prompt %d + max_new %d exceeds the context: max_position_embeddings=%d (KV capacity %d)", S, max_new,
                m->c.max_positions, p->capS);
        finish_prefill(m, KvTarget{p->kc.as<bf16_t>(), p->vc.as<bf16_t>(), p->R, p->capS, rq.row0, p->kv_es}, nullptr);
        if (m->ev[2]) HIPCHK(hipEventRecord(m->ev[2], m->st));
        HIPCHK(hipEventRecord(rq.prefill_done, m->st));
        DBG_HIP("finish_prefill");
        gate.unlock();
        fill_rows(rq.rec, B, g, S, tail.data(), rq.row0 * p->out_stride, p->out_stride);
        // ---- join, then sleep until the driver retires the request (streaming: wake per report)
        std::unique_lock<std::mutex> lk(p->mu);
        REQUIRE(!p->stop, VC_ERR_STATE, "the decode pool has stopped after an error");
        p->pending.push_back(&rq);
        done_prefilling();
        p->cv_driver.notify_all();
        int reported = 0;
        std::vector<int> part;
        auto report = [&](int upto, hipEvent_t after) {  // called with lk held; drops it around the copy + callback
            if (!cb || upto <= reported) return;
            lk.unlock();
            HIPCHK(hipStreamWaitEvent(m->st, after, 0));
            part.resize((size_t)B * (upto - reported));
            HIPCHK(hipMemcpy2DAsync(part.data(), (size_t)(upto - reported) * 4,
                                    p->out_ids.as<int>() + (size_t)rq.row0 * p->out_stride + reported,
                                    (size_t)p->out_stride * 4, (size_t)(upto - reported) * 4, B, hipMemcpyDeviceToHost, m->st));
            HIPCHK(hipStreamSynchronize(m->st));
            cb(cb_user, reported, upto - reported, B, part.data());
            reported = upto;
            lk.lock();
        };
        for (;;) {
            rq.cv.wait(lk, [&] { return rq.done || !rq.report_taken; });
            if (rq.done) break;
            const int upto = rq.avail;
            report(upto, rq.report_ev);
            rq.report_taken = true;
        }
        REQUIRE(!rq.failed, VC_ERR_HIP, "decode pool: %s", rq.err.c_str());
        int produced = rq.produced;
        lk.unlock();
        HIPCHK(hipStreamWaitEvent(m->st, rq.done_ev, 0));
        HIPCHK(hipMemcpy2DAsync(out_ids, (size_t)max_new * 4, p->out_ids.as<int>() + (size_t)rq.row0 * p->out_stride,
                                (size_t)p->out_stride * 4, (size_t)max_new * 4, B, hipMemcpyDeviceToHost, m->st));
        HIPCHK(hipStreamSynchronize(m->st));
        DBG_HIP("pool result copy");
        for (int b = 0; b < B; ++b)  // columns the loop never reached read as pad, like the session loop's pre-filled store
            for (int s_ = produced; s_ < max_new; ++s_) out_ids[(size_t)b * max_new + s_] = g.pad;
        produced = trim_columns(g, out_ids, max_new, tail.data(), B, produced);
        if (cb && produced > reported) {
            part.resize((size_t)B * (produced - reported));
            for (int b = 0; b < B; ++b)
                memcpy(part.data() + (size_t)b * (produced - reported), out_ids + (size_t)b * max_new + reported,
                       (size_t)(produced - reported) * 4);
            cb(cb_user, reported, produced - reported, B, part.data());
        }
        if (n_generated) *n_generated = produced;
        if (m->ev[0]) {
            (void)hipEventElapsedTime(&m->t_encode, m->ev[0], m->ev[1]);
            (void)hipEventElapsedTime(&m->t_prefill, m->ev[1], m->ev[2]);
            (void)hipEventElapsedTime(&m->t_decode, rq.join_ev, rq.done_ev);
        }
        DBG_HIP("pool timings");
    } catch (...) {
        {   // a request that is still queued / active must not outlive this frame
            std::unique_lock<std::mutex> lk(p->mu);
            done_prefilling();
            auto it = std::find(p->pending.begin(), p->pending.end(), &rq);
            if (it != p->pending.end()) p->pending.erase(it);
            else if (!rq.done && std::find(p->active.begin(), p->active.end(), &rq) != p->active.end()) {
                rq.steps_left = 0;  // the driver retires it at its next pass
                p->cv_driver.notify_all();
                rq.cv.wait(lk, [&] { return rq.done; });
            }
        }
        (void)hipStreamSynchronize(m->st);
        release_rows();
        cleanup();
        throw;
    }
    release_rows();
    cleanup();
}


// generate() on the session's own loop: prefill, then max_new - 1 graph-replayed (strict: eagerly enqueued) decode steps
void generate_on_session(vc_model* m, const int64_t* ids, int B, int T, const float* img, const float* seg, const float* depth,
                         int on_dev, const GenParams& g, const std::vector<int>& tail, vc_token_cb cb, void* cb_user,
                         int cb_every, int32_t* out_ids, int* n_generated) {
    const int max_new = g.max_new;
    m->cur_pos = -1;
    int S = 0;
    // Sessions of one process take turns in the MFMA-bound encode+prefill phase: two prefills side by side only slow
    // each other (and every decode in flight), while ONE prefill overlaps well with the HBM-bound decodes of the others.
    std::unique_lock<std::mutex> gate(g_prefill_gate);
    do_prefill(m, ids, B, T, img, seg, depth, on_dev, 1, max_new, true, &S);  // generate() always builds a mask
    m->last_S = S;
    REQUIRE(S + max_new <= m->capS, VC_ERR_INVALID, "prompt %d + max_new %d exceeds the KV capacity %d", S, max_new, m->capS);
    ensure_out_ids(m, B, max_new);
    finish_prefill(m, session_kv(m), nullptr);
    m->kmask_in_decode = false;   // generate(): the cached steps run under an all-ones mask (vcoder_ds_llava_arch.py:130-133)
    if (m->ev[2]) HIPCHK(hipEventRecord(m->ev[2], m->st));
    arm_session_rows(m, g, tail.data());
    const LoopView v = session_view(m);
    // token 0 comes from the prefill logits
    std::vector<int> fill((size_t)B * m->out_stride, g.pad);
    HIPCHK(hipMemcpyAsync(m->out_ids.p, fill.data(), fill.size() * 4, hipMemcpyHostToDevice, m->st));
    launch_select_embed(select_args(m, v, v.logits, B, 1), m->st);  // step 0 -> 1; the position stays at S
    HIPCHK(hipStreamSynchronize(m->st));
    gate.unlock();
    int produced = 1, reported = 0;
    const bool can_finish = g.eos >= 0 || g.n_stop > 0;
    std::vector<int> rec((size_t)B * RS_STRIDE), part;
    auto all_finished = [&]() {
        if (!can_finish) return false;
        HIPCHK(hipMemcpyAsync(rec.data(), m->rows.p, rec.size() * 4, hipMemcpyDeviceToHost, m->st));
        HIPCHK(hipStreamSynchronize(m->st));
        for (int b = 0; b < B; ++b)
            if (!rec[(size_t)b * RS_STRIDE + RS_FINISHED]) return false;
        return true;
    };
    auto report = [&](int upto) {  // streamer callback: columns [reported, upto) of every row
        if (!cb || upto <= reported) return;
        part.resize((size_t)B * (upto - reported));
        HIPCHK(hipMemcpy2DAsync(part.data(), (size_t)(upto - reported) * 4, m->out_ids.as<int>() + reported,
                                (size_t)m->out_stride * 4, (size_t)(upto - reported) * 4, B, hipMemcpyDeviceToHost, m->st));
        HIPCHK(hipStreamSynchronize(m->st));
        cb(cb_user, reported, upto - reported, B, part.data());
        reported = upto;
    };
    const int every = cb ? std::max(cb_every, 1) : 8;
    if (cb && every == 1) report(1);
    if (max_new > 1 && !all_finished()) {
        if (m->precision != 1) ensure_graph(m, B);
        for (int step = 1; step < max_new; ++step) {
            if (m->precision == 1) enqueue_decode_step_strict(m, B);
            else HIPCHK(hipGraphLaunch(m->graph, m->st));
            m->cur_pos += 1;
            produced = step + 1;
            // the reference checks its stopping criteria on the host every token; checking every few tokens only
            // trims later (rows past EOS already emit pad), it never changes the returned ids
            if ((step + 1) % every == 0 || step == max_new - 1) {
                if (cb) report(produced);
                if (can_finish) {
                    HIPCHK(hipStreamSynchronize(m->st));
                    if (all_finished()) break;
                }
            }
        }
    }
    if (m->ev[3]) HIPCHK(hipEventRecord(m->ev[3], m->st));
    HIPCHK(hipMemcpy2DAsync(out_ids, (size_t)max_new * 4, m->out_ids.p, (size_t)m->out_stride * 4, (size_t)max_new * 4, B,
                            hipMemcpyDeviceToHost, m->st));
    HIPCHK(hipStreamSynchronize(m->st));
    produced = trim_columns(g, out_ids, max_new, tail.data(), B, produced);
    if (cb) report(produced);
    if (n_generated) *n_generated = produced;
    if (m->ev[0]) {
        (void)hipEventElapsedTime(&m->t_encode, m->ev[0], m->ev[1]);
        (void)hipEventElapsedTime(&m->t_prefill, m->ev[1], m->ev[2]);
        (void)hipEventElapsedTime(&m->t_decode, m->ev[2], m->ev[3]);
    }
}

}  // namespace

/* generate(): encode + splice + prefill + (max_new - 1) decode steps with the token selection on the device — greedy
 * (samp NULL or do_sample 0) or temperature / top-k / top-p sampling — EOS / pad bookkeeping, device-side keyword stops
 * and an optional streamer callback.  See include/vcoder_hip.h. */
VC_API int vc_generate(vc_model* m, const int64_t* ids, int B, int T, const float* img, const float* seg, const float* depth,
                       int pixels_on_device, int max_new, int eos_id, int pad_id, const int32_t* stop_ids,
                       const int32_t* stop_lens, int n_stop, const vc_sampling* samp, vc_token_cb cb, void* cb_user,
                       int cb_every, int32_t* out_ids, int* n_generated) {
    if (!m) return VC_ERR_INVALID;
    OneShotReset one_shot{m};
    GUARD_BEGIN
    USE_DEVICE(m->ctx);
    REQUIRE(max_new >= 1 && out_ids, VC_ERR_INVALID, "bad max_new/out_ids");
    REQUIRE(n_stop >= 0 && n_stop <= VC_MAX_STOP && (n_stop == 0 || (stop_ids && stop_lens)), VC_ERR_INVALID,
            "at most %d stop sequences", VC_MAX_STOP);
    // finished rows are fed the pad token's embedding: it must be a real row of embed_tokens
    REQUIRE(!(eos_id >= 0 || n_stop > 0) || (pad_id >= 0 && pad_id < m->c.vocab), VC_ERR_INDEX,
            "pad_token_id %d is outside the vocabulary (%d)", pad_id, m->c.vocab);
    GenParams g;
    g.max_new = max_new;
    g.eos = eos_id;
    g.pad = pad_id;
    g.n_stop = n_stop;
    for (int i = 0, off = 0; i < n_stop; ++i) {
        REQUIRE(stop_lens[i] >= 1 && stop_lens[i] <= VC_MAX_STOP_LEN, VC_ERR_INVALID, "stop sequence %d: 1..%d ids", i,
                VC_MAX_STOP_LEN);
        g.stop[i][0] = stop_lens[i];
        for (int j = 0; j < stop_lens[i]; ++j) g.stop[i][1 + j] = stop_ids[off + j];
        off += stop_lens[i];
    }
    if (samp && samp->do_sample) {
        REQUIRE(samp->temperature > 0.f, VC_ERR_INVALID, "temperature must be positive (got %g)", (double)samp->temperature);
        REQUIRE(samp->top_p > 0.f && samp->top_p <= 1.f, VC_ERR_INVALID, "top_p must be in (0, 1] (got %g)", (double)samp->top_p);
        g.do_sample = 1;
        g.temperature = samp->temperature;
        g.top_k = samp->top_k;
        g.top_p = samp->top_p;
        g.seed = samp->seed;
    }
    constexpr int TL = VC_MAX_STOP_LEN - 1;
    std::vector<int> tail((size_t)B * TL, INT32_MIN);  // ids never equal INT32_MIN
    for (int b = 0; b < B; ++b)
        for (int j = 0; j < TL && j < T; ++j) tail[(size_t)b * TL + TL - 1 - j] = (int)ids[(size_t)b * T + T - 1 - j];
    // concurrent generate() calls share their decode steps in the root model's pool (VC_POOL=0: every call on its own
    // loop); strict mode keeps fp32 caches of its own
    static const bool use_pool = !(getenv("VC_POOL") && atoi(getenv("VC_POOL")) == 0);
    if (use_pool && m->precision != 1)
        generate_on_pool(m, ids, B, T, img, seg, depth, pixels_on_device, g, tail, cb, cb_user, cb_every, out_ids, n_generated);
    else
        generate_on_session(m, ids, B, T, img, seg, depth, pixels_on_device, g, tail, cb, cb_user, cb_every, out_ids,
                            n_generated);
    GUARD_END(m->ctx)
}

VC_API int vc_generate_greedy_stop(vc_model* m, const int64_t* ids, int B, int T, const float* img, const float* seg,
                                   const float* depth, int pixels_on_device, int max_new, int eos_id, int pad_id,
                                   const int32_t* stop_ids, const int32_t* stop_lens, int n_stop, int32_t* out_ids,
                                   int* n_generated) {
    return vc_generate(m, ids, B, T, img, seg, depth, pixels_on_device, max_new, eos_id, pad_id, stop_ids, stop_lens, n_stop,
                       nullptr, nullptr, nullptr, 0, out_ids, n_generated);
}

/* spliced sequence length (text + feature rows) of the last vc_generate_greedy* call of this model / session */
VC_API int vc_last_spliced_len(vc_model* m) { return m ? m->last_S : VC_ERR_INVALID; }

VC_API int vc_generate_greedy(vc_model* m, const int64_t* ids, int B, int T, const float* img, const float* seg,
                              const float* depth, int pixels_on_device, int max_new, int eos_id, int pad_id,
                              int32_t* out_ids, int* n_generated) {
    return vc_generate_greedy_stop(m, ids, B, T, img, seg, depth, pixels_on_device, max_new, eos_id, pad_id, nullptr, nullptr,
                                   0, out_ids, n_generated);
}

// ---- image preprocessing: PIL's 8-bit bicubic coefficient tables (Pillow Resample.c: precompute_coeffs +
// normalize_coeffs_8bpc), built in double on the host ----------------------------------------------------------------
namespace {
double bicubic_filter(double x) {
    const double a = -0.5;
    if (x < 0.0) x = -x;
    if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1;
    if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * a;
    return 0.0;
}
// bounds[2*o] = first tap, bounds[2*o+1] = tap count; kk[o*ksize + t] = fixed-point weight
int resample_coeffs(int in_size, int out_size, std::vector<int>& bounds, std::vector<int>& kk) {
    const double scale = (double)in_size / out_size;
    const double fs = scale < 1.0 ? 1.0 : scale;
    const double support = 2.0 * fs;
    const int ksize = (int)ceil(support) * 2 + 1;
    bounds.assign((size_t)out_size * 2, 0);
    kk.assign((size_t)out_size * ksize, 0);
    std::vector<double> w(ksize);
    for (int xx = 0; xx < out_size; ++xx) {
        const double center = (xx + 0.5) * scale, ss = 1.0 / fs;
        int xmin = (int)(center - support + 0.5);
        if (xmin < 0) xmin = 0;
        int xmax = (int)(center + support + 0.5);
        if (xmax > in_size) xmax = in_size;
        const int n = xmax - xmin;
        double tot = 0.0;
        for (int x = 0; x < n; ++x) {
            w[x] = bicubic_filter((x + xmin - center + 0.5) * ss);
            tot += w[x];
        }
        for (int x = 0; x < n; ++x) {
            const double v = tot != 0.0 ? w[x] / tot : w[x];
            kk[(size_t)xx * ksize + x] = v < 0 ? (int)(-0.5 + v * (double)(1 << 22)) : (int)(0.5 + v * (double)(1 << 22));
        }
        bounds[2 * xx] = xmin;
        bounds[2 * xx + 1] = n;
    }
    return ksize;
}
}  // namespace

/* One image: uint8 RGB [h,w,3] (host) -> fp32 [3,S,S] CLIP-normalised pixels (device when out_on_device, else host).
 * pad_to_square = the reference's image_aspect_ratio == 'pad' path (mm_utils.py:31-35): expand2square with the mean
 * colour, then resize to S x S; otherwise resize the shortest edge to S (bicubic) and center-crop S x S. */
VC_API int vc_preprocess_image(vc_model* m, const uint8_t* rgb, int h, int w, int pad_to_square, const float* mean,
                               const float* stdv, float* out, int out_on_device) {
    if (!m) return VC_ERR_INVALID;
    GUARD_BEGIN
    USE_DEVICE(m->ctx);
    REQUIRE(rgb && out && mean && stdv && h > 0 && w > 0, VC_ERR_INVALID, "bad preprocess arguments");
    const int S = m->c.vit_image;
    m->pp_src.ensure((size_t)h * w * 3);
    HIPCHK(hipMemcpyAsync(m->pp_src.p, rgb, (size_t)h * w * 3, hipMemcpyHostToDevice, m->st));
    const uint8_t* cur = m->pp_src.as<uint8_t>();
    int ch = h, cw = w;
    if (pad_to_square && h != w) {
        const int side = h > w ? h : w;
        const int fill[3] = {(int)(mean[0] * 255), (int)(mean[1] * 255), (int)(mean[2] * 255)};  // int(x*255), mm_utils.py:33
        m->pp_sq.ensure((size_t)side * side * 3);
        launch_pad_square(cur, h, w, m->pp_sq.as<uint8_t>(), side, (side - w) / 2, (side - h) / 2, fill, m->st);
        cur = m->pp_sq.as<uint8_t>();
        ch = cw = side;
    }
    // shortest edge -> S, long edge = int(S * long / short)   ([HF] get_resize_output_image_size, default_to_square=False)
    int nh, nw;
    if (ch <= cw) { nh = S; nw = (int)((double)S * cw / ch); }
    else { nw = S; nh = (int)((double)S * ch / cw); }
    if (nh != ch || nw != cw) {
        std::vector<int> bh, kh, bv, kv;
        const int ksh = resample_coeffs(cw, nw, bh, kh), ksv = resample_coeffs(ch, nh, bv, kv);
        const size_t tab = bh.size() + kh.size() + bv.size() + kv.size();
        m->pp_tab.ensure(tab * 4);
        int* t = m->pp_tab.as<int>();
        int *d_bh = t, *d_kh = d_bh + bh.size(), *d_bv = d_kh + kh.size(), *d_kv = d_bv + bv.size();
        HIPCHK(hipMemcpyAsync(d_bh, bh.data(), bh.size() * 4, hipMemcpyHostToDevice, m->st));
        HIPCHK(hipMemcpyAsync(d_kh, kh.data(), kh.size() * 4, hipMemcpyHostToDevice, m->st));
        HIPCHK(hipMemcpyAsync(d_bv, bv.data(), bv.size() * 4, hipMemcpyHostToDevice, m->st));
        HIPCHK(hipMemcpyAsync(d_kv, kv.data(), kv.size() * 4, hipMemcpyHostToDevice, m->st));
        m->pp_tmp.ensure((size_t)ch * nw * 3);
        m->pp_out.ensure((size_t)nh * nw * 3);
        // PIL skips a pass whose size does not change, and runs horizontal first
        const uint8_t* src = cur;
        int th = ch;
        if (nw != cw) {
            launch_resample(src, ch, cw, m->pp_tmp.as<uint8_t>(), ch, nw, d_bh, d_kh, ksh, 1, m->st);
            src = m->pp_tmp.as<uint8_t>();
        }
        if (nh != ch) {
            launch_resample(src, th, nw, m->pp_out.as<uint8_t>(), nh, nw, d_bv, d_kv, ksv, 0, m->st);
            src = m->pp_out.as<uint8_t>();
        }
        cur = src;
        HIPCHK(hipStreamSynchronize(m->st));  // the host coefficient vectors go out of scope
    }
    const int top = (nh - S) / 2, left = (nw - S) / 2;
    float* dst = out;
    if (!out_on_device) {
        m->pp_f32.ensure((size_t)3 * S * S * 4);
        dst = m->pp_f32.as<float>();
    }
    launch_crop_normalize(cur, nh, nw, top, left, dst, S, mean, stdv, m->st);
    if (!out_on_device)
        HIPCHK(hipMemcpyAsync(out, dst, (size_t)3 * S * S * 4, hipMemcpyDeviceToHost, m->st));
    HIPCHK(hipStreamSynchronize(m->st));
    GUARD_END(m->ctx)
}

VC_API int vc_last_timings(vc_model* m, float* encode_ms, float* prefill_ms, float* decode_ms) {
    if (!m) return VC_ERR_INVALID;
    if (encode_ms) *encode_ms = m->t_encode;
    if (prefill_ms) *prefill_ms = m->t_prefill;
    if (decode_ms) *decode_ms = m->t_decode;
    return VC_OK;
}

/* times `reps` sweeps of the decode attention launches of one step (one per layer) over `B` rows at context `ctx` (keys per
 * row before the append; row b sits at ctx - (7 b) % 64 so the rows differ like concurrent requests do) with HIP events on the
 * model's stream; the KV contents are whatever the cache holds (the kernel's speed does not depend on the values).  Returns
 * launches per sweep, average microseconds per launch and the algorithmic KV bytes per launch (4 * sum(ctx_b + 1) * hidden). */
VC_API int vc_profile_decode_attention(vc_model* m, int B, int ctx, int reps, int* launches, double* avg_us, double* avg_bytes) {
    if (!m) return VC_ERR_INVALID;
    GUARD_BEGIN
    USE_DEVICE(m->ctx);
    REQUIRE(m->finalized && B >= 1 && B <= VC_POOL_ROWS && reps >= 1 && ctx >= 64, VC_ERR_INVALID, "bad profile arguments");
    REQUIRE(m->precision == 0, VC_ERR_STATE, "the profile hooks time the bf16 path's kernels");
    const vc_model_cfg& c = m->c;
    LoopView v;
    if (B <= VC_MAX_ROWS) {
        ensure_llm(m, B, ctx + 64);
        v = session_view(m);
    } else {
        vc_pool* p = pool_for(m, ctx + 64, 1);
        pool_release(p);  // measurement hook: the caller guarantees that no generate() runs meanwhile
        {
            std::lock_guard<std::mutex> lk(p->mu);
            REQUIRE(p->users == 0 && p->active.empty() && p->pending.empty(), VC_ERR_STATE, "the decode pool is busy");
        }
        HIPCHK(hipStreamSynchronize(p->st));
        v = pool_view(p);
        v.st = m->st;
    }
    REQUIRE(ctx + 1 <= v.capS, VC_ERR_INVALID, "context %d exceeds the cache capacity %d", ctx, v.capS);
    std::vector<int> rec((size_t)B * RS_STRIDE, 0);
    double keys = 0;
    for (int b = 0; b < B; ++b) {
        rec[(size_t)b * RS_STRIDE + RS_ACTIVE] = 1;
        rec[(size_t)b * RS_STRIDE + RS_POS] = ctx - (7 * b) % 64;
        keys += rec[(size_t)b * RS_STRIDE + RS_POS] + 1;
    }
    HIPCHK(hipMemcpyAsync(v.rows, rec.data(), rec.size() * 4, hipMemcpyHostToDevice, m->st));
    HIPCHK(hipStreamSynchronize(m->st));
    auto sweep = [&]() {
        for (int l = 0; l < c.layers; ++l) {
            AttnDecodeFusedArgs da{v.qkv_dec, kcache(v, m, l), vcache(v, m, l), v.attn_dec, B, c.heads, m->hd, v.capS,
                                   v.rows + RS_POS, m->rope_cos, m->rope_sin, 1.0f / sqrtf((float)m->hd), RS_STRIDE,
                                   v.rows + RS_ACTIVE, v.es == 1 ? 3 : 0};
            launch_attention_decode_fused(da, m->st);
        }
    };
    sweep();  // warm
    HIPCHK(hipEventRecord(m->ev[0], m->st));
    for (int r = 0; r < reps; ++r) sweep();
    HIPCHK(hipEventRecord(m->ev[1], m->st));
    HIPCHK(hipStreamSynchronize(m->st));
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, m->ev[0], m->ev[1]));
    if (launches) *launches = c.layers;
    if (avg_us) *avg_us = (double)ms * 1e3 / ((double)reps * c.layers);
    if (avg_bytes) *avg_bytes = 2.0 * (double)v.es * keys * (double)c.hidden;   // K + V rows of `es` bytes per element
    HIPCHK(hipMemsetAsync(v.rows, 0, (size_t)B * RS_STRIDE * 4, m->st));
    HIPCHK(hipStreamSynchronize(m->st));
    m->cur_pos = -1;
    GUARD_END(m->ctx)
}

/* cumulative number of pooled decode steps launched over 8, 16, 24 and 32 rows since the pool of m's root model was
 * created (zeros if it has none): lets bench.py weight the per-row-count kernel timings by what the timed run executed */
VC_API int vc_pool_step_counts(vc_model* m, unsigned long long* counts4) {
    if (!m || !counts4) return VC_ERR_INVALID;
    vc_model* root = m->root ? m->root : m;
    for (int i = 0; i < 4; ++i) counts4[i] = 0;
    if (root->pool) {
        std::lock_guard<std::mutex> lk(root->pool->mu);
        for (int i = 0; i < 4; ++i) counts4[i] = root->pool->steps_by_span[i];
        // (a 64-row pool, vc_pool_set_rows: its steps over 40 .. 64 rows are reported with the 32-row count — two weight passes each)
        for (int i = 4; i < VC_POOL_ROWS_MAX / 8; ++i) counts4[3] += root->pool->steps_by_span[i];
    }
    return VC_OK;
}

/* In-situ timing of the pool's decode-step kernels.  on != 0: the pool's step graphs are (re)captured with a timing slot per launch
 * — the first thread of every workgroup stamps {earliest start, latest end} with the device's constant-rate wall clock — and one
 * tiny launch per step folds the slots into per-(span, kind) sums, so that a measurement covers every launch of a timed region as
 * it ran there (beside whatever the other sessions had on the GPU), not a replay.  Takes effect when the pool is next (re)built,
 * i.e. while it is idle.  bf16 path only (the split step keeps its graphs). */
/* Pool scheduling policy.  on = 1 (default): the pool does not launch a decode step while a generate() call that already holds
 * rows is still in its encode / prefill phase — it waits for that request to join (a step beside a prefill's GEMMs runs at a
 * fraction of its speed, and the joiner needs a full set of steps of its own anyway).  Throughput policy: the requests already
 * decoding see one pause of about a prefill when somebody joins.  on = 0: step whatever rows are active (lowest inter-token
 * latency for the requests in flight).  Applies to every session of the model. */
VC_API int vc_pool_set_hold(vc_model* m, int on) {
    if (!m) return VC_ERR_INVALID;
    vc_model* root = m->root ? m->root : m;
    root->pool_hold.store(on != 0);
    // the pool may be rebuilt (destroyed and re-created) by pool_for under g_pool_create while this runs on another thread
    std::lock_guard<std::mutex> create_lk(g_pool_create);
    if (root->pool) {
        std::lock_guard<std::mutex> lk(root->pool->mu);
        root->pool->cv_driver.notify_all();
    }
    return VC_OK;
}

/* Rows of the decode pool: 32 (default; one weight pass per step: two MFMA token-slot groups) or 64 (measurement, round 6: a step
 * then takes TWO 32-row weight passes, i.e. what two 32-row pools would stream, plus the attention of 64 rows).  Takes effect when
 * the pool is next (re)built, i.e. while it is idle; a 64-row pool serves the bf16 step only and carries no in-situ timing slots. */
VC_API int vc_pool_set_rows(vc_model* m, int rows) {
    if (!m || (rows != VC_POOL_ROWS && rows != VC_POOL_ROWS_MAX)) return VC_ERR_INVALID;
    (m->root ? m->root : m)->pool_rows = rows;
    return VC_OK;
}

VC_API int vc_pool_profile(vc_model* m, int on) {
    if (!m) return VC_ERR_INVALID;
    vc_model* root = m->root ? m->root : m;
    root->pool_profile = on != 0;
    return VC_OK;
}

/* sums since the last reset, for span s (8 (s + 1) rows) and kind k (0 qkv, 1 decode attention, 2 o_proj, 3 gate/up, 4 down,
 * 5 lm_head), each [4][6]: exec_us = earliest workgroup start -> latest workgroup end of the launches; period_us = latest end of
 * the previous launch of the step -> latest end of this one (dispatch, drain and the inter-kernel gap included: what the step's
 * dependency chain pays per launch); launches.  reset != 0 zeroes the sums.  The pool must be idle (no generate() in flight). */
VC_API int vc_pool_profile_read(vc_model* m, double* exec_us, double* period_us, unsigned long long* launches, int reset) {
    if (!m || !exec_us || !period_us || !launches) return VC_ERR_INVALID;
    GUARD_BEGIN
    USE_DEVICE(m->ctx);
    vc_model* root = m->root ? m->root : m;
    const int nspan = VC_POOL_ROWS / 8;
    for (int i = 0; i < nspan * PROF_KINDS; ++i) {
        exec_us[i] = period_us[i] = 0;
        launches[i] = 0;
    }
    vc_pool* p = root->pool;
    if (p && p->prof) {
        {
            std::lock_guard<std::mutex> lk(p->mu);
            REQUIRE(p->users == 0 && p->active.empty() && p->pending.empty(), VC_ERR_STATE, "the decode pool is busy");
        }
        HIPCHK(hipStreamSynchronize(p->st));
        std::vector<unsigned long long> h((size_t)nspan * PROF_KINDS * 3);
        HIPCHK(hipMemcpy(h.data(), p->prof_acc.p, h.size() * 8, hipMemcpyDeviceToHost));
        int khz = 100000;   // the constant-rate wall clock: 100 MHz on gfx9
#ifndef VC_EMU
        (void)hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, root->ctx->device);
        if (khz <= 0) khz = 100000;
#endif
        for (int i = 0; i < nspan * PROF_KINDS; ++i) {
            exec_us[i] = (double)h[3 * i] * 1e3 / (double)khz;
            period_us[i] = (double)h[3 * i + 1] * 1e3 / (double)khz;
            launches[i] = h[3 * i + 2];
        }
        if (reset) HIPCHK(hipMemset(p->prof_acc.p, 0, h.size() * 8));
    }
    GUARD_END(m->ctx)
}

VC_API int vc_profile_decode_gemv(vc_model* m, int B, int reps, int* launches, double* avg_us, double* avg_bytes) {
    if (!m) return VC_ERR_INVALID;
    GUARD_BEGIN
    USE_DEVICE(m->ctx);
    REQUIRE(m->finalized && B >= 1 && B <= VC_POOL_ROWS && reps >= 1, VC_ERR_INVALID, "bad profile arguments");
    REQUIRE(m->precision == 0, VC_ERR_STATE, "the profile hooks time the bf16 path's kernels");
    const vc_model_cfg& c = m->c;
    const int D = c.hidden, F = c.ffn;
    LoopView v;
    if (B <= VC_MAX_ROWS) {
        ensure_llm(m, B, 64);
        v = session_view(m);
    } else {  // 17..32 rows: the decode pool's buffers (it must be idle), launches on this session's stream
        vc_pool* p = pool_for(m, 64, 1);
        pool_release(p);  // measurement hook: the caller guarantees that no generate() runs meanwhile
        {
            std::lock_guard<std::mutex> lk(p->mu);
            REQUIRE(p->users == 0 && p->active.empty() && p->pending.empty(), VC_ERR_STATE, "the decode pool is busy");
        }
        HIPCHK(hipStreamSynchronize(p->st));
        v = pool_view(p);
        v.st = m->st;
    }
    auto sweep = [&]() { decode_linears(m, v, B, [](int) {}); };
    sweep();  // warm
    HIPCHK(hipEventRecord(m->ev[0], m->st));
    for (int r = 0; r < reps; ++r) sweep();
    HIPCHK(hipEventRecord(m->ev[1], m->st));
    HIPCHK(hipStreamSynchronize(m->st));
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, m->ev[0], m->ev[1]));
    const int n = 4 * c.layers + 1;
    const double wb = m->weight_format >= 1 ? 1.0 : 2.0;  // bytes per decoder-linear weight (lm_head stays bf16)
    const double bytes = wb * (double)c.layers * (4.0 * D * D + 3.0 * D * F) + 2.0 * (double)D * c.vocab;
    if (launches) *launches = n;
    if (avg_us) *avg_us = (double)ms * 1e3 / ((double)reps * n);
    if (avg_bytes) *avg_bytes = bytes / n;
    HIPCHK(hipMemsetAsync(v.x_dec, 0, (size_t)rup(B, 16) * D * 4, m->st));
    HIPCHK(hipStreamSynchronize(m->st));
    GUARD_END(m->ctx)
}
