/* vcoder_hip.h — C ABI of libvcoder_hip.so: the MI355X-native VCoder inference hot path.
 *
 * The reference (SHI-Labs/VCoder) has no FFI: its boundary is the Python model API
 *   load_pretrained_model()                     vcoder_llava/model/builder.py:25-154
 *   VCoder[DS]LlavaLlamaForCausalLM.forward()   vcoder_llava/model/language_model/vcoder_ds_llava_llama.py:57-118
 *   .generate() (HF GenerationMixin)            called at vcoder_llava/serve/cli.py:122-132
 * This library owns all device work behind that API; vcoder_amd/ (Python, ctypes) re-creates the
 * reference's import surface on top of it.  Plain C: opaque handles, pointers and sizes,
 * no C++ types, no exceptions across the boundary.
 *
 * Ownership: the caller owns every input/output buffer; the library owns weights, KV cache, workspaces,
 * streams and hipGraph objects.  `vc_model_load_tensor` copies (the caller may free immediately).
 * Threading: one vc_model = one device + one stream; not thread-safe (matches the reference's callers).
 * Errors: every call returns VC_OK (0) or a negative vc_status; vc_last_error() gives the message.
 */
#ifndef VCODER_HIP_H
#define VCODER_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct vc_ctx vc_ctx;
typedef struct vc_model vc_model;

typedef enum {
    VC_OK = 0,
    VC_IGNORED = 1,          /* load_tensor: key accepted but dead at inference */
    VC_ERR_INVALID = -1,     /* bad argument / shape / unknown key            -> ValueError   */
    VC_ERR_HIP = -2,         /* HIP runtime failure                           -> RuntimeError */
    VC_ERR_STATE = -3,       /* call order violation (e.g. decode before prefill) -> RuntimeError */
    VC_ERR_INDEX = -4,       /* a placeholder id reached the embedding lookup -> IndexError (vcoder_llava_arch.py:187 quirk) */
    VC_ERR_UNEQUAL = -5      /* unequal spliced lengths with attention_mask   -> UnboundLocalError (vcoder_ds_llava_arch.py:295-297) */
} vc_status;

/* which reference class: llava_arch.py / vcoder_llava_arch.py / vcoder_ds_llava_arch.py */
typedef enum { VC_VARIANT_LLAVA = 0, VC_VARIANT_VCODER = 1, VC_VARIANT_VCODER_DS = 2 } vc_variant;
typedef enum { VC_F32 = 0, VC_BF16 = 1 } vc_dtype;
typedef enum { VC_MOD_IMAGE = 0, VC_MOD_SEG = 1, VC_MOD_DEPTH = 2 } vc_modality;

/* Mirrors the config keys the reference constructors read. */
typedef struct vc_model_cfg {
    int32_t variant;            /* vc_variant */
    /* CLIP ViT (multimodal_encoder/clip_encoder.py) */
    int32_t vit_hidden, vit_heads, vit_ffn, vit_layers, vit_layers_used, vit_image, vit_patch;
    int32_t vit_keep_cls;       /* mm_vision_select_feature == 'cls_patch' */
    float vit_ln_eps;
    /* Llama decoder */
    int32_t hidden, heads, ffn, layers, vocab, max_positions;
    float rms_eps, rope_theta;
    /* adapters: 0 = identity, 1 = linear, N = mlpNx_gelu  (multimodal_projector/builder.py:33-51) */
    int32_t mm_proj_depth, seg_proj_depth;
    int32_t pad_token_id;
} vc_model_cfg;

/* ---- lifetime ------------------------------------------------------------------------------ */
int vc_init(int device_id, vc_ctx** out);
void vc_shutdown(vc_ctx* ctx);
const char* vc_last_error(vc_ctx* ctx);
int vc_synchronize(vc_ctx* ctx);
void* vc_stream(vc_ctx* ctx);                 /* the hipStream_t all work of this context is enqueued on */

int vc_model_create(vc_ctx* ctx, const vc_model_cfg* cfg, vc_model** out);
void vc_model_destroy(vc_model* m);
/* another session (own stream = `ctx`, own KV cache / workspaces / graph) on the finalized weights of `parent`,
 * shared read-only; `parent` must outlive it.  Sessions on different contexts may run concurrently. */
int vc_model_create_shared(vc_ctx* ctx, vc_model* parent, vc_model** out);

/* ---- weights: replaces HF from_pretrained() of builder.py:93-108 + CLIPVisionTower.load_model() ---- */
/* hf_key = state-dict key of the reference checkpoint; host_ptr = row-major tensor of `dtype`; the full shape is checked
 * (a transposed matrix is an error, not a silent load). */
int vc_model_load_tensor(vc_model* m, const char* hf_key, const void* host_ptr, int dtype, const int64_t* shape,
                         int ndim);
/* device-side deterministic generator, bit-identical to vcoder_amd/synth.py:synth_tensor (benchmarks, tests) */
int vc_model_synth_tensor(vc_model* m, const char* hf_key, const int64_t* shape, int ndim, uint32_t tensor_seed,
                          float offset, float halfwidth);
/* after the last tensor: fuses QKV / interleaves gate-up / packs decode copies; fails listing a missing key */
/* on = 1: a sample's results do not depend on the batch (or the rank's shard) it runs in — the prefill GEMMs skip their split-K
 * remainder round, whose slicing follows the tile count; everything else already is batch-invariant.  The reference's rows are
 * independent of the batch size; with it the ids of N ranks x B equal those of one rank x N B bit for bit.
 * Costs a short last round of tiles per GEMM.  Default 0.  Applies to every session of the model. */
int vc_model_set_batch_invariant(vc_model* m, int on);
/* Round 6: the prefill's QKV GEMM applies RoPE, splits the heads and writes Q / the KV-cache rows / the V^T scratch in its epilogue
 * (replaces [HF] llama/modeling_llama.py:254-262 "q/k/v_proj -> view -> apply_rotary_pos_emb -> cache update" as ONE launch); default 1.
 * 0 = the separate GEMM + split launches of rounds 1-5 (regression / A-B); 1 = fused from 1024 token rows (where the 256 x 256 GEMM
 * kernel runs anyway); 2 = fused for every problem size.  Needs head_dim 128 and hidden_size % 256 == 0, else the separate launches. */
int vc_model_set_qkv_fused(vc_model* m, int on);
/* weight format 2 ("fp8"): KV cache of the decode steps in e4m3 (default 1) or bf16 (0).  Before vc_model_finalize. */
int vc_model_set_fp8_kv(vc_model* m, int on);
/* vc_model_synth_tensor with the value classes of the reference's checkpoints: rounding 0 bf16, 1 fp16-valued, 2 unrounded fp32
 * (vcoder_amd/synth.py synth_tensor(rounding=...)); 1 and 2 take the fp32 load path and keep weight lo planes */
int vc_model_synth_tensor_rounded(vc_model* m, const char* hf_key, const int64_t* shape, int ndim, uint32_t tensor_seed, float offset,
                                  float halfwidth, int rounding);
int vc_model_finalize(vc_model* m);
/* Number of loaded tensors whose fp32 source held values bf16 cannot represent — the reference's own checkpoints: an fp16 LLM
 * (model/builder.py:25-40, torch_dtype=float16) and an fp32 CLIP hub checkpoint (multimodal_encoder/clip_encoder.py:22-27).  Every
 * such matrix keeps a second bf16 plane lo = bf16(w - bf16(w)); precision modes "strict" and "split" contract against hi + lo (the
 * checkpoint's values to ~16 mantissa bits; exact for fp16 values), the bf16 fast path uses the bf16-rounded weights alone.
 * 0 for a bf16 checkpoint.  >= 0, or a negative vc_status. */
int vc_model_inexact_tensors(vc_model* m);
/* Round 6: which 16-bit format the MFMA operands / stored activations of THIS library have: 0 = bfloat16 (libvcoder_hip.so), 1 = IEEE
 * fp16 (libvcoder_hip_f16.so — same sources, same C ABI, built with -DVC_F16: fp16 operands on v_mfma_f32_16x16x32_f16 at the bf16
 * rate, conversions saturating at 65504; the operand precision of the reference's own GPU path, vcoder_llava/model/builder.py:39
 * torch_dtype=float16 and :142).  "bf16" in the names and comments of this header then reads "the library's operand format": an
 * fp16-valued checkpoint is held exactly (vc_model_inexact_tensors counts what fp16 cannot hold). */
int vc_operand_format(void);

/* arithmetic mode: 0 = bf16 MFMA operands, fp32 accumulate/residual/softmax (default; what bench.py measures);
 * 1 = strict: fp32 activations end to end on fp32 MFMA (slow) — within ~1e-5 of the reference's fp32 CPU path;
 * 2 = split: fp32 activations in HBM, every MFMA operand carried as two bf16 values (x = hi + lo, ~16 mantissa bits) on the
 *     FAST kernels — GEMMs contract the [hi | lo] rows against the (exactly bf16) weight twice, attention uses three MFMAs per
 *     product, fp24 KV cache (the top 24 bits of the fp32 values; VC_SPLIT_KV=32: fp32), one weight pass per 32-row pooled
 *     decode step; runs in sessions, the hipGraph loop and the decode pool at 0.60 of the bf16 path's rate.
 * 1 and 2 meet the "logits within 1e-3, greedy ids bit-exact" bar of BASELINE.json (2: 1.9e-4 at full 7b depth over 128
 * tokens, every greedy id equal to the fp32 reference's; the same at 13b, B = 16).  Takes effect at the next prefill. */
int vc_model_set_precision(vc_model* m, int mode);

/* parity diagnostic: prefills evaluate only the first n decoder layers (0 = all), final norm + lm_head applied to that
 * hidden state — error growth with depth is measured on one loaded model (tests/test_gpu_fulldepth.py) */
int vc_model_set_layer_limit(vc_model* m, int n_layers);

/* parity diagnostic (per-layer teacher forcing): decoder layers [l0, l1) of a prefill applied to a caller-supplied residual
 * stream x_in [B, S, hidden] (host fp32, positions 0..S-1) in the model's weight format / precision mode; x_out [B, S, hidden]
 * = the residual stream behind layer l1 - 1.  Every layer can be fed the oracle's own input, so that rounding / quantisation
 * noise of the layers in front of it does not compound (tests/test_gpu_e2e.py: fp8 formats per layer). */
int vc_debug_prefill_layers(vc_model* m, int l0, int l1, const float* x_in, int B, int S, float* x_out);

/* decoder weight storage: 0 = bf16 (default); 1 = W8A16 — the seven linears of every decoder layer are quantised at
 * vc_model_finalize to OCP fp8 e4m3 with one power-of-two scale per output row and streamed as bytes by the decode
 * GEMV (half the HBM traffic of the decode step); the prefill GEMMs read the same dequantised values in bf16, so both
 * phases compute with one set of effective weights.  2 = fp8, BASELINE.json configs[4] ("fp8 weights, CDNA4 fp8 MFMA"):
 * the weights and decode steps of 1, and the prefill's decoder linears quantise their activation rows to e4m3 (one
 * power-of-two scale per token row) and run e4m3 x e4m3 on v_mfma_scale_f32_16x16x128_f8f6f4 (W8A8, twice the bf16
 * MFMA rate), and the KV cache holds e4m3 rows (no scale, saturating; vc_model_set_fp8_kv(m, 0) before vc_model_finalize: bf16 rows) — half the bytes of the stream
 * that dominates the pooled 13b decode step.  The reference's counterpart is `load_8bit` (builder.py:31-33, bitsandbytes LLM.int8 — also 8-bit
 * weights x 8-bit activations).  Call before vc_model_finalize. */
int vc_model_set_weight_format(vc_model* m, int fmt);

/* ---- hot path ------------------------------------------------------------------------------ */
/* encode_images / encode_seg_images / encode_depth_images (vcoder_ds_llava_arch.py:106-119):
 * pixels fp32 [B,3,S,S] (host, or device when pixels_on_device) -> projected features fp32 [B,P,hidden] on host. */
int vc_encode(vc_model* m, int modality, const float* pixels, int pixels_on_device, int B, float* out_feats_host);

/* CLIPVisionTower.forward + feature_select (vcoder_llava/model/multimodal_encoder/clip_encoder.py:29-51): the un-projected
 * tower output the reference's encode_* hand to the adapters — hidden_states[mm_vision_select_layer] of N images with the
 * CLS row dropped for 'patch'.  pixels fp32 [N,3,S,S] -> out fp32 [N, R, vit_hidden] on the host (R = patches, +1 for
 * 'cls_patch'). */
int vc_vision_tower_forward(vc_model* m, const float* pixels, int pixels_on_device, int N, float* out_host);

/* images per sample for the NEXT vc_prefill* / vc_generate* call (one-shot): the reference's list / 5-D image form
 * (vcoder_ds_llava_arch.py:135-169) — sample b owns counts[b] images of a modality, whose feature rows are spliced as ONE
 * block at its placeholder; the pixel pointer of that modality then holds sum(counts) images.  NULL = one per sample. */
int vc_set_image_counts(vc_model* m, const int32_t* img_counts, const int32_t* seg_counts, const int32_t* depth_counts, int B);

/* output_hidden_states for the NEXT vc_prefill (one-shot): out (host, cap_floats floats) receives [(layers + 1), B, S, hidden]
 * fp32 — inputs_embeds, the residual stream behind every decoder layer, the last entry after the final RMSNorm: the tuple
 * [HF] LlamaModel.forward returns as hidden_states (vcoder_ds_llava_llama.py:81-90,117).  out must stay valid until that
 * vc_prefill returns; NULL cancels.  Requested before a vc_decode_step instead, it is that cached step's tuple:
 * [(layers + 1), B, 1, hidden] (the step then runs eagerly, outside its hipGraph). */
int vc_request_hidden_states(vc_model* m, float* out, size_t cap_floats);

/* output_attentions for the NEXT vc_prefill (one-shot): out (host, cap_floats floats) receives [layers, B, heads, S, S] fp32 — the
 * attention probabilities HF's eager attention returns as `attentions` (vcoder_ds_llava_llama.py:81-90,118; [HF]
 * llama/modeling_llama.py eager_attention_forward): softmax(q k^T / sqrt(hd) + causal mask + padding mask), recomputed per layer
 * from that layer's q / k by a diagnostic kernel (the flash kernels never materialise them).  S <= 4096.  Requested before a
 * vc_decode_step: [layers, B, heads, 1, pos + 1] — the new token's query over every cached key and itself. */
int vc_request_attentions(vc_model* m, float* out, size_t cap_floats);

/* Padded batches: the caller's 2-D attention_mask [B, T] (bytes, 0 = hidden) for the NEXT vc_prefill* / vc_generate* call
 * (one-shot).  As in the reference, it is LEFT-extended with "visible" over the S - T rows the splice adds — by position
 * (vcoder_ds_llava_arch.py:305-311) — and a hidden position is hidden as a KEY from every query of its sequence during the
 * prefill ([HF] LlamaModel: causal mask + padding mask); position ids stay arange(S).  Spliced lengths must be equal (quirk 6)
 * and position 0 visible.  Cached steps: vc_generate* runs them under an all-ones mask, like the reference's multimodal decode
 * path (vcoder_ds_llava_arch.py:130-133 replaces the mask by ones: the padded positions' keys become visible); a vc_prefill +
 * vc_decode_step loop keeps the prefill's hidden keys hidden (a caller carrying its mask through the steps) until
 * vc_clear_attention_mask(). */
int vc_set_attention_mask(vc_model* m, const uint8_t* mask, int B, int T);
int vc_clear_attention_mask(vc_model* m);

/* KV-cache slots the next vc_prefill keeps free behind the prompt for vc_decode_step loops (default 64, clamped to
 * max_position_embeddings).  A loop that outruns the reserve still works: the cache grows (one copy of the live prefix). */
int vc_model_reserve_decode(vc_model* m, int max_new_tokens);

/* prepare_inputs_labels_for_multimodal + LlamaModel + lm_head (vcoder_ds_llava_llama.py:57-118), prefill.
 * ids [B,T] int64 host with -200/-300/-400 placeholders; seg/depth may be NULL.  has_attention_mask only
 * selects the reference's behaviour for unequal spliced lengths (error vs zero right-padding).
 * logits_last [B,V] and/or logits_all [B,S,V] (host fp32) may be NULL.  Leaves the KV cache at length S. */
int vc_prefill(vc_model* m, const int64_t* ids, int B, int T, const float* img, const float* seg, const float* depth,
               int pixels_on_device, int has_attention_mask, float* logits_last, float* logits_all, int* S_out);
/* encode + splice only: inputs_embeds [B,S,hidden] fp32 to host (what prepare_inputs_labels_for_multimodal returns) */
int vc_prefill_embeds_only(vc_model* m, const int64_t* ids, int B, int T, const float* img, const float* seg,
                           const float* depth, int pixels_on_device, int has_attention_mask, float* out_host,
                           int* S_out);
/* the spliced length S the same arguments would give a vc_prefill / vc_generate call — the splice plan of
 * prepare_inputs_labels_for_multimodal (vcoder_ds_llava_arch.py:175-276) alone: no tower pass, no KV / mask state touched;
 * img / seg only say whether the modality is present, the depth pixels are read for is_depth_zero (:161).  Same plan errors
 * (VC_ERR_INDEX, VC_ERR_UNEQUAL) as the real call. */
int vc_plan_spliced_len(vc_model* m, const int64_t* ids, int B, int T, const float* img, const float* seg, const float* depth,
                        int pixels_on_device, int has_attention_mask, int* S_out);

/* one cached decode step (input_ids.shape[1]==1 fast path, vcoder_ds_llava_arch.py:130-133).
 * tok [B] host (NULL: use the token the previous step selected on device); logits [B,V] host or NULL;
 * next_tok [B] host (greedy argmax, lowest index on ties) or NULL. */
int vc_decode_step(vc_model* m, const int32_t* tok, float* logits, int32_t* next_tok);

/* beam search support: the KV rows of the current vc_prefill / vc_decode_step loop are permuted, row r <- old row src_rows[r] —
 * `past_key_values` reordered by beam_idx after a beam step ([HF] generation/utils.py beam_search; the reference's eval loaders
 * forward num_beams: eval/model_seg_loader.py:129-139) */
int vc_reorder_cache(vc_model* m, const int32_t* src_rows, int B);

/* greedy generate(): encode + splice + prefill + (max_new-1) hipGraph-replayed decode steps, HF semantics
 * (SURVEY.md Appendix C): eos_id < 0 disables EOS; finished rows emit pad_id; stops when all rows finished.
 * generate() always carries an attention_mask, so unequal spliced lengths fail with VC_ERR_UNEQUAL (quirk 6).
 * out_ids [B,max_new] int32 host (row-major, unused tail = pad_id); n_generated = columns produced. */
int vc_generate_greedy(vc_model* m, const int64_t* ids, int B, int T, const float* img, const float* seg,
                       const float* depth, int pixels_on_device, int max_new, int eos_id, int pad_id,
                       int32_t* out_ids, int* n_generated);
/* the same with a device-side keyword stop: n_stop token sequences (stop_ids flattened,
 * stop_lens[i] ids each; at most 8 sequences of at most 8 ids).  A row is finished — later tokens are pad_id — once
 * its ids (prompt tail + generated) end with one of them; generation ends when every row is finished by EOS or a stop.
 * Batched, hipGraph-friendly form of KeywordsStoppingCriteria's id match (vcoder_llava/mm_utils.py:128-151: batch size
 * 1, evaluated on the host after every token). */
int vc_generate_greedy_stop(vc_model* m, const int64_t* ids, int B, int T, const float* img, const float* seg,
                            const float* depth, int pixels_on_device, int max_new, int eos_id, int pad_id,
                            const int32_t* stop_ids, const int32_t* stop_lens, int n_stop, int32_t* out_ids,
                            int* n_generated);

/* generate() in full — what vcoder_llava/serve/cli.py:122-132 and serve/chat.py:141-151 call: greedy, or sampling on the
 * device when samp->do_sample (HF order: temperature, top-k, top-p, multinomial).  The draw is a
 * counter-based generator keyed by (seed, row of the batch, step, vocabulary index): the same seed gives the same tokens.
 * cb (may be NULL): streamer hook, called on the calling thread with the ids of steps [first_step, first_step + n_steps)
 * of every row (ids [B][n_steps] row-major) every cb_every steps — the decode loop stays hipGraph-replayed in between. */
typedef struct vc_sampling {
    int32_t do_sample;   /* 0: greedy */
    float temperature;   /* > 0 */
    int32_t top_k;       /* <= 0: off (HF's GenerationConfig default is 50) */
    float top_p;         /* (0, 1]; 1: off */
    uint64_t seed;
} vc_sampling;
typedef void (*vc_token_cb)(void* user, int first_step, int n_steps, int B, const int32_t* ids);
int vc_generate(vc_model* m, const int64_t* ids, int B, int T, const float* img, const float* seg, const float* depth,
                int pixels_on_device, int max_new, int eos_id, int pad_id, const int32_t* stop_ids, const int32_t* stop_lens,
                int n_stop, const vc_sampling* samp, vc_token_cb cb, void* cb_user, int cb_every, int32_t* out_ids,
                int* n_generated);

/* spliced sequence length (text rows + feature rows) of the last vc_generate_greedy* call: lets a caller that splits a large
 * batch into replica-sized pieces reproduce the reference's whole-batch behaviour for unequal lengths (quirk 6) */
int vc_last_spliced_len(vc_model* m);

/* ---- next row §8(f)2: image preprocessing on the device ------------------------------------------------------------
 * process_images() of vcoder_llava/mm_utils.py:28-40 for ONE image: expand2square(mean colour) when pad_to_square,
 * PIL-exact bicubic resize (shortest edge -> S) + center crop, rescale 1/255, (x-mean)/std, HWC->CHW.
 * rgb: uint8 [h,w,3] on the host; out: fp32 [3,S,S] (device pointer when out_on_device, else host). */
int vc_preprocess_image(vc_model* m, const uint8_t* rgb, int h, int w, int pad_to_square, const float* mean,
                        const float* stdv, float* out, int out_on_device);

/* ---- multi-GPU: the one exchange of the data-parallel path ------------------------------------------
 * Image batches shard over the GPUs of a node with no collective on the data path; the only exchange is an all-gather of
 * the generated token ids (the reference: one answers file per GPU process + `cat`, scripts/v1_5/eval/cost_depth.sh:10-34).
 * RCCL (ncclAllGather over xGMI) is called directly on the context's stream; librccl is bound with dlopen at first use.
 * Rank 0 obtains the 128-byte RCCL unique id and distributes it through any side channel; world 1 needs no id and no RCCL. */
typedef struct vc_comm vc_comm;
int vc_comm_unique_id(vc_ctx* ctx, void* out128);
int vc_comm_create(vc_ctx* ctx, int rank, int world, const void* unique_id128, vc_comm** out);
/* 1 when the communicator's gathers run through RCCL: world > 1, or a SINGLE rank created with VC_COMM_FORCE_RCCL=1 in the
 * environment (a one-rank ncclCommInitRank + ncclAllGather on the stream: the multi-GPU code path on a one-GPU box) */
int vc_comm_uses_rccl(vc_comm* comm);
/* global[r * n + i] = rank r's local[i]; host buffers (n int32 per rank) */
int vc_allgather_tokens(vc_comm* comm, const int32_t* local, int n, int32_t* global);
void vc_comm_destroy(vc_comm* comm);

/* ---- measurement hooks (bench.py) ---------------------------------------------------------------- */
/* times `reps` sweeps of every decode GEMV launch of one step (4 per layer + lm_head) over B <= 32 rows with HIP events on
 * the model's stream; returns launches per sweep, average microseconds per launch, algorithmic weight bytes per launch */
int vc_profile_decode_gemv(vc_model* m, int B, int reps, int* launches, double* avg_us, double* avg_bytes);
/* cumulative number of pooled decode steps launched over 8 / 16 / 24 / 32 rows since the pool of m's root model was created
 * (zeros without a pool): weights the per-row-count kernel timings by what a timed run actually executed */
int vc_pool_step_counts(vc_model* m, unsigned long long* counts4);
/* In-situ timing of the pool's decode-step kernels (bench.py `roofline`: the dominant kernel AS IT RAN in the timed region, beside
 * whatever the other sessions had on the GPU — not a replay).  on != 0: the pool's step graphs are (re)captured with one timing slot
 * per launch; the first thread of every workgroup writes its own {start, end} on the device's constant-rate wall clock (plain
 * stores, no atomics), and one small launch per step folds the slots into per-(span, kind) sums of (latest end - earliest start).  Takes effect when the pool is next (re)built, i.e. while
 * no generate() is in flight.  bf16 path only.  No reference counterpart (the reference has no timing code). */
int vc_pool_profile(vc_model* m, int on);
/* Pool scheduling: on = 1 (default) the pool does not step while a generate() call that already holds rows is still prefilling —
 * it waits for that request to join instead of running a step beside the prefill's GEMMs (2-7x a step's time) that the joiner
 * would need again anyway; on = 0 steps whatever rows are active (lower inter-token latency for the calls in flight, lower
 * throughput).  No reference counterpart (the reference runs one generate() at a time). */
int vc_pool_set_hold(vc_model* m, int on);
/* Rows of the shared decode pool: 32 (default) or 64 (round-6 measurement: two 32-row weight passes per step).  Takes effect when the
 * pool is next built (idle).  64 rows: bf16 step only, no in-situ timing. */
int vc_pool_set_rows(vc_model* m, int rows);
/* sums since the last reset, each [4 spans: 8 / 16 / 24 / 32 rows][6 kinds: qkv, decode attention, o_proj, gate/up, down, lm_head]:
 * exec_us = earliest workgroup start -> latest workgroup end of the launches; period_us = latest end of the previous launch of the
 * step -> latest end of this one (dispatch, drain and inter-kernel gap included: what the step's dependency chain pays per launch;
 * rocprofv3's per-kernel duration lies between the two); launches.  reset != 0 zeroes them.  The pool must be idle. */
int vc_pool_profile_read(vc_model* m, double* exec_us, double* period_us, unsigned long long* launches, int reset);
/* the same for the decode attention launches of one step (one per layer) over B rows at context ~ctx: launches per sweep,
 * average microseconds per launch, algorithmic KV bytes per launch (K and V rows of every key, bf16) */
int vc_profile_decode_attention(vc_model* m, int B, int ctx, int reps, int* launches, double* avg_us, double* avg_bytes);
/* wall-clock split of the last vc_generate_greedy in ms (HIP events): encode, prefill, decode */
int vc_last_timings(vc_model* m, float* encode_ms, float* prefill_ms, float* decode_ms);

#ifdef __cplusplus
}
#endif
#endif /* VCODER_HIP_H */
