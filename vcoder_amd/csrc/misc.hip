// misc.hip — splice/gather, greedy select, step bookkeeping, synthetic-weight generator, dtype converts.
//
//   splice        : embedding gather + <image>/<seg>/<depth> feature splice into inputs_embeds
//                   (vcoder_ds_llava_arch.py:173-276,305; vcoder_llava_arch.py:180-287)                   K10
//   embed_tokens  : decode-step embedding lookup ([HF] llama/modeling_llama.py:377)                        K10
//   greedy        : fp32 argmax, lowest index on ties, EOS -> pad bookkeeping
//                   ([HF] generation/utils.py:2894,2925-2929)                        K19
#include "vc_device.h"
#include "kernels.h"

namespace vc {

// one wave per destination row: bf16 source row -> fp32 residual-stream row
__global__ __launch_bounds__(256) void splice_kernel(const int* row_src, int nrows, const bf16_t* embed,
                                                     const bf16_t* feats, float* x, int D) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= nrows) return;
    const int lane = threadIdx.x & 63;
    const int kind = row_src[2 * row], src = row_src[2 * row + 1];
    const bf16_t* sp = kind == 0 ? embed + (size_t)src * D : feats + (size_t)src * D;
    float* dp = x + (size_t)row * D;
    for (int c = lane; c < D / 8; c += 64) {
        u32x4 v = {0u, 0u, 0u, 0u};
        if (kind != 2) v = ld16(sp + c * 8);
        st16f(dp + c * 8, f32x4{bf2f_lo(v[0]), bf2f_hi(v[0]), bf2f_lo(v[1]), bf2f_hi(v[1])});
        st16f(dp + c * 8 + 4, f32x4{bf2f_lo(v[2]), bf2f_hi(v[2]), bf2f_lo(v[3]), bf2f_hi(v[3])});
    }
}
void launch_splice(const int* row_src, int nrows, const bf16_t* embed, const bf16_t* feats, float* x, int D,
                   hipStream_t s) {
    VC_LAUNCH(splice_kernel, dim3((nrows + 3) / 4), dim3(256), 0, s, row_src, nrows, embed, feats, x, D);
}

__global__ __launch_bounds__(256) void embed_tokens_kernel(const int* tok, const bf16_t* embed, float* x, int B, int D,
                                                           const bf16_t* embed_lo) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= B) return;
    const int lane = threadIdx.x & 63;
    const bf16_t* sp = embed + (size_t)tok[row] * D;
    float* dp = x + (size_t)row * D;
    for (int c = lane; c < D / 8; c += 64) {
        const u32x4 v = ld16(sp + c * 8);
        f32x4 a = {bf2f_lo(v[0]), bf2f_hi(v[0]), bf2f_lo(v[1]), bf2f_hi(v[1])};
        f32x4 b = {bf2f_lo(v[2]), bf2f_hi(v[2]), bf2f_lo(v[3]), bf2f_hi(v[3])};
        if (embed_lo != nullptr) {   // the lo plane of an inexact checkpoint (strict / split modes)
            const u32x4 w = ld16(embed_lo + (size_t)tok[row] * D + c * 8);
            a = a + f32x4{bf2f_lo(w[0]), bf2f_hi(w[0]), bf2f_lo(w[1]), bf2f_hi(w[1])};
            b = b + f32x4{bf2f_lo(w[2]), bf2f_hi(w[2]), bf2f_lo(w[3]), bf2f_hi(w[3])};
        }
        st16f(dp + c * 8, a);
        st16f(dp + c * 8 + 4, b);
    }
}
void launch_embed_tokens(const int* tok, const bf16_t* embed, float* x, int B, int D, hipStream_t s, const bf16_t* embed_lo) {
    VC_LAUNCH(embed_tokens_kernel, dim3((B + 3) / 4), dim3(256), 0, s, tok, embed, x, B, D, embed_lo);
}

// ---- greedy select: one workgroup per batch row ------------------------------------------------------
VC_DEV void argmax_combine(float& v, int& i, float ov, int oi) {
    if (ov > v || (ov == v && oi < i)) { v = ov; i = oi; }
}
__global__ __launch_bounds__(256) void greedy_kernel(GreedyArgs p) {
    __shared__ float sv[4];
    __shared__ int si[4];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* lg = p.logits + (size_t)b * p.V;
    float best = -INFINITY;
    int bi = 0x7FFFFFFF;
    for (int i = tid; i < p.V; i += 256) argmax_combine(best, bi, lg[i], i);
#pragma unroll
    for (int mk = 32; mk >= 1; mk >>= 1) {
        const float ov = shfl_xor(best, mk);
        const int oi = shfl_xor(bi, mk);
        argmax_combine(best, bi, ov, oi);
    }
    if (lane == 0) { sv[wave] = best; si[wave] = bi; }
    __syncthreads();
    if (tid != 0) return;
    for (int w = 1; w < 4; ++w) argmax_combine(best, bi, sv[w], si[w]);
    int tok = bi;
    if (p.eos_id >= 0) {
        if (p.finished[b]) tok = p.pad_id;
        if (tok == p.eos_id) p.finished[b] = 1;
    }
    p.next_tok[b] = tok;
    const int step = *p.step_dev;
    if (step < p.max_new) p.out_ids[(size_t)b * p.max_new + step] = tok;
}
void launch_greedy(const GreedyArgs& a, hipStream_t s) {
    VC_LAUNCH(greedy_kernel, dim3(a.B), dim3(256), 0, s, a);
}

__global__ __launch_bounds__(64) void advance_kernel(int* step_dev, int* pos_dev, int* ctx_dev) {
    if (threadIdx.x != 0) return;
    if (step_dev) *step_dev += 1;
    if (pos_dev) *pos_dev += 1;
    if (ctx_dev) *ctx_dev += 1;
}
void launch_advance(int* step_dev, int* pos_dev, int* ctx_dev, hipStream_t s) {
    VC_LAUNCH(advance_kernel, dim3(1), dim3(64), 0, s, step_dev, pos_dev, ctx_dev);
}

// ---- synthetic checkpoint generator: bit-identical to vcoder_amd/synth.py:synth_tensor ----------------
// (`__fmul_rn` / `__fadd_rn` are plain `*` / `+` in this ROCm's headers, and HIP compiles with -ffp-contract=fast: without the
// pragma the multiply-add below is ONE fma — one rounding where vcoder_amd/synth.py (numpy) has two: the fp32 values with an offset
// differed by an ulp in 5 % of the elements (found by the fp16 / fp32 value classes of round 5; the bf16 class rounds it away))
VC_DEV float synth_value_f32(uint32_t idx, uint32_t tseed, float offset, float scale) {
#pragma clang fp contract(off)
    uint32_t x = idx * 0x9E3779B1u + tseed;
    x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16;
    const float u = (float)(x >> 8);
    float v = (u - 8388607.5f) * scale;
    if (offset != 0.f) v = v + offset;
    return v;
}
VC_DEV float synth_value(uint32_t idx, uint32_t tseed, float offset, float scale) {   // on the bfloat16 grid in EVERY build (synth.py)
    return true_bf2f(true_f2bf(synth_value_f32(idx, tseed, offset, scale)));
}
__global__ __launch_bounds__(256) void synth_bf16_kernel(bf16_t* out, size_t n, uint32_t tseed, float offset,
                                                         float scale) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        out[i] = f2bf(synth_value((uint32_t)i, tseed, offset, scale));
}
__global__ __launch_bounds__(256) void synth_f32_kernel(float* out, size_t n, uint32_t tseed, float offset, float scale) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        out[i] = synth_value((uint32_t)i, tseed, offset, scale);
}
// the same generator with the value classes of the reference's own checkpoints (vcoder_amd/synth.py synth_tensor(rounding=...)):
// rounding 1 = the value an fp16 checkpoint holds (RNE to 11 significant bits, fp16 range / subnormals), 2 = unrounded fp32
VC_DEV float round_to_fp16(float v) {
    // RNE to fp16 and back, in integer arithmetic (identical on the device and the emulator): normal range 2^-14 .. 65504,
    // subnormals in steps of 2^-24
    const float a = fabsf(v);
    if (a >= 65520.f) return v < 0 ? -INFINITY : INFINITY;
    float q;
    if (a < 6.103515625e-05f) {                                  // subnormal: multiples of 2^-24
        q = rintf(a * 16777216.f) * 5.9604644775390625e-08f;
    } else {
        uint32_t u = __builtin_bit_cast(uint32_t, a);
        u = (u + 0x00000FFFu + ((u >> 13) & 1u)) & 0xFFFFE000u;   // 23 -> 10 mantissa bits, ties to even
        q = __builtin_bit_cast(float, u);
    }
    return v < 0 ? -q : q;
}
__global__ __launch_bounds__(256) void synth_f32_rounded_kernel(float* out, size_t n, uint32_t tseed, float offset, float scale, int rounding) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float v = synth_value_f32((uint32_t)i, tseed, offset, scale);
        out[i] = rounding == 1 ? round_to_fp16(v) : v;
    }
}
static unsigned grid_for(size_t n) { return (unsigned)min((size_t)8192, (n + 255) / 256); }
void launch_synth_bf16(bf16_t* out, size_t n, uint32_t tseed, float offset, float halfwidth, hipStream_t s) {
    VC_LAUNCH(synth_bf16_kernel, dim3(grid_for(n)), dim3(256), 0, s, out, n, tseed, offset,
              (float)(halfwidth / 8388608.0));
}
void launch_synth_f32_rounded(float* out, size_t n, uint32_t tseed, float offset, float halfwidth, int rounding, hipStream_t s) {
    VC_LAUNCH(synth_f32_rounded_kernel, dim3(grid_for(n)), dim3(256), 0, s, out, n, tseed, offset, (float)(halfwidth / 8388608.0), rounding);
}
void launch_synth_f32(float* out, size_t n, uint32_t tseed, float offset, float halfwidth, hipStream_t s) {
    VC_LAUNCH(synth_f32_kernel, dim3(grid_for(n)), dim3(256), 0, s, out, n, tseed, offset,
              (float)(halfwidth / 8388608.0));
}

// per-row fp32 partial sums (is_depth_zero = mean(depth)==0, vcoder_ds_llava_arch.py:161): ROW_SUM_PARTS workgroups per
// row, each a contiguous slice; the host adds the partials (only "== 0" matters, and zeros sum to zero in any order)
__global__ __launch_bounds__(256) void row_sum_kernel(const float* x, size_t n_per_row, float* out) {
    __shared__ float red[4];
    const int part = blockIdx.y;
    const size_t per = (n_per_row + ROW_SUM_PARTS - 1) / ROW_SUM_PARTS;
    const size_t i0 = (size_t)part * per, i1 = min(n_per_row, i0 + per);
    const float* r = x + (size_t)blockIdx.x * n_per_row;
    float s = 0.f;
    for (size_t i = i0 + threadIdx.x; i < i1; i += 256) s += r[i];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) out[(size_t)blockIdx.x * ROW_SUM_PARTS + part] = (red[0] + red[1]) + (red[2] + red[3]);
}
void launch_row_sum(const float* x, size_t n_per_row, int rows, float* out, hipStream_t s) {
    VC_LAUNCH(row_sum_kernel, dim3(rows, ROW_SUM_PARTS), dim3(256), 0, s, x, n_per_row, out);
}

// ---- KV-cache row permutation (beam search: `past_key_values` reordered by beam_idx, [HF] generation/utils.py _reorder_cache)
// cache [rows][H][capS][hd] (elements of es bytes); phase 0: tmp[r][h][s] = cache[perm[r]][h][s] for s < live; phase 1:
// cache[r][h][s] = tmp[r][h][s].  One thread = 16 bytes.
__global__ __launch_bounds__(256) void kv_permute_kernel(char* cache, char* tmp, const int* perm, int rows, int H, size_t cap_row_bytes,
                                                         size_t live_row_bytes, int phase) {
    const size_t chunks = live_row_bytes / 16;                 // per (row, head)
    const size_t total = (size_t)rows * H * chunks;
    for (size_t id = (size_t)blockIdx.x * 256 + threadIdx.x; id < total; id += (size_t)gridDim.x * 256) {
        const size_t c = id % chunks, rh = id / chunks;
        const int h = (int)(rh % H), r = (int)(rh / H);
        char* t = tmp + ((size_t)r * H + h) * live_row_bytes + c * 16;
        if (phase == 0) st16(t, ld16(cache + ((size_t)perm[r] * H + h) * cap_row_bytes + c * 16));
        else st16(cache + ((size_t)r * H + h) * cap_row_bytes + c * 16, ld16(t));
    }
}
void launch_kv_permute(void* cache, void* tmp, const int* perm, int rows, int H, size_t cap_row_bytes, size_t live_row_bytes,
                       hipStream_t s) {
    const size_t total = (size_t)rows * H * (live_row_bytes / 16);
    if (total == 0) return;
    const unsigned grid = (unsigned)min((size_t)4096, (total + 255) / 256);
    VC_LAUNCH(kv_permute_kernel, dim3(grid), dim3(256), 0, s, reinterpret_cast<char*>(cache), reinterpret_cast<char*>(tmp), perm, rows, H,
              cap_row_bytes, live_row_bytes, 0);
    VC_LAUNCH(kv_permute_kernel, dim3(grid), dim3(256), 0, s, reinterpret_cast<char*>(cache), reinterpret_cast<char*>(tmp), perm, rows, H,
              cap_row_bytes, live_row_bytes, 1);
}

__global__ __launch_bounds__(256) void f32_to_bf16_kernel(const float* in, bf16_t* out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) out[i] = f2bf(in[i]);
}
__global__ __launch_bounds__(256) void bf16_to_f32_kernel(const bf16_t* in, float* out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) out[i] = bf2f(in[i]);
}
// checkpoint values that bf16 cannot hold (an fp16 / fp32 checkpoint): hi = bf16(x), lo = bf16(x - hi) (x = hi + lo to ~16 mantissa
// bits; exact for fp16 values), *inexact |= any x != hi.  lo == nullptr: hi and the flag only.
__global__ __launch_bounds__(256) void f32_to_bf16_planes_kernel(const float* in, bf16_t* hi, bf16_t* lo, size_t n, unsigned* inexact) {
    bool any = false;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float x = in[i];
        const bf16_t h = f2bf(x);
        const float r = x - bf2f(h);
        hi[i] = h;
        if (lo != nullptr) lo[i] = f2bf(r);
        any = any || r != 0.f;
    }
    if (any && inexact != nullptr) *inexact = 1u;   // (every writer stores the same value)
}
// checkpoint data that arrives as bfloat16 bits -> fp32 (whatever this build's 16-bit operand format is)
__global__ __launch_bounds__(256) void truebf16_to_f32_kernel(const uint16_t* in, float* out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) out[i] = true_bf2f(in[i]);
}
void launch_truebf16_to_f32(const uint16_t* in, float* out, size_t n, hipStream_t s) {
    VC_LAUNCH(truebf16_to_f32_kernel, dim3(grid_for(n)), dim3(256), 0, s, in, out, n);
}
void launch_f32_to_bf16_planes(const float* in, bf16_t* hi, bf16_t* lo, size_t n, unsigned* inexact, hipStream_t s) {
    VC_LAUNCH(f32_to_bf16_planes_kernel, dim3(grid_for(n)), dim3(256), 0, s, in, hi, lo, n, inexact);
}
void launch_f32_to_bf16(const float* in, bf16_t* out, size_t n, hipStream_t s) {
    VC_LAUNCH(f32_to_bf16_kernel, dim3(grid_for(n)), dim3(256), 0, s, in, out, n);
}
void launch_bf16_to_f32(const bf16_t* in, float* out, size_t n, hipStream_t s) {
    VC_LAUNCH(bf16_to_f32_kernel, dim3(grid_for(n)), dim3(256), 0, s, in, out, n);
}

}  // namespace vc
