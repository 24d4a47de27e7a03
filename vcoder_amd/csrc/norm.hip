// norm.hip — wavefront-reduction normalisation kernels + the ViT front end.
//
//   LayerNorm  ([HF] clip/modeling_clip.py:370,379,642; nn.LayerNorm, biased variance, eps 1e-5)       K2/K3
//   RMSNorm    ([HF] llama/modeling_llama.py:53-70: fp32 variance, x*rsqrt(var+eps), then *weight)       K11
//   im2col     (Conv2d(3->Dv, k=s=14, no bias) of [HF] clip :149-155,209 restated as a GEMM operand)      K1
//   embed+preLN([HF] clip :212-217 CLS concat + position add, :642 pre_layrnorm)                         K2
//   feature_select (multimodal_encoder/clip_encoder.py:29-37: hidden_states[-2], drop CLS)               K8
//
// One 64-lane wave owns one row; the row lives in registers (float4 per lane per 256 columns), statistics
// are two-pass in fp32 and reduced with 6 cross-lane xor steps — no LDS, no barriers.  HBM-bound.
#include "vc_device.h"
#include "kernels.h"

namespace vc {

// MAXV float4 per lane -> rows up to MAXV*256 columns
// OUT: 0 = bf16, 1 = fp32, 2 = split: bf16 hi = bf16(o) at y and lo = bf16(o - hi) at y + lo_off (precision mode "split":
// the K-concatenated [hi | lo] operand of a kwrap GEMM, or the stacked hi / lo rows of a split decode GEMV)
template <int MAXV, bool RMS, int OUT>
VC_DEV void norm_row(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ b,
                     void* __restrict__ y, int D, float eps, const float* __restrict__ add0,
                     const float* __restrict__ add1, size_t lo_off = 0) {
    const int lane = lane_id();
    const int nch = D >> 2;
    // Every load is UNCONDITIONAL at a clamped chunk (masked afterwards) and the loops that load only load (round 6, ISA): a load under
    // `if (c < nch)` — true for every lane at D = 4096, which the compiler cannot know — is waited for with vmcnt(0) where its guard ends,
    // so a wave had ONE 1-KiB load in flight at a time and the kernel's rate was occupancy x 1 KiB per HBM latency (5.3 TB/s at 64 VGPRs,
    // 2.5 TB/s for the 128-VGPR instances).  Now a wave requests its whole row, then the weights in batches of WB.
    f32x4 v[MAXV];
#pragma unroll
    for (int i = 0; i < MAXV; ++i) v[i] = ld16f(x + min(lane + i * 64, nch - 1) * 4);
    if (add0) {
        f32x4 a[MAXV];
#pragma unroll
        for (int i = 0; i < MAXV; ++i) a[i] = ld16f(add0 + min(lane + i * 64, nch - 1) * 4);
#pragma unroll
        for (int i = 0; i < MAXV; ++i) v[i] = v[i] + a[i];
    }
    if (add1) {
        f32x4 a[MAXV];
#pragma unroll
        for (int i = 0; i < MAXV; ++i) a[i] = ld16f(add1 + min(lane + i * 64, nch - 1) * 4);
#pragma unroll
        for (int i = 0; i < MAXV; ++i) v[i] = v[i] + a[i];
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        if (lane + i * 64 >= nch) v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
    }
    float mean = 0.f;
    if (!RMS) mean = wave_sum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = lane + i * 64;
        if (c < nch) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float d = v[i][e] - mean;
                q += d * d;
            }
        }
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)D + eps);
    constexpr int WB = MAXV < 8 ? MAXV : (MAXV % 8 == 0 ? 8 : 4);   // chunks per weight batch (MAXV in 4, 16, 20, 32)
    static_assert(MAXV % WB == 0, "weight batches");
#pragma unroll
    for (int i0 = 0; i0 < MAXV; i0 += WB) {
        f32x4 wv[WB], bv[WB];
#pragma unroll
        for (int k = 0; k < WB; ++k) {
            const int cc = min(lane + (i0 + k) * 64, nch - 1);
            wv[k] = ld16f(w + cc * 4);
            if (!RMS) bv[k] = ld16f(b + cc * 4);
        }
#pragma unroll
        for (int k = 0; k < WB; ++k) {   // (an empty asm that takes the registers: the batch is waited for HERE, not inside the guards below)
            pin_vgprs(wv[k]);
            if (!RMS) pin_vgprs(bv[k]);
        }
#pragma unroll
        for (int k = 0; k < WB; ++k) {
            const int i = i0 + k, c = lane + i * 64;
            if (c < nch) {
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = (v[i][e] - mean) * rstd * wv[k][e];
                if (!RMS) o = o + bv[k];
                if constexpr (OUT == 1) {
                    st16f(reinterpret_cast<float*>(y) + c * 4, o);
                } else {
                    u32x2 pk = {pack_bf2(o[0], o[1]), pack_bf2(o[2], o[3])};
                    st8(reinterpret_cast<bf16_t*>(y) + c * 4, pk);
                    if constexpr (OUT == 2) {
                        const u32x2 lo = {pack_bf2(o[0] - bf2f_lo(pk[0]), o[1] - bf2f_hi(pk[0])),
                                          pack_bf2(o[2] - bf2f_lo(pk[1]), o[3] - bf2f_hi(pk[1]))};
                        st8(reinterpret_cast<bf16_t*>(y) + lo_off + c * 4, lo);
                    }
                }
            }
        }
    }
}

template <int MAXV, bool RMS>
__global__ __launch_bounds__(256) void norm_kernel(const float* x, const int* row_idx, const float* w, const float* b,
                                                   bf16_t* y, int rows, int D, float eps, int ldy) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;  // wave-uniform
    const int src = row_idx ? row_idx[row] : row;
    norm_row<MAXV, RMS, 0>(x + (size_t)src * D, w, b, y + (size_t)row * ldy, D, eps, nullptr, nullptr);
}

// fp32-out form for the strict path
template <int MAXV, bool RMS>
__global__ __launch_bounds__(256) void norm_f32_kernel(const float* x, const int* row_idx, const float* w, const float* b,
                                                       float* y, int rows, int D, float eps) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int src = row_idx ? row_idx[row] : row;
    norm_row<MAXV, RMS, 1>(x + (size_t)src * D, w, b, y + (size_t)row * D, D, eps, nullptr, nullptr);
}
template <bool RMS>
static void launch_norm_f32(const float* x, const int* idx, const float* w, const float* b, float* y, int rows, int D,
                            float eps, hipStream_t s) {
    const dim3 grid((rows + 3) / 4), block(256);
    if (D <= 1024) VC_LAUNCH((norm_f32_kernel<4, RMS>), grid, block, 0, s, x, idx, w, b, y, rows, D, eps);
    else if (D <= 4096) VC_LAUNCH((norm_f32_kernel<16, RMS>), grid, block, 0, s, x, idx, w, b, y, rows, D, eps);
    else if (D <= 5120) VC_LAUNCH((norm_f32_kernel<20, RMS>), grid, block, 0, s, x, idx, w, b, y, rows, D, eps);
    else VC_LAUNCH((norm_f32_kernel<32, RMS>), grid, block, 0, s, x, idx, w, b, y, rows, D, eps);
}
void launch_layernorm_f32(const float* x, const float* w, const float* b, float* y, int rows, int D, float eps,
                          hipStream_t s) {
    launch_norm_f32<false>(x, nullptr, w, b, y, rows, D, eps, s);
}
void launch_rmsnorm_f32(const float* x, const int* row_idx, const float* w, float* y, int rows, int D, float eps,
                        hipStream_t s) {
    launch_norm_f32<true>(x, row_idx, w, nullptr, y, rows, D, eps, s);
}

// split-out form: row r -> hi at y + r * ldy, lo at y + r * ldy + lo_off
template <int MAXV, bool RMS>
__global__ __launch_bounds__(256) void norm_split_kernel(const float* x, const int* row_idx, const float* w, const float* b,
                                                         bf16_t* y, int rows, int D, float eps, int ldy, size_t lo_off) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;  // wave-uniform
    const int src = row_idx ? row_idx[row] : row;
    norm_row<MAXV, RMS, 2>(x + (size_t)src * D, w, b, y + (size_t)row * ldy, D, eps, nullptr, nullptr, lo_off);
}
template <bool RMS>
static void launch_norm_split(const float* x, const int* idx, const float* w, const float* b, bf16_t* y, int rows, int D,
                              float eps, int ldy, size_t lo_off, hipStream_t s) {
    const dim3 grid((rows + 3) / 4), block(256);
    if (D <= 1024) VC_LAUNCH((norm_split_kernel<4, RMS>), grid, block, 0, s, x, idx, w, b, y, rows, D, eps, ldy, lo_off);
    else if (D <= 4096) VC_LAUNCH((norm_split_kernel<16, RMS>), grid, block, 0, s, x, idx, w, b, y, rows, D, eps, ldy, lo_off);
    else if (D <= 5120) VC_LAUNCH((norm_split_kernel<20, RMS>), grid, block, 0, s, x, idx, w, b, y, rows, D, eps, ldy, lo_off);
    else VC_LAUNCH((norm_split_kernel<32, RMS>), grid, block, 0, s, x, idx, w, b, y, rows, D, eps, ldy, lo_off);
}
void launch_layernorm_split(const float* x, const float* w, const float* b, bf16_t* y, int rows, int D, float eps, int ldy,
                            size_t lo_off, hipStream_t s) {
    launch_norm_split<false>(x, nullptr, w, b, y, rows, D, eps, ldy, lo_off, s);
}
void launch_rmsnorm_split(const float* x, const int* row_idx, const float* w, bf16_t* y, int rows, int D, float eps, int ldy,
                          size_t lo_off, hipStream_t s) {
    launch_norm_split<true>(x, row_idx, w, nullptr, y, rows, D, eps, ldy, lo_off, s);
}

template <bool RMS>
static void launch_norm(const float* x, const int* idx, const float* w, const float* b, bf16_t* y, int rows, int D,
                        float eps, hipStream_t s, int ldy = 0) {
    const dim3 grid((rows + 3) / 4), block(256);
    if (ldy <= 0) ldy = D;
    if (D <= 1024) VC_LAUNCH((norm_kernel<4, RMS>), grid, block, 0, s, x, idx, w, b, y, rows, D, eps, ldy);
    else if (D <= 4096) VC_LAUNCH((norm_kernel<16, RMS>), grid, block, 0, s, x, idx, w, b, y, rows, D, eps, ldy);
    else if (D <= 5120) VC_LAUNCH((norm_kernel<20, RMS>), grid, block, 0, s, x, idx, w, b, y, rows, D, eps, ldy);
    else VC_LAUNCH((norm_kernel<32, RMS>), grid, block, 0, s, x, idx, w, b, y, rows, D, eps, ldy);
}

void launch_layernorm(const float* x, const float* w, const float* b, bf16_t* y, int rows, int D, float eps,
                      hipStream_t s) {
    launch_norm<false>(x, nullptr, w, b, y, rows, D, eps, s);
}
void launch_rmsnorm(const float* x, const float* w, bf16_t* y, int rows, int D, float eps, hipStream_t s, int ldy) {
    launch_norm<true>(x, nullptr, w, nullptr, y, rows, D, eps, s, ldy);
}
void launch_rmsnorm_rows(const float* x, const int* row_idx, const float* w, bf16_t* y, int rows, int D, float eps,
                         hipStream_t s) {
    launch_norm<true>(x, row_idx, w, nullptr, y, rows, D, eps, s);
}

// ---- RMSNorm straight into the e4m3 activation operand of the fp8 prefill GEMM (weight format 2) -------------------------
// One wave per row: y = bf16(x * rstd * w) exactly as norm_row computes it, then the row's power-of-two scale and the e4m3
// bytes by the rule of quant_act_rows_kernel (decode.hip) — the same bytes that kernel would produce from the bf16 row,
// without writing and re-reading it.
template <int MAXV>
__global__ __launch_bounds__(256) void rmsnorm_q8_kernel(const float* x, const float* w, uint8_t* q, float* scale, int rows,
                                                         int D, float eps) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;  // wave-uniform
    const int lane = lane_id();
    const int nch = D >> 2;
    const float* xr = x + (size_t)row * D;
    // unconditional loads at clamped chunks, masked afterwards; the row first, then the weights (norm_row says why)
    f32x4 v[MAXV];
#pragma unroll
    for (int i = 0; i < MAXV; ++i) v[i] = ld16f(xr + min(lane + i * 64, nch - 1) * 4);
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        if (lane + i * 64 >= nch) v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int e = 0; e < 4; ++e) ss += v[i][e] * v[i][e];
    }
    const float rstd = rsqrtf(wave_sum(ss) / (float)D + eps);
    float amax = 0.f;
    constexpr int WB = MAXV < 8 ? MAXV : (MAXV % 8 == 0 ? 8 : 4);
    static_assert(MAXV % WB == 0, "weight batches");
#pragma unroll
    for (int i0 = 0; i0 < MAXV; i0 += WB) {
        f32x4 wv[WB];
#pragma unroll
        for (int k = 0; k < WB; ++k) wv[k] = ld16f(w + min(lane + (i0 + k) * 64, nch - 1) * 4);
#pragma unroll
        for (int k = 0; k < WB; ++k) {
            const int i = i0 + k;
            if (lane + i * 64 < nch) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[i][e] = bf2f(f2bf(v[i][e] * rstd * wv[k][e]));
                    amax = fmaxf(amax, fabsf(v[i][e]));
                }
            }
        }
    }
    amax = wave_max(amax);
    int ex = 0;
    if (amax > 0.f) {
        const uint32_t u = __builtin_bit_cast(uint32_t, amax);
        ex = (int)(u >> 23) - 127 - ((u & 0x007FFFFFu) <= 0x00600000u ? 8 : 7);  // 448 = 1.75 * 2^8
    }
    const float inv = __builtin_bit_cast(float, (uint32_t)(127 - ex) << 23);
    if (lane == 0) scale[row] = __builtin_bit_cast(float, (uint32_t)(ex + 127) << 23);
    uint8_t* qr = q + (size_t)row * D;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = lane + i * 64;
        if (c < nch) {
            const uint32_t pk = f32x4_to_fp8x4_inrange(v[i][0] * inv, v[i][1] * inv, v[i][2] * inv, v[i][3] * inv);
            *reinterpret_cast<uint32_t*>(qr + c * 4) = pk;
        }
    }
}
void launch_rmsnorm_q8(const float* x, const float* w, uint8_t* q, float* scale, int rows, int D, float eps, hipStream_t s) {
    const dim3 grid((rows + 3) / 4), block(256);
    if (D <= 1024) VC_LAUNCH((rmsnorm_q8_kernel<4>), grid, block, 0, s, x, w, q, scale, rows, D, eps);
    else if (D <= 4096) VC_LAUNCH((rmsnorm_q8_kernel<16>), grid, block, 0, s, x, w, q, scale, rows, D, eps);
    else if (D <= 5120) VC_LAUNCH((rmsnorm_q8_kernel<20>), grid, block, 0, s, x, w, q, scale, rows, D, eps);
    else VC_LAUNCH((rmsnorm_q8_kernel<32>), grid, block, 0, s, x, w, q, scale, rows, D, eps);
}

// ---- K2: CLS concat + position embedding + pre-LayerNorm -> fp32 residual stream ----------------------
template <int MAXV>
__global__ __launch_bounds__(256) void vit_embed_ln_kernel(const float* patches, const float* cls, const float* pos,
                                                           const float* w, const float* b, float* x, int n_img, int T,
                                                           int D, float eps) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n_img * T) return;
    const int n = row / T, t = row % T;
    const float* src = t == 0 ? cls : patches + ((size_t)n * (T - 1) + (t - 1)) * D;
    norm_row<MAXV, false, 1>(src, w, b, x + (size_t)row * D, D, eps, pos + (size_t)t * D, nullptr);
}
void launch_vit_embed_ln(const float* patches, const float* cls, const float* pos, const float* w, const float* b,
                         float* x, int n_img, int T, int D, float eps, hipStream_t s) {
    const dim3 grid((n_img * T + 3) / 4), block(256);
    if (D <= 1024) VC_LAUNCH((vit_embed_ln_kernel<4>), grid, block, 0, s, patches, cls, pos, w, b, x, n_img, T, D, eps);
    else VC_LAUNCH((vit_embed_ln_kernel<16>), grid, block, 0, s, patches, cls, pos, w, b, x, n_img, T, D, eps);
}

// ---- K1: im2col (pixels fp32 NCHW -> bf16 [N*g*g, Kpad]); column = c*P*P + py*P + px ---------------------
// ldc = row stride of cols (Kpad, or 2 * Kpad with lo_off = Kpad for the split [hi | lo] form)
__global__ __launch_bounds__(256) void im2col_kernel(const float* pixels, bf16_t* cols, int n_img, int S, int P,
                                                     int Kpad, int ldc, int lo_off) {
    const int g = S / P, chunks = Kpad >> 3;
    const size_t total = (size_t)n_img * g * g * chunks;
    const size_t id = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (id >= total) return;
    const int ch = (int)(id % chunks);
    const size_t row = id / chunks;
    const int n = (int)(row / (g * g)), gy = (int)((row / g) % g), gx = (int)(row % g);
    const int Kreal = 3 * P * P;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int col = ch * 8 + e;
        if (col < Kreal) {
            const int c = col / (P * P), py = (col / P) % P, px = col % P;
            v[e] = pixels[(((size_t)n * 3 + c) * S + (gy * P + py)) * S + gx * P + px];
        } else {
            v[e] = 0.f;
        }
    }
    u32x4 o = {pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]), pack_bf2(v[4], v[5]), pack_bf2(v[6], v[7])};
    st16(cols + row * ldc + ch * 8, o);
    if (lo_off) {
        u32x4 l;
#pragma unroll
        for (int e = 0; e < 4; ++e) l[e] = pack_bf2(v[2 * e] - bf2f_lo(o[e]), v[2 * e + 1] - bf2f_hi(o[e]));
        st16(cols + row * ldc + lo_off + ch * 8, l);
    }
}
void launch_im2col(const float* pixels, bf16_t* cols, int n_img, int image, int patch, int Kpad, hipStream_t s, bool split) {
    const int g = image / patch;
    const size_t total = (size_t)n_img * g * g * (Kpad / 8);
    VC_LAUNCH(im2col_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, pixels, cols, n_img, image, patch,
              Kpad, split ? 2 * Kpad : Kpad, split ? Kpad : 0);
}

// ---- K8: feature_select — drop the first `skip` rows of every image, fp32 -> bf16 ----------------------
__global__ __launch_bounds__(256) void select_rows_kernel(const float* x, bf16_t* y, int n_img, int T, int skip,
                                                          int D, int ldy, int lo_off) {
    const int chunks = D >> 3;
    const size_t total = (size_t)n_img * (T - skip) * chunks;
    const size_t id = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (id >= total) return;
    const int ch = (int)(id % chunks);
    const size_t orow = id / chunks;
    const size_t n = orow / (T - skip), t = orow % (T - skip) + skip;
    const float* src = x + (n * T + t) * D + ch * 8;
    const f32x4 a = ld16f(src), b = ld16f(src + 4);
    u32x4 o = {pack_bf2(a[0], a[1]), pack_bf2(a[2], a[3]), pack_bf2(b[0], b[1]), pack_bf2(b[2], b[3])};
    st16(y + orow * ldy + ch * 8, o);
    if (lo_off) {
        const u32x4 l = {pack_bf2(a[0] - bf2f_lo(o[0]), a[1] - bf2f_hi(o[0])), pack_bf2(a[2] - bf2f_lo(o[1]), a[3] - bf2f_hi(o[1])),
                         pack_bf2(b[0] - bf2f_lo(o[2]), b[1] - bf2f_hi(o[2])), pack_bf2(b[2] - bf2f_lo(o[3]), b[3] - bf2f_hi(o[3]))};
        st16(y + orow * ldy + lo_off + ch * 8, l);
    }
}
void launch_select_rows_bf16(const float* x, bf16_t* y, int n_img, int T, int skip, int D, hipStream_t s, bool split) {
    const size_t total = (size_t)n_img * (T - skip) * (D / 8);
    VC_LAUNCH(select_rows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, x, y, n_img, T, skip, D,
              split ? 2 * D : D, split ? D : 0);
}

}  // namespace vc
