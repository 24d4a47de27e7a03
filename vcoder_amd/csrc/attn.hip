// attn.hip — attention for the VCoder hot path on gfx950.
//
//   qkv_split   : fused-QKV GEMM output -> Q [B,H,Tq,hd], K cache [B,H,S,hd], V^T cache [B,H,hd,S] (+RoPE)
//                 RoPE = rotate_half form of [HF] llama/modeling_llama.py:113-160 (cos/sin fp32 table)      K13/K14
//   attention   : flash-style softmax(Q K^T * scale [+causal]) V, fp32 online softmax, MFMA 16x16x32 bf16
//                 ViT: [HF] clip/modeling_clip.py:259-277,320-330 (non-causal, hd 64, T 577)              K5
//                 LLM prefill: [HF] llama eager_attention_forward :191-214 (causal, hd 128)               K15
//   attention_decode : q_len = 1 over the KV cache (HBM-streaming, VALU dot products)                    K15
//
// Layout choices are ours (the reference has none): K is key-major so K tiles are MFMA A-operands for
// S^T = K Q^T straight from LDS; V is stored TRANSPOSED (d-major) so V^T tiles are A-operands for
// O^T = V^T P^T; with S^T in the MFMA C layout each lane already owns the P values of ONE query, so
// row max / row sum need only 2 cross-lane steps and P feeds the second MFMA from registers (the
// contraction order over keys is permuted identically on both operands).
#include <stdlib.h>

#include "vc_device.h"
#include "kernels.h"

namespace vc {

// =============================================================================================
// qkv split (+RoPE), prefill form: one workgroup = (64-token tile, head, batch)
// =============================================================================================
template <int HD>
__global__ __launch_bounds__(256) void qkv_split_kernel(QkvSplitArgs p) {
    const float* __restrict__ rope_cos = p.rope_cos;
    const float* __restrict__ rope_sin = p.rope_sin;
    __shared__ __attribute__((aligned(16))) bf16_t vt_tile[64][HD + 8];  // +8 bf16 pad: transposed reads spread banks
    const int t0 = blockIdx.x * 64, h = blockIdx.y, b = blockIdx.z;
    const int tid = threadIdx.x;
    const int D = p.H * HD;
    constexpr int CPT = HD / 16;  // rope work items per token: chunk c pairs d0=8c with d0+HD/2
    const bool rope = rope_cos != nullptr;
    for (int w = tid; w < 64 * CPT; w += 256) {
        const int tl = w / CPT, c = w % CPT, t = t0 + tl;
        if (t >= p.T) continue;
        const bf16_t* row = p.qkv + ((size_t)b * p.T + t) * (3 * D) + h * HD;
#pragma unroll
        for (int which = 0; which < 2; ++which) {  // 0 = q, 1 = k
            const bf16_t* src = row + which * D;
            const u32x4 lo = ld16(src + c * 8), hi = ld16(src + HD / 2 + c * 8);
            u32x4 olo = lo, ohi = hi;
            if (rope) {
                const float* cs = rope_cos + (size_t)t * (HD / 2) + c * 8;
                const float* sn = rope_sin + (size_t)t * (HD / 2) + c * 8;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float x0 = bf2f_lo(lo[e]), x1 = bf2f_hi(lo[e]), y0 = bf2f_lo(hi[e]), y1 = bf2f_hi(hi[e]);
                    const float c0 = cs[2 * e], c1 = cs[2 * e + 1], s0 = sn[2 * e], s1 = sn[2 * e + 1];
                    // out[d] = x*cos - y*sin ; out[d+hd/2] = y*cos + x*sin
                    olo[e] = pack_bf2(x0 * c0 - y0 * s0, x1 * c1 - y1 * s1);
                    ohi[e] = pack_bf2(y0 * c0 + x0 * s0, y1 * c1 + x1 * s1);
                }
            }
            bf16_t* dst = which == 0 ? p.q + (((size_t)b * p.H + h) * p.q_stride + t) * HD
                                     : p.k + (((size_t)b * p.H + h) * p.kv_stride + t) * HD;
            st16(dst + c * 8, olo);
            st16(dst + HD / 2 + c * 8, ohi);
        }
    }
    // V: stage [token][d] tile, then write V^T rows 16 bytes (8 tokens) at a time
    for (int w = tid; w < 64 * (HD / 8); w += 256) {
        const int tl = w / (HD / 8), c = w % (HD / 8), t = t0 + tl;
        u32x4 v = {0u, 0u, 0u, 0u};
        if (t < p.T) v = ld16(p.qkv + ((size_t)b * p.T + t) * (3 * D) + 2 * D + h * HD + c * 8);
        st16(&vt_tile[tl][c * 8], v);
    }
    __syncthreads();
    for (int w = tid; w < HD * 8; w += 256) {
        const int d = w >> 3, tc = w & 7;
        if (t0 + tc * 8 >= p.T) continue;
        uint32_t o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e)
            o[e] = (uint32_t)vt_tile[tc * 8 + 2 * e][d] | ((uint32_t)vt_tile[tc * 8 + 2 * e + 1][d] << 16);
        st16(p.vt + (((size_t)b * p.H + h) * HD + d) * p.kv_stride + t0 + tc * 8, u32x4{o[0], o[1], o[2], o[3]});
    }
}

// decode form (T == 1): position read from a device scalar; one wave per (b,h)
template <int HD>
__global__ __launch_bounds__(64) void qkv_append_kernel(QkvSplitArgs p) {
    const float* __restrict__ rope_cos = p.rope_cos;
    const float* __restrict__ rope_sin = p.rope_sin;
    const int h = blockIdx.x, b = blockIdx.y, d = threadIdx.x;
    if (d >= HD / 2) return;
    const int pos = *p.pos0_dev;
    const int D = p.H * HD;
    const bf16_t* row = p.qkv + (size_t)b * (3 * D) + h * HD;
    const float c = rope_cos[(size_t)pos * (HD / 2) + d], s = rope_sin[(size_t)pos * (HD / 2) + d];
    const float q0 = bf2f(row[d]), q1 = bf2f(row[d + HD / 2]);
    const float k0 = bf2f(row[D + d]), k1 = bf2f(row[D + d + HD / 2]);
    bf16_t* qo = p.q + ((size_t)b * p.H + h) * HD;
    qo[d] = f2bf(q0 * c - q1 * s);
    qo[d + HD / 2] = f2bf(q1 * c + q0 * s);
    bf16_t* ko = p.k + (((size_t)b * p.H + h) * p.kv_stride + pos) * HD;
    ko[d] = f2bf(k0 * c - k1 * s);
    ko[d + HD / 2] = f2bf(k1 * c + k0 * s);
    bf16_t* vo = p.vt + ((size_t)b * p.H + h) * HD * (size_t)p.kv_stride + pos;
    vo[(size_t)d * p.kv_stride] = row[2 * D + d];
    vo[(size_t)(d + HD / 2) * p.kv_stride] = row[2 * D + d + HD / 2];
}

void launch_qkv_split(const QkvSplitArgs& a, hipStream_t s) {
    if (a.pos0_dev != nullptr) {  // decode append
        const dim3 grid(a.H, a.B), block(64);
        if (a.hd == 128) VC_LAUNCH((qkv_append_kernel<128>), grid, block, 0, s, a);
        else VC_LAUNCH((qkv_append_kernel<64>), grid, block, 0, s, a);
        return;
    }
    const dim3 grid((a.T + 63) / 64, a.H, a.B), block(256);
    if (a.hd == 128) VC_LAUNCH((qkv_split_kernel<128>), grid, block, 0, s, a);
    else VC_LAUNCH((qkv_split_kernel<64>), grid, block, 0, s, a);
}

// =============================================================================================
// flash attention: workgroup = 128 queries of one (b,h) as WAVES waves x QS 16-query sub-tiles; KV tiles of 64 keys
// =============================================================================================
template <int HD> VC_DEV int swz_k(int row, int chunk) {  // K tile [64][HD] bf16, 16-B chunks
    if constexpr (HD == 128) return row * 256 + ((chunk ^ (row & 15)) << 4);
    else return row * 128 + ((chunk ^ (row & 7)) << 4);
}
VC_DEV int swz_v(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }  // V^T tile [HD][64]

template <int HD, bool CAUSAL, int WAVES, int QS>
__global__ __launch_bounds__(WAVES * 64) void attention_kernel(AttnArgs p) {
    constexpr int NTH = WAVES * 64;       // threads per workgroup
    constexpr int QB = WAVES * QS * 16;   // queries per workgroup
    constexpr int KS = HD / 32;        // k-steps of the QK^T contraction
    constexpr int DT = HD / 16;        // 16-wide d tiles of the output
    constexpr int KCH = HD / 8;        // 16-B chunks per K row
    __shared__ __attribute__((aligned(16))) char k_lds[64 * HD * 2];
    __shared__ __attribute__((aligned(16))) char v_lds[HD * 128];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, j = lane & 15;
    const int q0 = blockIdx.x * QB, h = blockIdx.y, b = blockIdx.z;
    const size_t bh = (size_t)b * p.H + h;
    const bf16_t* qbase = p.q + bh * p.q_stride * HD;
    const bf16_t* kbase = p.k + bh * p.kv_stride * HD;
    const bf16_t* vbase = p.vt + bh * HD * (size_t)p.kv_stride;

    // Q fragments (MFMA B operand): lane holds Q[query j][d = ks*32 + g*8 .. +8]
    u32x4 qf[QS][KS];
#pragma unroll
    for (int qs = 0; qs < QS; ++qs) {
        const int qrow = min(q0 + wave * (QS * 16) + qs * 16 + j, p.T - 1);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) qf[qs][ks] = ld16(qbase + (size_t)qrow * HD + ks * 32 + g * 8);
    }
    f32x4 o[QS][DT];
#pragma unroll
    for (int qs = 0; qs < QS; ++qs)
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) o[qs][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m_run[QS], l_run[QS];
#pragma unroll
    for (int qs = 0; qs < QS; ++qs) { m_run[qs] = -INFINITY; l_run[qs] = 0.f; }

    const int kv_end = CAUSAL ? min(p.T, q0 + QB) : p.T;
    const int nkt = (kv_end + 63) / 64;
    constexpr int KLD = (64 * KCH) / NTH, VLD = (HD * 8) / NTH;  // staged 16-B chunks per thread
    u32x4 rk[KLD], rv[VLD];
    auto load_tile = [&](int kt) {
#pragma unroll
        for (int i = 0; i < KLD; ++i) {
            const int c = tid + i * NTH, row = c / KCH, ch = c % KCH;
            rk[i] = ld16(kbase + (size_t)(kt * 64 + row) * HD + ch * 8);
        }
#pragma unroll
        for (int i = 0; i < VLD; ++i) {
            const int c = tid + i * NTH, row = c >> 3, ch = c & 7;
            rv[i] = ld16(vbase + (size_t)row * p.kv_stride + kt * 64 + ch * 8);
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int i = 0; i < KLD; ++i) {
            const int c = tid + i * NTH;
            st16(k_lds + swz_k<HD>(c / KCH, c % KCH), rk[i]);
        }
#pragma unroll
        for (int i = 0; i < VLD; ++i) {
            const int c = tid + i * NTH;
            st16(v_lds + swz_v(c >> 3, c & 7), rv[i]);
        }
    };

    load_tile(0);
    for (int kt = 0; kt < nkt; ++kt) {
        store_tile();
        __syncthreads();
        if (kt + 1 < nkt) load_tile(kt + 1);
        const int k0 = kt * 64;
        // ---- S^T = K Q^T
        f32x4 sacc[QS][4];
#pragma unroll
        for (int qs = 0; qs < QS; ++qs)
#pragma unroll
            for (int sub = 0; sub < 4; ++sub) sacc[qs][sub] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int sub = 0; sub < 4; ++sub)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const u32x4 kf = ld16(k_lds + swz_k<HD>(sub * 16 + j, ks * 4 + g));
#pragma unroll
                for (int qs = 0; qs < QS; ++qs) sacc[qs][sub] = mfma16(kf, qf[qs][ks], sacc[qs][sub]);
            }
        // ---- online softmax (lane owns query j of each q-subtile; keys spread over regs and the 4 lane groups)
        u32x4 pb[QS][2];
#pragma unroll
        for (int qs = 0; qs < QS; ++qs) {
            const int query = q0 + wave * (QS * 16) + qs * 16 + j;
            float mx = -INFINITY;
#pragma unroll
            for (int sub = 0; sub < 4; ++sub)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int key = k0 + sub * 16 + g * 4 + r;
                    float v = sacc[qs][sub][r] * p.scale;
                    if (key >= p.T || (CAUSAL && key > query)) v = -INFINITY;
                    sacc[qs][sub][r] = v;
                    mx = fmaxf(mx, v);
                }
            mx = fmaxf(mx, shfl_xor(mx, 16));
            mx = fmaxf(mx, shfl_xor(mx, 32));
            const float m_new = fmaxf(m_run[qs], mx);
            const float alpha = __expf(m_run[qs] - m_new);  // first tile: exp(-inf) = 0
            m_run[qs] = m_new;
            float rs = 0.f;
#pragma unroll
            for (int sub = 0; sub < 4; ++sub)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float e = __expf(sacc[qs][sub][r] - m_new);
                    sacc[qs][sub][r] = e;
                    rs += e;
                }
            l_run[qs] = l_run[qs] * alpha + rs;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) o[qs][dt] = o[qs][dt] * alpha;
#pragma unroll
            for (int kh = 0; kh < 2; ++kh)
                pb[qs][kh] = u32x4{pack_bf2(sacc[qs][2 * kh][0], sacc[qs][2 * kh][1]),
                                   pack_bf2(sacc[qs][2 * kh][2], sacc[qs][2 * kh][3]),
                                   pack_bf2(sacc[qs][2 * kh + 1][0], sacc[qs][2 * kh + 1][1]),
                                   pack_bf2(sacc[qs][2 * kh + 1][2], sacc[qs][2 * kh + 1][3])};
        }
        // ---- O^T += V^T P^T   (contraction slot (g,e) <-> key kh*32 + (e>>2)*16 + 4g + (e&3) on both operands)
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int kh = 0; kh < 2; ++kh) {
                const int row = dt * 16 + j;
                const int c0 = kh * 4 + (g >> 1), within = (g & 1) * 8;
                const u32x2 lo = ld8(v_lds + swz_v(row, c0) + within);
                const u32x2 hi = ld8(v_lds + swz_v(row, c0 + 2) + within);
                const u32x4 vf = {lo[0], lo[1], hi[0], hi[1]};
#pragma unroll
                for (int qs = 0; qs < QS; ++qs) o[qs][dt] = mfma16(vf, pb[qs][kh], o[qs][dt]);
            }
        __syncthreads();
    }
    // ---- normalise and store: lane holds out[query j][d = dt*16 + g*4 .. +4]
#pragma unroll
    for (int qs = 0; qs < QS; ++qs) {
        float l = l_run[qs];
        l += shfl_xor(l, 16);
        l += shfl_xor(l, 32);
        const float inv = 1.0f / l;
        const int query = q0 + wave * (QS * 16) + qs * 16 + j;
        if (query < p.T) {
            bf16_t* dst = p.out + ((size_t)b * p.T + query) * ((size_t)p.H * HD) + h * HD + g * 4;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                const f32x4 v = o[qs][dt] * inv;
                st8(dst + dt * 16, u32x2{pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3])});
            }
        }
    }
}

template <int HD, int WAVES, int QS>
static void launch_attention_v(const AttnArgs& a, hipStream_t s) {
    const int QB = WAVES * QS * 16;
    const dim3 grid((a.T + QB - 1) / QB, a.H, a.B), block(WAVES * 64);
    if (a.causal) VC_LAUNCH((attention_kernel<HD, true, WAVES, QS>), grid, block, 0, s, a);
    else VC_LAUNCH((attention_kernel<HD, false, WAVES, QS>), grid, block, 0, s, a);
}

void launch_attention(const AttnArgs& a, hipStream_t s) {
    // 8 waves x 16 queries keeps the hd-128 kernel at 128 VGPRs (the 4x32 form needs ~250 -> 1 wave/SIMD); measured on
    // MI355X: prefill (hd 128, T 1216, causal) 288 us vs 378 us; ViT (hd 64, T 577) 97 us vs 117 us.
    static const int variant = getenv("VC_ATTN_VARIANT") ? atoi(getenv("VC_ATTN_VARIANT")) : 0;
    if (a.hd == 128) {
        if (variant == 1) launch_attention_v<128, 4, 2>(a, s);
        else launch_attention_v<128, 8, 1>(a, s);
    } else {
        if (variant == 1) launch_attention_v<64, 4, 2>(a, s);
        else launch_attention_v<64, 8, 1>(a, s);
    }
}

// =============================================================================================
// decode attention: one 512-thread workgroup per (b,h); K rows / V^T rows streamed once from HBM
// =============================================================================================
constexpr int DEC_MAX_CTX = 4096;

template <int HD>
__global__ __launch_bounds__(512) void attention_decode_kernel(AttnDecodeArgs p) {
    constexpr int LPK = HD / 8;           // lanes per key (each lane owns 8 dims)
    constexpr int KPW = 64 / LPK;         // keys per wave-instruction
    __shared__ __attribute__((aligned(16))) float sc[DEC_MAX_CTX];
    __shared__ float red[8];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = blockIdx.x, b = blockIdx.y;
    const size_t bh = (size_t)b * p.H + h;
    const int ctx = *p.ctx_len_dev;
    const int ctx64 = (ctx + 63) & ~63;
    const bf16_t* kbase = p.k + bh * p.kv_stride * HD;
    const bf16_t* vbase = p.vt + bh * HD * (size_t)p.kv_stride;

    // ---- phase 1: scores
    float qv[8];
    {
        const u32x4 q = ld16(p.q + bh * HD + (lane % LPK) * 8);
#pragma unroll
        for (int e = 0; e < 4; ++e) { qv[2 * e] = bf2f_lo(q[e]); qv[2 * e + 1] = bf2f_hi(q[e]); }
    }
    for (int kb = wave * KPW; kb < ctx64; kb += 8 * KPW) {
        const int key = kb + lane / LPK;
        const u32x4 kv = ld16(kbase + (size_t)min(key, ctx - 1) * HD + (lane % LPK) * 8);
        float s = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) s += qv[2 * e] * bf2f_lo(kv[e]) + qv[2 * e + 1] * bf2f_hi(kv[e]);
#pragma unroll
        for (int mk = 1; mk < LPK; mk <<= 1) s += shfl_xor(s, mk);
        if ((lane % LPK) == 0) sc[key] = key < ctx ? s * p.scale : -INFINITY;
    }
    __syncthreads();
    // ---- phase 2: softmax over sc[0..ctx64)
    float mx = -INFINITY;
    for (int i = tid; i < ctx64; i += 512) mx = fmaxf(mx, sc[i]);
    mx = wave_max(mx);
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    mx = red[0];
#pragma unroll
    for (int w = 1; w < 8; ++w) mx = fmaxf(mx, red[w]);
    __syncthreads();
    float sum = 0.f;
    for (int i = tid; i < ctx64; i += 512) {
        const float e = __expf(sc[i] - mx);  // masked keys: exp(-inf) = 0
        sc[i] = e;
        sum += e;
    }
    sum = wave_sum(sum);
    if (lane == 0) red[wave] = sum;
    __syncthreads();
    sum = 0.f;
#pragma unroll
    for (int w = 0; w < 8; ++w) sum += red[w];
    const float inv = 1.0f / sum;
    // ---- phase 3: out[d] = sum_key p[key] V^T[d][key]; 8 lanes per d row, 64 keys per instruction
    constexpr int DG = HD / 64;  // d-row groups (of 8 rows) per wave: 8 waves x 8 rows = 64 rows per pass
    const int dr = lane >> 3, kc = lane & 7;
    float acc[DG];
#pragma unroll
    for (int i = 0; i < DG; ++i) acc[i] = 0.f;
    for (int kb = 0; kb < ctx64; kb += 64) {
        const f32x4 p0 = ld16f(&sc[kb + kc * 8]), p1 = ld16f(&sc[kb + kc * 8 + 4]);
#pragma unroll
        for (int i = 0; i < DG; ++i) {
            const int d = (i * 8 + wave) * 8 + dr;
            const u32x4 v = ld16(vbase + (size_t)d * p.kv_stride + kb + kc * 8);
            acc[i] += p0[0] * bf2f_lo(v[0]) + p0[1] * bf2f_hi(v[0]) + p0[2] * bf2f_lo(v[1]) + p0[3] * bf2f_hi(v[1]) +
                      p1[0] * bf2f_lo(v[2]) + p1[1] * bf2f_hi(v[2]) + p1[2] * bf2f_lo(v[3]) + p1[3] * bf2f_hi(v[3]);
        }
    }
#pragma unroll
    for (int i = 0; i < DG; ++i) {
        float a = acc[i];
        a += shfl_xor(a, 1);
        a += shfl_xor(a, 2);
        a += shfl_xor(a, 4);
        if (kc == 0) p.out[(size_t)b * p.H * HD + h * HD + (i * 8 + wave) * 8 + dr] = f2bf(a * inv);
    }
}

void launch_attention_decode(const AttnDecodeArgs& a, hipStream_t s) {
    const dim3 grid(a.H, a.B), block(512);
    if (a.hd == 128) VC_LAUNCH((attention_decode_kernel<128>), grid, block, 0, s, a);
    else VC_LAUNCH((attention_decode_kernel<64>), grid, block, 0, s, a);
}

// =============================================================================================
// fused decode attention: RoPE + KV append + softmax(q K^T) V for the one new token of each (b,h).
// One 512-thread workgroup per (b,h).  K rows and V^T rows are streamed once from HBM with many independent
// 16-byte loads in flight per wave (4 keys x 4 per wave in the score pass; 16 d-rows per wave in the PV pass).
// =============================================================================================
// HALF: the V^T stream of a 64-key block is taken in two halves of HD/2 rows with one half always in flight while the other
// accumulates — half the registers of the two-full-sets form (212 -> <= 128 VGPRs at HD = 128), so TWO workgroups are
// resident per CU and one's RoPE/append, softmax and reduction phases (HBM idle for that workgroup) hide behind the other's
// streaming.  Same loads, same accumulation order per output element.
template <int HD, bool HALF>
__global__ __launch_bounds__(512, HALF ? 4 : 2) void attention_decode_fused_kernel(AttnDecodeFusedArgs p) {
    constexpr int LPK = HD / 8;           // lanes per key
    constexpr int KPW = 64 / LPK;         // keys per wave-instruction
    constexpr int UK = 8;                 // independent key loads in flight per lane in the score pass
    constexpr int NR = HD / 8;            // d-row groups (8 rows per wave-instruction) in the PV pass
    __shared__ __attribute__((aligned(16))) float sc[DEC_MAX_CTX];
    __shared__ __attribute__((aligned(16))) float q_s[HD];
    __shared__ __attribute__((aligned(16))) float part[8][HD];
    __shared__ float red[8];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = blockIdx.x, b = blockIdx.y;
    const size_t bh = (size_t)b * p.H + h;
    if (p.active_dev != nullptr && p.active_dev[(size_t)b * p.pos_stride] == 0) return;  // a free row of the decode pool
    const int pos = p.pos_dev[(size_t)b * p.pos_stride];
    const int ctx = pos + 1;
    const int D = p.H * HD;
    bf16_t* kbase = p.k + bh * p.kv_stride * HD;
    bf16_t* vbase = p.vt + bh * HD * (size_t)p.kv_stride;
    // ---- phase 0: rotate q,k of the new token, append k / v to the cache (global) and keep q in LDS
    if (tid < HD / 2) {
        const int d = tid;
        const bf16_t* row = p.qkv + (size_t)b * (3 * D) + h * HD;
        const float c = p.rope_cos[(size_t)pos * (HD / 2) + d], s = p.rope_sin[(size_t)pos * (HD / 2) + d];
        const float q0 = bf2f(row[d]), q1 = bf2f(row[d + HD / 2]);
        const float k0 = bf2f(row[D + d]), k1 = bf2f(row[D + d + HD / 2]);
        q_s[d] = bf2f(f2bf(q0 * c - q1 * s));            // q is rounded to bf16 exactly like the unfused path
        q_s[d + HD / 2] = bf2f(f2bf(q1 * c + q0 * s));
        bf16_t* ko = kbase + (size_t)pos * HD;
        ko[d] = f2bf(k0 * c - k1 * s);
        ko[d + HD / 2] = f2bf(k1 * c + k0 * s);
        vbase[(size_t)d * p.kv_stride + pos] = row[2 * D + d];
        vbase[(size_t)(d + HD / 2) * p.kv_stride + pos] = row[2 * D + d + HD / 2];
    }
    __syncthreads();  // workgroup-scope release/acquire: the appended K row / V^T column are visible to this block
    // ---- phase 1: scores
    float qv[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) qv[e] = q_s[(lane % LPK) * 8 + e];
    const int ctx_pad = (ctx + 8 * KPW * UK - 1) / (8 * KPW * UK) * (8 * KPW * UK);
    for (int kb = wave * KPW * UK; kb < ctx_pad; kb += 8 * KPW * UK) {
        u32x4 kv[UK];
#pragma unroll
        for (int u = 0; u < UK; ++u) {
            const int key = kb + u * KPW + lane / LPK;
            kv[u] = ld16_stream(kbase + (size_t)min(key, ctx - 1) * HD + (lane % LPK) * 8);
        }
#pragma unroll
        for (int u = 0; u < UK; ++u) {
            const int key = kb + u * KPW + lane / LPK;
            float s = 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) s += qv[2 * e] * bf2f_lo(kv[u][e]) + qv[2 * e + 1] * bf2f_hi(kv[u][e]);
#pragma unroll
            for (int mk = 1; mk < LPK; mk <<= 1) s += shfl_xor(s, mk);
            if ((lane % LPK) == 0) sc[key] = key < ctx ? s * p.scale : -INFINITY;
        }
    }
    // the first V^T batch of every wave does not depend on the scores: request it now so HBM stays busy through the
    // LDS-only softmax below
    const int dr = lane >> 3, kc = lane & 7;
    const int ctx64 = (ctx + 63) & ~63;
    constexpr int NV = HALF ? NR / 2 : NR;  // row groups per register set
    u32x4 v0[NV], v1[NV];
    // address = wave-uniform row-group base (scalar registers) + ONE 32-bit per-lane offset shared by all the loads of a
    // set: 64-bit per-row pointers kept live across the loop are what spilled the half-set form
    const uint32_t v_lane_off = ((uint32_t)dr * (uint32_t)p.kv_stride + (uint32_t)kc * 8u) * 2u;
    // live == false (past the last block; wave-uniform): every lane re-reads one cached 16-byte word instead — a branch
    // around the loads makes the compiler rotate the loop and spill
    auto load_v = [&](u32x4 (&v)[NV], int kb, int h, bool live = true) {  // h: which half of the d rows (HALF) — 0 otherwise
        const uint32_t loff = live ? v_lane_off : 0u;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const char* ub = live ? reinterpret_cast<const char*>(vbase) + ((size_t)((h * NV + i) * 8) * p.kv_stride + kb) * 2
                                  : reinterpret_cast<const char*>(p.rope_cos);
            v[i] = ld16_stream(ub + loff);
        }
    };
    if (wave * 64 < ctx64) {
        load_v(v0, wave * 64, 0);
        if constexpr (HALF) load_v(v1, wave * 64, 1);
    }
    __syncthreads();
    // ---- phase 2: softmax over sc[0..ctx_pad)
    float mx = -INFINITY;
    for (int i = tid; i < ctx_pad; i += 512) mx = fmaxf(mx, sc[i]);
    mx = wave_max(mx);
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    mx = red[0];
#pragma unroll
    for (int w = 1; w < 8; ++w) mx = fmaxf(mx, red[w]);
    __syncthreads();
    float sum = 0.f;
    for (int i = tid; i < ctx_pad; i += 512) {
        const float e = __expf(sc[i] - mx);
        sc[i] = e;
        sum += e;
    }
    sum = wave_sum(sum);
    if (lane == 0) red[wave] = sum;
    __syncthreads();
    sum = 0.f;
#pragma unroll
    for (int w = 0; w < 8; ++w) sum += red[w];
    const float inv = 1.0f / sum;
    // ---- phase 3: waves split the 64-key blocks; per block a wave issues NR independent 16-byte V^T loads
    float acc[NR];
#pragma unroll
    for (int i = 0; i < NR; ++i) acc[i] = 0.f;
    auto pv = [&](const u32x4 (&v)[NV], int kb, int h) {
        const f32x4 p0 = ld16f(&sc[kb + kc * 8]), p1 = ld16f(&sc[kb + kc * 8 + 4]);
#pragma unroll
        for (int i = 0; i < NV; ++i)
            acc[h * NV + i] += p0[0] * bf2f_lo(v[i][0]) + p0[1] * bf2f_hi(v[i][0]) + p0[2] * bf2f_lo(v[i][1]) +
                               p0[3] * bf2f_hi(v[i][1]) + p1[0] * bf2f_lo(v[i][2]) + p1[1] * bf2f_hi(v[i][2]) +
                               p1[2] * bf2f_lo(v[i][3]) + p1[3] * bf2f_hi(v[i][3]);
    };
    if constexpr (HALF) {
        // v0 = rows [0, HD/2), v1 = rows [HD/2, HD) of the current block; each set is re-requested for the next block as soon
        // as it has been consumed, so one half is always in flight
        // (the scheduling fences keep the compiler from hoisting a set's next loads above the arithmetic that still reads
        // it — that needs a third set of registers and spilled)
        for (int kb = wave * 64; kb < ctx64; kb += 8 * 64) {
            const int nxt = kb + 8 * 64;
            pv(v0, kb, 0);
            sched_fence();
            load_v(v0, nxt, 0, nxt < ctx64);
            sched_fence();
            pv(v1, kb, 1);
            sched_fence();
            load_v(v1, nxt, 1, nxt < ctx64);
            sched_fence();
        }
    } else {
        // software-pipelined over 64-key blocks with two register sets: the next block's V^T loads are in flight while
        // this one accumulates
        for (int kb = wave * 64; kb < ctx64; kb += 2 * 8 * 64) {
            const int kb1 = kb + 8 * 64, kb2 = kb + 2 * 8 * 64;
            if (kb1 < ctx64) load_v(v1, kb1, 0);
            pv(v0, kb, 0);
            if (kb1 < ctx64) {
                if (kb2 < ctx64) load_v(v0, kb2, 0);
                pv(v1, kb1, 0);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < NR; ++i) {
        float a = acc[i];
        a += shfl_xor(a, 1);
        a += shfl_xor(a, 2);
        a += shfl_xor(a, 4);
        if (kc == 0) part[wave][i * 8 + dr] = a;
    }
    __syncthreads();
    if (tid < HD) {
        float a = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) a += part[w][tid];
        p.out[(size_t)b * D + h * HD + tid] = f2bf(a * inv);
    }
}

// =============================================================================================
// single-pass ("flash") form of the fused decode attention (contexts beyond DEC_MAX_CTX; VC_DATTN_VARIANT=1).  Same phase 0 (RoPE + append); then every wave
// walks its own 64-key blocks (wave, wave+8, ...) with an online softmax, so the K rows and V^T rows of a (b,h) are ONE
// continuous non-temporal stream — no workgroup-wide softmax barrier with HBM idle behind it, no context-length limit:
//   scores of the block (16 lanes per key)  ->  wave-private LDS  ->  running max / sum, rescale of the accumulators
//   ->  P·V^T with the block's probabilities (lane = 8 d-rows x 8 key-chunks)
// The next block's K rows are requested as soon as the scores are out of the registers, its V^T rows as soon as the PV
// sums are done, so 16-32 KiB per wave stay in flight.  The 8 waves' (max, sum, acc) are merged once at the end.
// =============================================================================================
template <int HD>
__global__ __launch_bounds__(512) void attention_decode_flash_kernel(AttnDecodeFusedArgs p) {
    constexpr int LPK = HD / 8;           // lanes per key
    constexpr int KPW = 64 / LPK;         // keys per wave-instruction
    constexpr int NKI = 64 / KPW;         // K instructions per 64-key block
    constexpr int NR = HD / 8;            // V^T instructions per block (8 d-rows each)
    __shared__ __attribute__((aligned(16))) float q_s[HD];
    __shared__ __attribute__((aligned(16))) float sc[8][64];
    __shared__ __attribute__((aligned(16))) float part[8][HD];
    __shared__ float ml[8][2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = blockIdx.x, b = blockIdx.y;
    const size_t bh = (size_t)b * p.H + h;
    if (p.active_dev != nullptr && p.active_dev[(size_t)b * p.pos_stride] == 0) return;  // a free row of the decode pool
    const int pos = p.pos_dev[(size_t)b * p.pos_stride];
    const int ctx = pos + 1;
    const int D = p.H * HD;
    bf16_t* kbase = p.k + bh * p.kv_stride * HD;
    bf16_t* vbase = p.vt + bh * HD * (size_t)p.kv_stride;
    if (tid < HD / 2) {  // phase 0: rotate q,k of the new token, append k / v (global), keep q in LDS
        const int d = tid;
        const bf16_t* row = p.qkv + (size_t)b * (3 * D) + h * HD;
        const float c = p.rope_cos[(size_t)pos * (HD / 2) + d], s = p.rope_sin[(size_t)pos * (HD / 2) + d];
        const float q0 = bf2f(row[d]), q1 = bf2f(row[d + HD / 2]);
        const float k0 = bf2f(row[D + d]), k1 = bf2f(row[D + d + HD / 2]);
        q_s[d] = bf2f(f2bf(q0 * c - q1 * s));
        q_s[d + HD / 2] = bf2f(f2bf(q1 * c + q0 * s));
        bf16_t* ko = kbase + (size_t)pos * HD;
        ko[d] = f2bf(k0 * c - k1 * s);
        ko[d + HD / 2] = f2bf(k1 * c + k0 * s);
        vbase[(size_t)d * p.kv_stride + pos] = row[2 * D + d];
        vbase[(size_t)(d + HD / 2) * p.kv_stride + pos] = row[2 * D + d + HD / 2];
    }
    __syncthreads();  // workgroup-scope release/acquire: the appended K row / V^T column are visible to this block
    float qv[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) qv[e] = q_s[(lane % LPK) * 8 + e];
    const int nblk = (ctx + 63) >> 6;
    const int dr = lane >> 3, kc = lane & 7;
    float m_run = -INFINITY, l_run = 0.f;
    float acc[NR];
#pragma unroll
    for (int i = 0; i < NR; ++i) acc[i] = 0.f;
    u32x4 kv[NKI], vv[NR];
    auto load_k = [&](int blk) {
#pragma unroll
        for (int u = 0; u < NKI; ++u) {
            const int key = blk * 64 + u * KPW + lane / LPK;
            kv[u] = ld16_stream(kbase + (size_t)min(key, ctx - 1) * HD + (lane % LPK) * 8);
        }
    };
    auto load_v = [&](int blk) {
#pragma unroll
        for (int i = 0; i < NR; ++i) vv[i] = ld16_stream(vbase + (size_t)(i * 8 + dr) * p.kv_stride + blk * 64 + kc * 8);
    };
    int blk = wave;
    if (blk < nblk) {
        load_k(blk);
        load_v(blk);
    }
    for (; blk < nblk; blk += 8) {
#pragma unroll
        for (int u = 0; u < NKI; ++u) {
            const int key = blk * 64 + u * KPW + lane / LPK;
            float s = 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) s += qv[2 * e] * bf2f_lo(kv[u][e]) + qv[2 * e + 1] * bf2f_hi(kv[u][e]);
#pragma unroll
            for (int mk = 1; mk < LPK; mk <<= 1) s += shfl_xor(s, mk);
            if ((lane % LPK) == 0) sc[wave][u * KPW + lane / LPK] = key < ctx ? s * p.scale : -INFINITY;
        }
        const int nxt = blk + 8;
        if (nxt < nblk) load_k(nxt);
        wave_lds_fence();
        const float s_l = sc[wave][lane];
        const float m_new = fmaxf(m_run, wave_max(s_l));  // finite: every block holds at least one live key
        const float corr = __expf(m_run - m_new);
        const float p_l = __expf(s_l - m_new);
        l_run = l_run * corr + wave_sum(p_l);
        m_run = m_new;
        wave_lds_fence();
        sc[wave][lane] = p_l;
        wave_lds_fence();
        const f32x4 p0 = ld16f(&sc[wave][kc * 8]), p1 = ld16f(&sc[wave][kc * 8 + 4]);
#pragma unroll
        for (int i = 0; i < NR; ++i)
            acc[i] = acc[i] * corr + (p0[0] * bf2f_lo(vv[i][0]) + p0[1] * bf2f_hi(vv[i][0]) + p0[2] * bf2f_lo(vv[i][1]) +
                                      p0[3] * bf2f_hi(vv[i][1]) + p1[0] * bf2f_lo(vv[i][2]) + p1[1] * bf2f_hi(vv[i][2]) +
                                      p1[2] * bf2f_lo(vv[i][3]) + p1[3] * bf2f_hi(vv[i][3]));
        wave_lds_fence();  // the probabilities are consumed before the next block's scores overwrite them
        if (nxt < nblk) load_v(nxt);
    }
#pragma unroll
    for (int i = 0; i < NR; ++i) {
        float a = acc[i];
        a += shfl_xor(a, 1);
        a += shfl_xor(a, 2);
        a += shfl_xor(a, 4);
        if (kc == 0) part[wave][i * 8 + dr] = a;
    }
    if (lane == 0) { ml[wave][0] = m_run; ml[wave][1] = l_run; }
    __syncthreads();
    if (tid < HD) {
        float M = ml[0][0];
#pragma unroll
        for (int w = 1; w < 8; ++w) M = fmaxf(M, ml[w][0]);
        float L = 0.f, a = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) {
            const float f = __expf(ml[w][0] - M);  // 0 for waves that saw no block
            L += ml[w][1] * f;
            a += part[w][tid] * f;
        }
        p.out[(size_t)b * D + h * HD + tid] = f2bf(a / L);
    }
}

void launch_attention_decode_fused(const AttnDecodeFusedArgs& a, hipStream_t s) {
    const dim3 grid(a.H, a.B), block(512);
    // default: the two-pass form (scores of the whole context in LDS, so the cache capacity must be <= DEC_MAX_CTX): two
    // long fully independent load phases stream better (33.0 us = 5.1 TB/s at B=8, ctx 1281) than the single-pass
    // online-softmax form (38.9 us: 2-3 serial score -> rescale -> PV chains per wave), which is used for longer
    // contexts and kept selectable with VC_DATTN_VARIANT=1
    static const int variant = getenv("VC_DATTN_VARIANT") ? atoi(getenv("VC_DATTN_VARIANT")) : 0;
    if ((variant == 0 || variant == 2) && a.kv_stride <= DEC_MAX_CTX) {
        // 0 (default): V^T in two half-sets, two workgroups per CU; 2: two full register sets, one workgroup per CU
        if (a.hd == 128) {
            if (variant == 0) VC_LAUNCH((attention_decode_fused_kernel<128, true>), grid, block, 0, s, a);
            else VC_LAUNCH((attention_decode_fused_kernel<128, false>), grid, block, 0, s, a);
        } else {
            VC_LAUNCH((attention_decode_fused_kernel<64, false>), grid, block, 0, s, a);   // 126 VGPRs already
        }
        return;
    }
    if (a.hd == 128) VC_LAUNCH((attention_decode_flash_kernel<128>), grid, block, 0, s, a);
    else VC_LAUNCH((attention_decode_flash_kernel<64>), grid, block, 0, s, a);
}

}  // namespace vc
