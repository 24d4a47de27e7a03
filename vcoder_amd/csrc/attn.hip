// attn.hip — attention for the VCoder hot path on gfx950.
//
//   qkv_split   : fused-QKV GEMM output -> Q [B,H,Tq,hd], K cache [B,H,S,hd], V^T scratch [B,H,hd,S] in the flash kernel's
//                 key order (vt_chunk_key0 below) (+RoPE)
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

// Key order of the V^T scratch (the flash kernel's A operand of O^T = V^T P^T).  Inside every aligned block of 32 keys the
// 16-byte chunk c (0..3) of a row holds keys {4c..4c+3, 16+4c..16+4c+3}: position 8c + e <-> key 4c + (e & 3) + 16 (e >> 2)
// (vt_key_pos in tests/kernel_cases.py).  That is the order in which the S^T accumulators of a lane (MFMA C layout: sub-tile
// `sub`, lane group g, register r <-> key 16 sub + 4g + r) are packed into the P operand, so a lane's 8 contraction values of
// a 32-key half are one contiguous chunk.  The scratch is written and read only by the kernels of this file.
VC_DEV int vt_chunk_key0(int chunk_in_tile) { return (chunk_in_tile >> 2) * 32 + (chunk_in_tile & 3) * 4; }

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
                    float a0, b0, a1, b1;   // out[d] = x*cos - y*sin ; out[d+hd/2] = y*cos + x*sin
                    rope_pair(x0, y0, c0, s0, a0, b0);
                    rope_pair(x1, y1, c1, s1, a1, b1);
                    olo[e] = pack_bf2(a0, a1);
                    ohi[e] = pack_bf2(b0, b1);
                }
            }
            bf16_t* dst = which == 0 ? p.q + (((size_t)b * p.H + h) * p.q_stride + t) * HD
                                     : p.k + (((size_t)b * p.H + h) * p.kv_stride + t) * HD;
            st16(dst + c * 8, olo);
            st16(dst + HD / 2 + c * 8, ohi);
            if (which == 1 && p.k8 != nullptr) {   // e4m3 cache row: the bf16 values above, re-rounded
                uint8_t* d8 = p.k8 + (((size_t)b * p.H + h) * p.kv8_stride + t) * HD;
                st8(d8 + c * 8, u32x2{f32x4_to_fp8x4(bf2f_lo(olo[0]), bf2f_hi(olo[0]), bf2f_lo(olo[1]), bf2f_hi(olo[1])),
                                      f32x4_to_fp8x4(bf2f_lo(olo[2]), bf2f_hi(olo[2]), bf2f_lo(olo[3]), bf2f_hi(olo[3]))});
                st8(d8 + HD / 2 + c * 8, u32x2{f32x4_to_fp8x4(bf2f_lo(ohi[0]), bf2f_hi(ohi[0]), bf2f_lo(ohi[1]), bf2f_hi(ohi[1])),
                                               f32x4_to_fp8x4(bf2f_lo(ohi[2]), bf2f_hi(ohi[2]), bf2f_lo(ohi[3]), bf2f_hi(ohi[3]))});
            }
        }
    }
    // V: rows as they are into the cache (key-major, what the decode steps stream); and a staged [token][d] tile written
    // back as V^T rows, 16 bytes (8 tokens) at a time, for the flash kernel of THIS prefill / ViT layer
    const int vts = p.vt_stride > 0 ? p.vt_stride : p.kv_stride;
    for (int w = tid; w < 64 * (HD / 8); w += 256) {
        const int tl = w / (HD / 8), c = w % (HD / 8), t = t0 + tl;
        u32x4 v = {0u, 0u, 0u, 0u};
        if (t < p.T) {
            v = ld16(p.qkv + ((size_t)b * p.T + t) * (3 * D) + 2 * D + h * HD + c * 8);
            if (p.v != nullptr) st16(p.v + (((size_t)b * p.H + h) * p.kv_stride + t) * HD + c * 8, v);
            if (p.v8 != nullptr)
                st8(p.v8 + (((size_t)b * p.H + h) * p.kv8_stride + t) * HD + c * 8,
                    u32x2{f32x4_to_fp8x4(bf2f_lo(v[0]), bf2f_hi(v[0]), bf2f_lo(v[1]), bf2f_hi(v[1])),
                          f32x4_to_fp8x4(bf2f_lo(v[2]), bf2f_hi(v[2]), bf2f_lo(v[3]), bf2f_hi(v[3]))});
        }
        st16(&vt_tile[tl][c * 8], v);
    }
    __syncthreads();
    for (int w = tid; w < HD * 8; w += 256) {
        const int d = w >> 3, tc = w & 7;
        const int k0 = vt_chunk_key0(tc);      // chunk tc of the tile holds keys k0..k0+3 and k0+16..k0+19 (vt_key_pos)
        if (t0 + k0 >= p.T) continue;
        uint32_t o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int key = k0 + (e >> 1) * 16 + (e & 1) * 2;
            o[e] = (uint32_t)vt_tile[key][d] | ((uint32_t)vt_tile[key + 1][d] << 16);
        }
        st16(p.vt + (((size_t)b * p.H + h) * HD + d) * vts + t0 + tc * 8, u32x4{o[0], o[1], o[2], o[3]});
    }
}

// ---- precision mode "split": fp32 projection output -> RoPE in fp32 -> (i) the fp32 K / V cache rows the decode steps read,
// (ii) bf16 hi / lo planes of Q, K and V^T for the MFMA flash kernel of THIS prefill / ViT layer (x = hi + lo to ~16 bits)
VC_DEV void split8(const float* v, u32x4& hi, u32x4& lo) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        hi[e] = pack_bf2(v[2 * e], v[2 * e + 1]);
        lo[e] = pack_bf2(v[2 * e] - bf2f_lo(hi[e]), v[2 * e + 1] - bf2f_hi(hi[e]));
    }
}
template <int HD>
__global__ __launch_bounds__(256) void qkv_split32_kernel(QkvSplit32Args p) {
    const float* __restrict__ rope_cos = p.rope_cos;
    const float* __restrict__ rope_sin = p.rope_sin;
    __shared__ __attribute__((aligned(16))) bf16_t vt_tile[2][64][HD + 8];  // [hi / lo][token][d]
    const int t0 = blockIdx.x * 64, h = blockIdx.y, b = blockIdx.z;
    const int tid = threadIdx.x;
    const int D = p.H * HD;
    constexpr int CPT = HD / 16;  // rope work items per token: chunk c pairs d0 = 8c with d0 + HD/2
    const bool rope = rope_cos != nullptr;
    const size_t bh = (size_t)b * p.H + h;
    for (int w = tid; w < 64 * CPT; w += 256) {
        const int tl = w / CPT, c = w % CPT, t = t0 + tl;
        if (t >= p.T) continue;
        const float* row = p.qkv + ((size_t)b * p.T + t) * (3 * D) + h * HD;
#pragma unroll
        for (int which = 0; which < 2; ++which) {  // 0 = q, 1 = k
            const float* src = row + which * D;
            float x[8], y[8], ox[8], oy[8];
            *reinterpret_cast<f32x4*>(x) = ld16f(src + c * 8);
            *reinterpret_cast<f32x4*>(x + 4) = ld16f(src + c * 8 + 4);
            *reinterpret_cast<f32x4*>(y) = ld16f(src + HD / 2 + c * 8);
            *reinterpret_cast<f32x4*>(y + 4) = ld16f(src + HD / 2 + c * 8 + 4);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float cs = 1.f, sn = 0.f;
                if (rope) {
                    cs = rope_cos[(size_t)t * (HD / 2) + c * 8 + e];
                    sn = rope_sin[(size_t)t * (HD / 2) + c * 8 + e];
                }
                ox[e] = x[e] * cs - y[e] * sn;   // out[d] = x*cos - y*sin ; out[d+hd/2] = y*cos + x*sin
                oy[e] = y[e] * cs + x[e] * sn;
            }
            u32x4 xh, xl, yh, yl;
            split8(ox, xh, xl);
            split8(oy, yh, yl);
            const size_t off = which == 0 ? (bh * p.q_stride + t) * HD : (bh * p.ks_stride + t) * HD;
            bf16_t* dh = (which == 0 ? p.q_hi : p.k_hi) + off;
            bf16_t* dl = (which == 0 ? p.q_lo : p.k_lo) + off;
            st16(dh + c * 8, xh);
            st16(dl + c * 8, xl);
            st16(dh + HD / 2 + c * 8, yh);
            st16(dl + HD / 2 + c * 8, yl);
            if (which == 1 && p.k32 != nullptr) {
                if (p.kv24) {   // fp24 cache row: hd x u16 hi plane | hd x u8 lo plane
                    char* kc = reinterpret_cast<char*>(p.k32) + (bh * p.kv_stride + t) * (size_t)(3 * HD);
                    u32x4 h8;
                    u32x2 l8;
                    pack_f24x8(ox, h8, l8);
                    st16(kc + c * 16, h8);
                    st8(kc + 2 * HD + c * 8, l8);
                    pack_f24x8(oy, h8, l8);
                    st16(kc + HD + c * 16, h8);
                    st8(kc + 2 * HD + HD / 2 + c * 8, l8);
                } else {
                    float* kc = p.k32 + (bh * p.kv_stride + t) * HD;
                    st16f(kc + c * 8, *reinterpret_cast<f32x4*>(ox));
                    st16f(kc + c * 8 + 4, *reinterpret_cast<f32x4*>(ox + 4));
                    st16f(kc + HD / 2 + c * 8, *reinterpret_cast<f32x4*>(oy));
                    st16f(kc + HD / 2 + c * 8 + 4, *reinterpret_cast<f32x4*>(oy + 4));
                }
            }
        }
    }
    for (int w = tid; w < 64 * (HD / 8); w += 256) {
        const int tl = w / (HD / 8), c = w % (HD / 8), t = t0 + tl;
        float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (t < p.T) {
            const float* src = p.qkv + ((size_t)b * p.T + t) * (3 * D) + 2 * D + h * HD + c * 8;
            *reinterpret_cast<f32x4*>(v) = ld16f(src);
            *reinterpret_cast<f32x4*>(v + 4) = ld16f(src + 4);
            if (p.v32 != nullptr) {
                if (p.kv24) {
                    char* vc = reinterpret_cast<char*>(p.v32) + (bh * p.kv_stride + t) * (size_t)(3 * HD);
                    u32x4 h8;
                    u32x2 l8;
                    pack_f24x8(v, h8, l8);
                    st16(vc + c * 16, h8);
                    st8(vc + 2 * HD + c * 8, l8);
                } else {
                    float* vc = p.v32 + (bh * p.kv_stride + t) * HD + c * 8;
                    st16f(vc, *reinterpret_cast<f32x4*>(v));
                    st16f(vc + 4, *reinterpret_cast<f32x4*>(v + 4));
                }
            }
        }
        u32x4 vh, vl;
        split8(v, vh, vl);
        st16(&vt_tile[0][tl][c * 8], vh);
        st16(&vt_tile[1][tl][c * 8], vl);
    }
    __syncthreads();
    for (int w = tid; w < 2 * HD * 8; w += 256) {
        const int pl = w / (HD * 8), d = (w >> 3) % HD, tc = w & 7;
        const int k0 = vt_chunk_key0(tc);
        if (t0 + k0 >= p.T) continue;
        uint32_t o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int key = k0 + (e >> 1) * 16 + (e & 1) * 2;
            o[e] = (uint32_t)vt_tile[pl][key][d] | ((uint32_t)vt_tile[pl][key + 1][d] << 16);
        }
        st16((pl ? p.vt_lo : p.vt_hi) + (bh * HD + d) * p.vt_stride + t0 + tc * 8, u32x4{o[0], o[1], o[2], o[3]});
    }
}
void launch_qkv_split32(const QkvSplit32Args& a, hipStream_t s) {
    const dim3 grid((a.T + 63) / 64, a.H, a.B), block(256);
    if (a.hd == 128) VC_LAUNCH((qkv_split32_kernel<128>), grid, block, 0, s, a);
    else VC_LAUNCH((qkv_split32_kernel<64>), grid, block, 0, s, a);
}

void launch_qkv_split(const QkvSplitArgs& a, hipStream_t s) {
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

// SPLIT (precision mode "split"): every operand comes as two bf16 planes (x = hi + lo, ~16 mantissa bits): S^T = Kh Qh + Kh Ql
// + Kl Qh, the softmax numerators are split the same way in registers, O^T += Vh Ph + Vh Pl + Vl Ph (the lo x lo terms are
// below 2^-18 relative), and the output row is written as [hi | lo] at columns h*HD and lo_off + h*HD of a row of ldo
// elements.  Twice the LDS per stage (dynamic shared memory: 128 KiB at hd 128, one workgroup per CU).
template <int HD, bool CAUSAL, int WAVES, int QS, bool SPLIT = false>
__global__ __launch_bounds__(WAVES * 64, (WAVES == 8 && QS == 1 && !SPLIT) ? 4 : 1) void attention_kernel(AttnArgs p) {  // 8 x 16: two workgroups per CU (<= 128 VGPRs)
    constexpr int NTH = WAVES * 64;       // threads per workgroup
    constexpr int QB = WAVES * QS * 16;   // queries per workgroup
    constexpr int KS = HD / 32;        // k-steps of the QK^T contraction
    constexpr int DT = HD / 16;        // 16-wide d tiles of the output
    constexpr int KCH = HD / 8;        // 16-B chunks per K row
    constexpr int KB_ = 64 * HD * 2, VB_ = HD * 128;   // bytes of a K tile [64 keys][HD] and a V^T tile [HD][64 keys]
    constexpr int PL = SPLIT ? 2 : 1;                  // operand planes
    constexpr int STAGE = PL * (KB_ + VB_);            // [K hi | K lo | V^T hi | V^T lo]
    char* lds0;                                        // [2][STAGE], double-buffered: the LDS-DMA of tile t+1 lands under tile t
#ifdef VC_EMU   // the emulator's guard page, NaN poison and race check cover DYNAMIC LDS: the test build takes both forms from it
    VC_DYNAMIC_SMEM(char, lds_dyn);
    lds0 = lds_dyn;
#else
    if constexpr (SPLIT) {
        VC_DYNAMIC_SMEM(char, lds_dyn);
        lds0 = lds_dyn;
    } else {
        __shared__ __attribute__((aligned(16))) char lds_st[2 * STAGE];
        lds0 = lds_st;
    }
#endif
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, j = lane & 15;
    // Workgroup order.  sched 0: the 3-D grid as launched (query block fastest).  Otherwise a 1-D grid whose linear id — the
    // hardware hands consecutive ids to consecutive XCDs — is mapped so that (1, 3) every XCD owns a contiguous range of
    // (b, h) pairs: the query blocks of a head run on ONE XCD and its K / V^T tiles are fetched into that L2 once, not into
    // eight; and (1, 2) the query blocks of a causal pass are taken heaviest first (block i has ~2 (i + 1) KV tiles), so the
    // launch ends on the light ones instead of a few 19-tile stragglers.
    int qb_ = blockIdx.x, h_ = blockIdx.y, b_ = blockIdx.z;
    if (p.sched != 0) {
        const int lin = blockIdx.x, BH = p.B * p.H;
        int head, qr;
        if (p.sched == 2) {
            head = lin % BH;
            qr = lin / BH;
        } else {
            const int xcd = lin & 7, r = lin >> 3, per = BH >> 3;   // the launcher checks BH % 8 == 0
            head = xcd * per + r / p.nqb;
            qr = r % p.nqb;
        }
        qb_ = (CAUSAL && p.sched != 3) ? p.nqb - 1 - qr : qr;
        h_ = head % p.H;
        b_ = head / p.H;
    }
    const int q0 = qb_ * QB, h = h_, b = b_;
    const size_t bh = (size_t)b * p.H + h;
    const bf16_t* qbase = p.q + bh * p.q_stride * HD;
    const bf16_t* kbase = p.k + bh * p.kv_stride * HD;
    const int vts = p.vt_stride > 0 ? p.vt_stride : p.kv_stride;
    const bf16_t* vbase = p.vt + bh * HD * (size_t)vts;
    const bf16_t* qbase_lo = SPLIT ? p.q_lo + bh * p.q_stride * HD : nullptr;
    const bf16_t* kbase_lo = SPLIT ? p.k_lo + bh * p.kv_stride * HD : nullptr;
    const bf16_t* vbase_lo = SPLIT ? p.vt_lo + bh * HD * (size_t)vts : nullptr;

    // Q fragments (MFMA B operand): lane holds Q[query j][d = ks*32 + g*8 .. +8]
    u32x4 qf[QS][KS], qfl[SPLIT ? QS : 1][SPLIT ? KS : 1];
#pragma unroll
    for (int qs = 0; qs < QS; ++qs) {
        const int qrow = min(q0 + wave * (QS * 16) + qs * 16 + j, p.T - 1);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            qf[qs][ks] = ld16(qbase + (size_t)qrow * HD + ks * 32 + g * 8);
            if constexpr (SPLIT) qfl[qs][ks] = ld16(qbase_lo + (size_t)qrow * HD + ks * 32 + g * 8);
        }
    }
    f32x4 o[QS][DT];
#pragma unroll
    for (int qs = 0; qs < QS; ++qs)
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) o[qs][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m_run[QS], l_run[QS];
#pragma unroll
    for (int qs = 0; qs < QS; ++qs) { m_run[qs] = -INFINITY; l_run[qs] = 0.f; }

    const float c2 = p.scale * 1.4426950408889634f;   // scale * log2(e): exp(scale * x) = exp2(c2 * x)
    const int kv_end = CAUSAL ? min(p.T, q0 + QB) : p.T;
    const int nkt = (kv_end + 63) / 64;
    // K / V^T tiles go global -> LDS by DMA (global_load_lds_dwordx4: no staging VGPRs, no ds_write pass), 1-KiB pieces =
    // 64 lanes x 16 B written linearly; the XOR swizzles of swz_k / swz_v are applied on the SOURCE side: the lane that
    // fills slot s of row r fetches chunk s ^ f(r) of that row.
    constexpr int KROWS = 1024 / (HD * 2);      // K rows per piece (4 at hd 128, 8 at hd 64)
    constexpr int KPIECES = KB_ / 1024 / WAVES, VPIECES = VB_ / 1024 / WAVES;
    static_assert(KPIECES >= 1 && VPIECES >= 1 && KB_ % (1024 * WAVES) == 0 && VB_ % (1024 * WAVES) == 0, "pieces per wave");
    const char* ksrc[KPIECES];
    const char* vsrc[VPIECES];
    size_t koff[KPIECES], voff[VPIECES];   // element offsets inside a (b,h) plane: the lo planes share them
#pragma unroll
    for (int i = 0; i < KPIECES; ++i) {
        const int piece = i * WAVES + wave, row = piece * KROWS + lane / KCH, slot = lane % KCH;
        const int ch = HD == 128 ? (slot ^ (row & 15)) : (slot ^ (row & 7));
        koff[i] = (size_t)row * HD + ch * 8;
        ksrc[i] = reinterpret_cast<const char*>(kbase + koff[i]);
    }
#pragma unroll
    for (int i = 0; i < VPIECES; ++i) {
        const int piece = i * WAVES + wave, row = piece * 8 + (lane >> 3), slot = lane & 7;
        voff[i] = (size_t)row * vts + ((slot ^ ((row >> 1) & 7)) * 8);
        vsrc[i] = reinterpret_cast<const char*>(vbase + voff[i]);
    }
    // stage layout: K hi at 0, [K lo at KB_,] V^T hi at PL*KB_, [V^T lo at PL*KB_ + VB_]
    auto issue_tile = [&](int kt, int buf) {
        char* st = lds0 + buf * STAGE;
#pragma unroll
        for (int i = 0; i < KPIECES; ++i) {
            glds16(ksrc[i] + (size_t)kt * (64 * HD * 2), st + (i * WAVES + wave) * 1024);
            if constexpr (SPLIT)
                glds16(reinterpret_cast<const char*>(kbase_lo + koff[i]) + (size_t)kt * (64 * HD * 2), st + KB_ + (i * WAVES + wave) * 1024);
        }
#pragma unroll
        for (int i = 0; i < VPIECES; ++i) {
            glds16(vsrc[i] + (size_t)kt * 128, st + PL * KB_ + (i * WAVES + wave) * 1024);
            if constexpr (SPLIT)
                glds16(reinterpret_cast<const char*>(vbase_lo + voff[i]) + (size_t)kt * 128, st + PL * KB_ + VB_ + (i * WAVES + wave) * 1024);
        }
    };

    const int w_first = q0 + wave * (QS * 16);   // smallest query of this wave
    issue_tile(0, 0);
    for (int kt = 0; kt < nkt; ++kt) {
        // tile kt has landed for every wave (own vmcnt wait + barrier) and every wave is done with tile kt-1, whose buffer
        // the next DMA overwrites
        wait_vmcnt<0>();
        __syncthreads();
        if (kt + 1 < nkt) issue_tile(kt + 1, (kt + 1) & 1);
        const char* k_lds = lds0 + (kt & 1) * STAGE;
        const char* v_lds = k_lds + PL * KB_;
        const int k0 = kt * 64;
        // Tiles this wave has nothing to do in (it still takes part in the DMA and the barrier above): every key of the tile
        // lies in the future of all its queries — the last tile of a causal block for the lower half of its waves — or all
        // its queries lie behind the sequence (the ragged last block: S = 1216 leaves waves 4..7 of block 9, the 19-tile one,
        // without a row).  Together 13 % of the wave-tiles of the prefill shape; the skipped work contributed exact zeros.
        if (w_first >= p.T || (CAUSAL && k0 > w_first + QS * 16 - 1)) continue;
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
                if constexpr (SPLIT) {
                    const u32x4 kl = ld16(k_lds + KB_ + swz_k<HD>(sub * 16 + j, ks * 4 + g));
#pragma unroll
                    for (int qs = 0; qs < QS; ++qs) {
                        sacc[qs][sub] = mfma16(kf, qfl[qs][ks], sacc[qs][sub]);
                        sacc[qs][sub] = mfma16(kl, qf[qs][ks], sacc[qs][sub]);
                    }
                }
            }
        // ---- online softmax (lane owns query j of each q-subtile; keys spread over regs and the 4 lane groups).  The
        // kernel is bound by these VALU / transcendental instructions, not by its MFMAs (per query-key pair: 256 MACs =
        // 0.25 MFMA cycles against one v_exp_f32 and the arithmetic around it), so the per-element work is the minimum:
        // the running maximum is kept on the RAW scores (scale > 0 commutes with max), scale * log2(e) is folded into the
        // one FMA that feeds v_exp_f32, and the key mask is only evaluated in tiles that contain a masked key (the
        // diagonal tiles of a causal pass, the ragged last tile) — a wave-uniform branch.
        u32x4 pb[QS][2], pbl[SPLIT ? QS : 1][2];
#pragma unroll
        for (int qs = 0; qs < QS; ++qs) {
            const int qfirst = q0 + wave * (QS * 16) + qs * 16;     // smallest query of this sub-tile
            const int query = qfirst + j;
            if ((CAUSAL && k0 + 63 > qfirst) || k0 + 64 > p.T || p.key_mask != nullptr) {
                // key_mask (padded batches): a key hidden by the caller's attention_mask is hidden from EVERY query of its
                // sequence, exactly as the additive mask HF builds from the 2-D mask ([HF] llama/modeling_llama.py causal mask +
                // padding mask); key 0 of a sequence is always visible (the engine checks), so the running maximum is finite
                const uint8_t* km = p.key_mask != nullptr ? p.key_mask + (size_t)b * p.mask_stride : nullptr;
                if (km == nullptr) {   // the common case: sixteen selects, no branches
#pragma unroll
                    for (int sub = 0; sub < 4; ++sub)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int key = k0 + sub * 16 + g * 4 + r;
                            const bool hide = key >= p.T || (CAUSAL && key > query);
                            sacc[qs][sub][r] = hide ? -INFINITY : sacc[qs][sub][r];
                        }
                } else {
#pragma unroll
                    for (int sub = 0; sub < 4; ++sub)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int key = k0 + sub * 16 + g * 4 + r;
                            if (key >= p.T || (CAUSAL && key > query) || km[key] == 0) sacc[qs][sub][r] = -INFINITY;
                        }
                }
            }
            // v_max3 on the raw accumulators (fmaxf would canonicalise every MFMA output first: 16 extra VALU per tile)
            float mx = vmax3(sacc[qs][0][0], sacc[qs][0][1], sacc[qs][0][2]);
            mx = vmax3(mx, sacc[qs][0][3], sacc[qs][1][0]);
            mx = vmax3(mx, sacc[qs][1][1], sacc[qs][1][2]);
            mx = vmax3(mx, sacc[qs][1][3], sacc[qs][2][0]);
            mx = vmax3(mx, sacc[qs][2][1], sacc[qs][2][2]);
            mx = vmax3(mx, sacc[qs][2][3], sacc[qs][3][0]);
            mx = vmax3(mx, sacc[qs][3][1], sacc[qs][3][2]);
            mx = vmax2(mx, sacc[qs][3][3]);
            mx = rows_max(mx);
            // Deferred maximum: the reference point m of exp(scale * (s - m)) only has to keep the exponentials in range, it
            // need not BE the maximum.  It moves (and O, l are rescaled) only when some query of the sub-tile exceeds it by more
            // than DEFER — a wave-uniform branch that is rarely taken once the first tiles are done; in between P <= e^8, exact
            // in fp32 and with the same relative bf16 rounding.  Precision mode "split" keeps the threshold at 0: the reference
            // point is then the running maximum itself and skipping the no-op rescale (alpha == 1) changes no bit.
            constexpr float DEFER = SPLIT ? 0.f : 11.5f;            // in log2 units: e^8
            const bool grew = wave_any((mx - m_run[qs]) * c2 > DEFER);   // first tile: m_run = -inf
            const float m_new = grew ? vmax2(m_run[qs], mx) : m_run[qs];   // raw-score units; finite from the first tile on (key 0)
            const float alpha = grew ? fast_exp2((m_run[qs] - m_new) * c2) : 1.0f;  // first tile: exp2(-inf) = 0
            m_run[qs] = m_new;
            const float mc = m_new * c2;
            float rs = 0.f;
#pragma unroll
            for (int sub = 0; sub < 4; ++sub)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float e = fast_exp2(fmaf(sacc[qs][sub][r], c2, -mc));   // = exp(scale * (s - m)); masked: 0
                    sacc[qs][sub][r] = e;
                    rs += e;
                }
            if (grew) {
                l_run[qs] = l_run[qs] * alpha;
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) o[qs][dt] = o[qs][dt] * alpha;
            }
            l_run[qs] += rs;
#pragma unroll
            for (int kh = 0; kh < 2; ++kh)
                pb[qs][kh] = u32x4{pack_bf2(sacc[qs][2 * kh][0], sacc[qs][2 * kh][1]),
                                   pack_bf2(sacc[qs][2 * kh][2], sacc[qs][2 * kh][3]),
                                   pack_bf2(sacc[qs][2 * kh + 1][0], sacc[qs][2 * kh + 1][1]),
                                   pack_bf2(sacc[qs][2 * kh + 1][2], sacc[qs][2 * kh + 1][3])};
            if constexpr (SPLIT) {
#pragma unroll
                for (int kh = 0; kh < 2; ++kh)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const f32x4& sv = sacc[qs][2 * kh + (e >> 1)];
                        const uint32_t hi = pb[qs][kh][e];
                        pbl[qs][kh][e] = pack_bf2(sv[2 * (e & 1)] - bf2f_lo(hi), sv[2 * (e & 1) + 1] - bf2f_hi(hi));
                    }
            }
        }
        // ---- O^T += V^T P^T   (contraction slot (g,e) <-> key kh*32 + (e>>2)*16 + 4g + (e&3) on both operands: the V^T scratch
        // stores the keys of every 32-key block in exactly that order, so a lane's 8 contraction values are ONE 16-byte chunk —
        // one conflict-free ds_read_b128 per MFMA.  Two 8-byte reads of a key-ordered tile were merged by the compiler into
        // ds_read2st64_b64, which banks mod 32 in contiguous 16-lane groups: 39 % of the kernel's LDS cycles were conflicts)
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int kh = 0; kh < 2; ++kh) {
                const int row = dt * 16 + j;
                const u32x4 vf = ld16(v_lds + swz_v(row, kh * 4 + g));   // keys 32 kh + {4g..4g+3, 16+4g..16+4g+3}: vt_key_pos
#pragma unroll
                for (int qs = 0; qs < QS; ++qs) o[qs][dt] = mfma16(vf, pb[qs][kh], o[qs][dt]);
                if constexpr (SPLIT) {
                    const u32x4 vl = ld16(v_lds + VB_ + swz_v(row, kh * 4 + g));
#pragma unroll
                    for (int qs = 0; qs < QS; ++qs) {
                        o[qs][dt] = mfma16(vf, pbl[qs][kh], o[qs][dt]);
                        o[qs][dt] = mfma16(vl, pb[qs][kh], o[qs][dt]);
                    }
                }
            }
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
            const size_t ldo = SPLIT ? (size_t)p.ldo : (size_t)p.H * HD;
            bf16_t* dst = p.out + ((size_t)b * p.T + query) * ldo + h * HD + g * 4;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                const f32x4 v = o[qs][dt] * inv;
                const u32x2 hi = {pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3])};
                st8(dst + dt * 16, hi);
                if constexpr (SPLIT)
                    st8(dst + p.lo_off + dt * 16, u32x2{pack_bf2(v[0] - bf2f_lo(hi[0]), v[1] - bf2f_hi(hi[0])),
                                                        pack_bf2(v[2] - bf2f_lo(hi[1]), v[3] - bf2f_hi(hi[1]))});
            }
        }
    }
}

// workgroup order (AttnArgs::sched, see attention_kernel): 0 = plain 3-D grid, 1 = XCD-grouped heads + heaviest query block first,
// 2 = heaviest first only, 3 = XCD-grouped only
static dim3 attention_grid(AttnArgs& a, int QB) {
    // measured (profiles/r03_l_*): causal prefill shape 2 < 3 < 0 = 1 (165-175 / 179-187 / 186-200 us); ViT shape 3 < 0 < 2
    // (78-85 / 85-92 / 88-94 us) — 2 for a causal pass, 3 otherwise
    a.nqb = (a.T + QB - 1) / QB;
    a.sched = a.causal ? 2 : 3;
    if ((a.sched == 1 || a.sched == 3) && (a.B * a.H) % 8 != 0) a.sched = a.sched == 1 ? 2 : 0;
    return a.sched ? dim3(a.nqb * a.H * a.B) : dim3(a.nqb, a.H, a.B);
}

template <int HD, int WAVES, int QS>
static void launch_attention_v(const AttnArgs& a0, hipStream_t s) {
    const int QB = WAVES * QS * 16;
    AttnArgs a = a0;
    const dim3 grid = attention_grid(a, QB), block(WAVES * 64);
#ifdef VC_EMU
    constexpr size_t shmem = 2 * (64 * HD * 2 + HD * 128);   // the two stages the product build holds in static LDS
#else
    constexpr size_t shmem = 0;
#endif
    if (a.causal) VC_LAUNCH((attention_kernel<HD, true, WAVES, QS>), grid, block, shmem, s, a);
    else VC_LAUNCH((attention_kernel<HD, false, WAVES, QS>), grid, block, shmem, s, a);
}

template <int HD>
static void launch_attention_split(const AttnArgs& a0, hipStream_t s) {
    AttnArgs a = a0;
    const dim3 grid = attention_grid(a, 128), block(512);
    constexpr size_t shmem = 2 * 2 * (64 * HD * 2 + HD * 128);   // [2 stages][K hi | K lo | V^T hi | V^T lo]
#ifndef VC_EMU
    static bool once = false;
    if (!once) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attention_kernel<HD, true, 8, 1, true>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attention_kernel<HD, false, 8, 1, true>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
        once = true;
    }
#endif
    if (a.causal) VC_LAUNCH((attention_kernel<HD, true, 8, 1, true>), grid, block, shmem, s, a);
    else VC_LAUNCH((attention_kernel<HD, false, 8, 1, true>), grid, block, shmem, s, a);
}

void launch_attention(const AttnArgs& a, hipStream_t s) {
    if (a.q_lo != nullptr) {  // precision mode "split": hi / lo operand planes
        if (a.hd == 128) launch_attention_split<128>(a, s);
        else launch_attention_split<64>(a, s);
        return;
    }
    // 8 waves x 16 queries keeps the hd-128 kernel at 128 VGPRs (the 4x32 form needs ~250 -> 1 wave/SIMD); measured on
    // MI355X: prefill (hd 128, T 1216, causal) 288 us vs 378 us; ViT (hd 64, T 577) 97 us vs 117 us.
    static const int variant = getenv("VC_ATTN_VARIANT") ? atoi(getenv("VC_ATTN_VARIANT")) : 0;
    // variant 2: 8 waves x 32 queries (256 queries per workgroup): every K / V^T fragment read from LDS feeds two MFMAs —
    // the 8 x 16 form reads 32 KiB of fragments per wave and 64-key tile for 32 MFMAs, i.e. it is bound by LDS bandwidth
    if (a.hd == 128) {
        if (variant == 1) launch_attention_v<128, 4, 2>(a, s);
        else if (variant == 2) launch_attention_v<128, 8, 2>(a, s);
        else launch_attention_v<128, 8, 1>(a, s);
    } else {
        if (variant == 1) launch_attention_v<64, 4, 2>(a, s);
        else if (variant == 2) launch_attention_v<64, 8, 2>(a, s);
        else launch_attention_v<64, 8, 1>(a, s);
    }
}

// =============================================================================================
// fused decode attention: RoPE + KV append + softmax(q K^T) V for the one new token of each (b,h).
// One 512-thread workgroup per (b,h).  The cache holds K AND V key-major ([S][hd] per (b,h)), so both passes read one
// contiguous stream of 2 * hd bytes per key — 16 lanes per key row, 4 rows per wave-instruction, 8 independent
// non-temporal loads in flight per lane — and the append is two contiguous 2*hd-byte rows.  (Round 1 kept V transposed
// for the prefill's MFMA flash kernel and streamed hd strided rows of ctx*2 bytes here: 5.1-5.3 TB/s; the prefill now
// gets its V^T tiles from a per-call scratch and the cache layout serves the decode steps, which read it 127 times.)
//   phase 0  rotate q / k of the new token, append the K and V rows, q to LDS
//   phase 1  scores -> LDS (<= DEC_MAX_CTX keys)          [K stream]
//   phase 2  block softmax in LDS (the first V batch is already in flight)
//   phase 3  out = sum_key p[key] * V[key]                  [V stream]; lanes reduce over their 4 key rows, waves over LDS
// Row b takes its position from pos_dev[b * pos_stride]; rows with active_dev == 0 are skipped (the decode pool).
// =============================================================================================
constexpr int DEC_MAX_CTX = 4096;

// KV32 (precision mode "split"): the projection output and the K / V cache are fp32 (4 dims per 16-byte load, HD / 4 lanes
// per key row), q is not rounded, and the output row is written as bf16 hi / lo rows of a stacked group layout (out_G)
// KVF == 2: fp24 caches (rows of hd x u16 | hd x u8; 8 elements per lane = one 16-byte + one 8-byte load) — 3 bytes per element
// at 2^-17 relative precision: the split step's attention streams 0.75 of the fp32 bytes
// KVF == 3: e4m3 caches (rows of hd bytes) under the bf16 step — the fp8 weight format's KV: half the bf16 bytes
template <int HD, int UK, int KVF = 0>               // UK = independent row loads in flight per lane; KVF: 0 bf16, 1 fp32, 2 fp24, 3 e4m3
__global__ __launch_bounds__(512) void attention_decode_fused_kernel(AttnDecodeFusedArgs p) {
    constexpr bool KV32 = KVF == 1 || KVF == 2;   // the split step's operand forms (fp32 projection rows, unrounded q, hi / lo output rows)
    constexpr bool F24 = KVF == 2;
    constexpr bool F8 = KVF == 3;         // e4m3 caches under the bf16 step (the fp8 weight format): 16 elements per 16-byte load
    constexpr int EPL = KVF == 1 ? 4 : (F8 ? 16 : 8);   // elements per lane
    constexpr int ESZ = KVF == 1 ? 4 : (F8 ? 1 : 2);    // bytes per element of the (hi) plane
    constexpr int ROWB = F24 ? 3 * HD : HD * ESZ;   // bytes per cache row
    constexpr int LPK = HD / EPL;         // lanes per key row
    constexpr int KPW = 64 / LPK;         // key rows per wave-instruction
    constexpr int BATCH = 8 * KPW * UK;   // keys one round of the 8 waves covers
    __shared__ __attribute__((aligned(16))) float sc[DEC_MAX_CTX + BATCH];  // scores, padded to whole rounds
    __shared__ __attribute__((aligned(16))) float q_s[HD];
    __shared__ __attribute__((aligned(16))) float part[8][HD];
    __shared__ float red[8];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = blockIdx.x, b = blockIdx.y;
    const size_t bh = (size_t)b * p.H + h;
    if (p.active_dev != nullptr && p.active_dev[(size_t)b * p.pos_stride] == 0) return;  // a free row of the decode pool
    stamp_begin(p.stamp, blockIdx.y * gridDim.x + blockIdx.x);
    const int pos = p.pos_dev[(size_t)b * p.pos_stride];
    const int ctx = pos + 1;
    const int D = p.H * HD;
    char* kbase = reinterpret_cast<char*>(p.k) + bh * p.kv_stride * ROWB;
    char* vbase = reinterpret_cast<char*>(p.v) + bh * p.kv_stride * ROWB;
    // Both streams run as a ROLLING window of UK loads per lane (phase 1 / 3 below).  wave-uniform base + one 32-bit byte offset per
    // lane (the scalar-base addressing form: half the address VGPRs); rows past the context are clamped to the last valid row
    constexpr int EPL_ = EPL;
    const int krow = lane / LPK, kcol = (lane % LPK) * EPL_;
    auto row_off = [&](int key) { return (uint32_t)(min(key, ctx - 1) * ROWB) + (uint32_t)(kcol * ESZ); };
    // fp24: the lo plane of the row, 8 bytes per lane
    auto lo_off = [&](int key) { return (uint32_t)(min(key, ctx - 1) * ROWB) + (uint32_t)(2 * HD + kcol); };
    u32x4 kv[UK];
    u32x2 kl[F24 ? UK : 1];
    auto first_k = [&]() {
#pragma unroll
        for (int u = 0; u < UK; ++u) {
            kv[u] = ld16_stream(kbase + row_off(wave * KPW * UK + u * KPW + krow));
            if constexpr (F24) kl[u] = ld8_stream(kbase + lo_off(wave * KPW * UK + u * KPW + krow));
        }
    };
    // The first K batch (keys < BATCH) does not depend on phase 0 unless it contains the new token's own row: it is requested BEHIND
    // phase 0's operand loads and BEFORE its arithmetic and stores (round 6) — loads return in order, so the rotation waits for its
    // operands only, and the barrier's drain of the appended rows overlaps the batch's flight instead of preceding it (one HBM
    // round trip less per launch).  Requested unconditionally (a branch around it makes the compiler count the operands' wait for
    // the path WITHOUT the batch, i.e. wait for half of it); a context shorter than one round requests it again behind the
    // barrier, the only case in which the first request can have read the new row before it was written.
    const bool early = pos >= BATCH;
    // ---- phase 0: rotate q,k of the new token, append k / v to the cache (global) and keep q in LDS
    if (tid < HD / 2) {
        const int d = tid;
        const float c = p.rope_cos[(size_t)pos * (HD / 2) + d], s = p.rope_sin[(size_t)pos * (HD / 2) + d];
        if constexpr (KV32) {
            const float* row = reinterpret_cast<const float*>(p.qkv) + (size_t)b * (3 * D) + h * HD;
            const float q0 = row[d], q1 = row[d + HD / 2];
            const float k0 = row[D + d], k1 = row[D + d + HD / 2];
            const float v0 = row[2 * D + d], v1 = row[2 * D + d + HD / 2];
            first_k();
            q_s[d] = q0 * c - q1 * s;
            q_s[d + HD / 2] = q1 * c + q0 * s;
            if constexpr (F24) {
                auto put = [&](char* rowp, int dd, float val) {
                    const uint32_t code = f32_to_f24(val);
                    reinterpret_cast<uint16_t*>(rowp)[dd] = (uint16_t)(code >> 8);
                    reinterpret_cast<uint8_t*>(rowp + 2 * HD)[dd] = (uint8_t)(code & 0xFFu);
                };
                char* ko = kbase + (size_t)pos * ROWB;
                put(ko, d, k0 * c - k1 * s);
                put(ko, d + HD / 2, k1 * c + k0 * s);
                char* vo = vbase + (size_t)pos * ROWB;
                put(vo, d, v0);
                put(vo, d + HD / 2, v1);
            } else {
                float* ko = reinterpret_cast<float*>(kbase) + (size_t)pos * HD;
                ko[d] = k0 * c - k1 * s;
                ko[d + HD / 2] = k1 * c + k0 * s;
                float* vo = reinterpret_cast<float*>(vbase) + (size_t)pos * HD;
                vo[d] = v0;
                vo[d + HD / 2] = v1;
            }
        } else {
            const bf16_t* row = p.qkv + (size_t)b * (3 * D) + h * HD;
            const bf16_t rq0 = row[d], rq1 = row[d + HD / 2], rk0 = row[D + d], rk1 = row[D + d + HD / 2];
            const bf16_t v0 = row[2 * D + d], v1 = row[2 * D + d + HD / 2];
            first_k();
            const float q0 = bf2f(rq0), q1 = bf2f(rq1);
            const float k0 = bf2f(rk0), k1 = bf2f(rk1);
            q_s[d] = bf2f(f2bf(q0 * c - q1 * s));            // q is rounded to bf16 exactly like the unfused path
            q_s[d + HD / 2] = bf2f(f2bf(q1 * c + q0 * s));
            if constexpr (F8) {   // e4m3 rows: the bf16 values the bf16 cache would hold, re-rounded (as the prefill's writer does)
                uint8_t* ko = reinterpret_cast<uint8_t*>(kbase) + (size_t)pos * HD;
                ko[d] = f2fp8(bf2f(f2bf(k0 * c - k1 * s)));
                ko[d + HD / 2] = f2fp8(bf2f(f2bf(k1 * c + k0 * s)));
                uint8_t* vo = reinterpret_cast<uint8_t*>(vbase) + (size_t)pos * HD;
                vo[d] = f2fp8(bf2f(v0));
                vo[d + HD / 2] = f2fp8(bf2f(v1));
            } else {
                bf16_t* ko = reinterpret_cast<bf16_t*>(kbase) + (size_t)pos * HD;
                ko[d] = f2bf(k0 * c - k1 * s);
                ko[d + HD / 2] = f2bf(k1 * c + k0 * s);
                bf16_t* vo = reinterpret_cast<bf16_t*>(vbase) + (size_t)pos * HD;
                vo[d] = v0;
                vo[d + HD / 2] = v1;
            }
        }
    } else {
        first_k();
    }
    __syncthreads();  // workgroup-scope release/acquire: the appended K / V rows are visible to this block
    // ---- phase 1: scores
    float qv[EPL];
#pragma unroll
    for (int e = 0; e < EPL; ++e) qv[e] = q_s[(lane % LPK) * EPL + e];
    const int ctx_pad = (ctx + BATCH - 1) / BATCH * BATCH;
    // keys hidden by the attention_mask of the row's prefill (vc_decode_step loops that keep a caller's mask; nullptr: none).
    // The new token's own key (position pos) is always visible.
    const uint8_t* kmask = p.key_mask != nullptr ? p.key_mask + (size_t)b * p.mask_stride : nullptr;
    // (a register of the rolling window is re-requested for the next batch as soon as its row has been consumed: UK rows in flight
    // with UK registers, no second set; no branch around the loads)
    auto dot = [&](const u32x4& r, const u32x2& rl) {
        float s = 0.f;
        if constexpr (F24) {
#pragma unroll
            for (int e = 0; e < 8; ++e) s += qv[e] * f24_elem(r, rl, e);
        } else if constexpr (F8) {
#pragma unroll
            for (int w4 = 0; w4 < 4; ++w4) {
                const f32x4 f = fp8x4_to_f32x4(r[w4]);
#pragma unroll
                for (int e = 0; e < 4; ++e) s += qv[4 * w4 + e] * f[e];
            }
        } else if constexpr (KV32) {
            const f32x4 f = __builtin_bit_cast(f32x4, r);
#pragma unroll
            for (int e = 0; e < 4; ++e) s += qv[e] * f[e];
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) s += qv[2 * e] * bf2f_lo(r[e]) + qv[2 * e + 1] * bf2f_hi(r[e]);
        }
        return s;
    };
    if (!early) first_k();
    for (int kb = wave * KPW * UK; kb < ctx_pad; kb += BATCH) {
#pragma unroll
        for (int u = 0; u < UK; ++u) {
            const int key = kb + u * KPW + krow;
            float s = dot(kv[u], kl[F24 ? u : 0]);
            kv[u] = ld16_stream(kbase + row_off(key + BATCH));
            if constexpr (F24) kl[u] = ld8_stream(kbase + lo_off(key + BATCH));
            s = lanes_sum<LPK>(s);
            if ((lane % LPK) == 0) sc[key] = (key < ctx && (kmask == nullptr || kmask[min(key, ctx - 1)] != 0)) ? s * p.scale : -INFINITY;
        }
    }
    // the first V batch of every wave does not depend on the scores: request it now so HBM stays busy through the
    // LDS-only softmax below
    u32x4 vv[UK];
    u32x2 vl[F24 ? UK : 1];
#pragma unroll
    for (int u = 0; u < UK; ++u) {
        vv[u] = ld16_stream(vbase + row_off(wave * KPW * UK + u * KPW + krow));
        if constexpr (F24) vl[u] = ld8_stream(vbase + lo_off(wave * KPW * UK + u * KPW + krow));
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
    // ---- phase 3: out[d] = sum_key p[key] * V[key][d]; a lane accumulates its dims over the key rows it loads (keys
    // past the context carry p = 0 and re-read the last valid row)
    float acc[EPL];
#pragma unroll
    for (int e = 0; e < EPL; ++e) acc[e] = 0.f;
    for (int kb = wave * KPW * UK; kb < ctx_pad; kb += BATCH) {
#pragma unroll
        for (int u = 0; u < UK; ++u) {
            const int key = kb + u * KPW + krow;
            const float pk = sc[key];
            const u32x4 v = vv[u];
            const u32x2 vlo = vl[F24 ? u : 0];
            vv[u] = ld16_stream(vbase + row_off(key + BATCH));
            if constexpr (F24) {
                vl[u] = ld8_stream(vbase + lo_off(key + BATCH));
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] += pk * f24_elem(v, vlo, e);
            } else if constexpr (F8) {
#pragma unroll
                for (int w4 = 0; w4 < 4; ++w4) {
                    const f32x4 f = fp8x4_to_f32x4(v[w4]);
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[4 * w4 + e] += pk * f[e];
                }
            } else if constexpr (KV32) {
                const f32x4 f = __builtin_bit_cast(f32x4, v);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[e] += pk * f[e];
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    acc[2 * e] += pk * bf2f_lo(v[e]);
                    acc[2 * e + 1] += pk * bf2f_hi(v[e]);
                }
            }
        }
    }
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
        float a = acc[e];
#pragma unroll
        for (int mk = LPK; mk < 64; mk <<= 1) a += shfl_xor(a, mk);  // over the KPW key rows of the wave-instruction
        acc[e] = a;
    }
    if (lane < LPK) {
#pragma unroll
        for (int e = 0; e < EPL; ++e) part[wave][lane * EPL + e] = acc[e];
    }
    __syncthreads();
    if (tid < HD) {
        float a = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) a += part[w][tid];
        a *= inv;
        if constexpr (KV32) {  // stacked hi / lo row groups of the split decode GEMV: row b -> group b / G, slot b % G
            const int G = p.out_G;
            const size_t orow = (size_t)(b / G) * 2 * G + b % G;
            const bf16_t hi = f2bf(a);
            p.out[orow * D + h * HD + tid] = hi;
            p.out[(orow + G) * D + h * HD + tid] = f2bf(a - bf2f(hi));
        } else {
            p.out[(size_t)b * D + h * HD + tid] = f2bf(a);
        }
    }
    stamp_end(p.stamp, blockIdx.y * gridDim.x + blockIdx.x);
}

void launch_attention_decode_fused(const AttnDecodeFusedArgs& a, hipStream_t s) {
    const dim3 grid(a.H, a.B), block(512);
    // 8 row loads in flight per lane (8 KiB per wave, two workgroups per CU): 12 and 16 measured slower at every row count
    // (profiles/r05_u_kbench_dattn_uk.txt: 32 rows 99.4 / 103.4 / 108.7 us) and their instantiations were removed in round 5
    if (a.kv32 == 3) {  // bf16 step over e4m3 caches (the fp8 weight format)
        // rows in flight per lane: 16 elements per load cost 16 + 16 query / accumulator registers; 8 rows took 156 VGPRs (one
        // workgroup per CU); 6 rows: 124 VGPRs, two workgroups per CU (profiles/r04_n_kbench_dattn_kv8_uk.txt)
        if (a.hd == 128) {
            VC_LAUNCH((attention_decode_fused_kernel<128, 6, 3>), grid, block, 0, s, a);
        } else {
            VC_LAUNCH((attention_decode_fused_kernel<64, 4, 3>), grid, block, 0, s, a);
        }
        return;
    }
    if (a.kv32 == 2) {  // precision mode "split", fp24 caches
        // 6 rows x 24 B in flight per lane (the bf16 kernel's 8 x 16 B take 114 VGPRs, 8 rows of fp24 171: one workgroup per CU)
        if (a.hd == 128) VC_LAUNCH((attention_decode_fused_kernel<128, 6, 2>), grid, block, 0, s, a);
        else VC_LAUNCH((attention_decode_fused_kernel<64, 6, 2>), grid, block, 0, s, a);
        return;
    }
    if (a.kv32) {  // precision mode "split", fp32 caches
        if (a.hd == 128) VC_LAUNCH((attention_decode_fused_kernel<128, 8, 1>), grid, block, 0, s, a);
        else VC_LAUNCH((attention_decode_fused_kernel<64, 8, 1>), grid, block, 0, s, a);
        return;
    }
    if (a.hd == 128) {
        VC_LAUNCH((attention_decode_fused_kernel<128, 8>), grid, block, 0, s, a);
    } else {
        VC_LAUNCH((attention_decode_fused_kernel<64, 8>), grid, block, 0, s, a);
    }
}

}  // namespace vc
