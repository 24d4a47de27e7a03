// decode.hip — weight-streaming skinny GEMM for the decode steps (M = batch <= 16 tokens).
//
//   out[m][n] = epilogue( sum_k X[m][k] * W[n][k] )        (Llama linears have no bias)
//
// Replaces nn.Linear at q/k/v/o_proj, gate/up/down_proj and lm_head of
// [HF] llama/modeling_llama.py:174-176,254-256,280,413 + vcoder_ds_llava_llama.py:93 during the
// generate() loop.  Each decode step streams every decoder weight
// exactly once, so this kernel is bound by HBM, not MFMA: the weights are PRE-PACKED at load time in
// MFMA-fragment order ([N/16][K/32][64 lanes][8 bf16]) so that one wave-instruction reads one fully
// contiguous 1 KiB block with non-temporal loads, and a v_mfma_f32_16x16x32_bf16 per block does the
// 16 outputs x 16 token-slots x 32 k dot products (tokens >= M are fed zeros).  K is split across the
// waves of a workgroup and reduced through LDS; epilogues (fp32 residual add, SwiGLU) are fused.
#include <stdlib.h>

#include "vc_device.h"
#include "kernels.h"

namespace vc {

#ifdef VC_EMU
VC_DEV u32x4 ld16_stream(const void* p) { return ld16(p); }
#else
VC_DEV u32x4 ld16_stream(const void* p) { return __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p)); }
#endif

// normalised bf16 fragment of 8 activations: bf16( (x * rstd) * w )   ([HF] llama/modeling_llama.py:62-67)
VC_DEV u32x4 norm_frag(const float* xp, const float* wp, float rstd) {
    const f32x4 x0 = ld16f(xp), x1 = ld16f(xp + 4), w0 = ld16f(wp), w1 = ld16f(wp + 4);
    return u32x4{pack_bf2((x0[0] * rstd) * w0[0], (x0[1] * rstd) * w0[1]), pack_bf2((x0[2] * rstd) * w0[2], (x0[3] * rstd) * w0[3]),
                 pack_bf2((x1[0] * rstd) * w1[0], (x1[1] * rstd) * w1[1]), pack_bf2((x1[2] * rstd) * w1[2], (x1[3] * rstd) * w1[3])};
}

// WAVES waves split K; each workgroup owns NT consecutive 16-output tiles so that one (normalised) activation
// fragment feeds NT weight tiles — this halves the L2 traffic of the activation operand for NT = 2.
//
// NORM (fused RMSNorm) comes in two forms.  STAGE: the workgroup first normalises all M rows ONCE into LDS
// (bf16, rows padded by 16 B so the 16 token rows of a fragment read hit 16 different bank groups) and the K loop
// then reads activation fragments with ds_read_b128 — the streaming loop is as lean as the plain one.  !STAGE
// (rows do not fit the 160 KiB LDS): every wave normalises its own fragments straight from L2.
template <int WAVES, int NT, int EPI, bool NORM, bool STAGE>
__global__ __launch_bounds__(WAVES * 64) void gemv_kernel(GemvArgs p) {
    VC_DYNAMIC_SMEM(char, dsm);  // STAGE: normalised activations, later re-used for the cross-wave reduction
    __shared__ __attribute__((aligned(16))) float red_static[STAGE ? 1 : WAVES * NT * 64 * 4];
    __shared__ float rstd_s[16];
    float* red = STAGE ? reinterpret_cast<float*>(dsm) : red_static;  // [WAVES][NT][64][4]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ntiles = p.N >> 4;
    const int nt0 = blockIdx.x * NT;
    const int nkt = p.K >> 5;
    const int m = lane & 15, g = lane >> 4;
    const bool mvalid = m < p.M;
    const bf16_t* wp[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) wp[t] = p.Wp + ((size_t)min(nt0 + t, ntiles - 1) * nkt * 64 + lane) * 8;
    const bf16_t* xp = p.X + (size_t)(mvalid ? m : 0) * p.K + g * 8;
    const float* xf = nullptr;
    const float* nw = nullptr;
    float rstd = 0.f;
    // STAGE: K is walked in chunks of p.kc columns so that the normalised rows of a chunk (Mp x kc bf16) fit in LDS
    const int kc_tiles = STAGE ? (p.kc >> 5) : nkt;
    const int row_bytes = (STAGE ? p.kc : p.K) * 2 + 16;
    const int Mp = p.M <= 8 ? 8 : 16;
    if constexpr (NORM && STAGE) {
        for (int r = wave; r < Mp; r += WAVES) {  // 1/rms per row from the producer's partials (fixed order)
            float ss = 0.f;
            for (int q = lane; q < p.npart; q += 64) ss += p.ssq_in[(size_t)r * p.npart + q];
            ss = wave_sum(ss);
            if (lane == 0) rstd_s[r] = rsqrtf(ss / (float)p.K + p.eps);
        }
    }
    if constexpr (NORM && !STAGE) {
        // per-row 1/rms from the producer's deterministic partial sums (fixed summation order -> bit-reproducible)
        const float* sp = p.ssq_in + (size_t)m * p.npart;
        float ss = 0.f;
        for (int q = g; q < (p.npart >> 2); q += 4) {
            const f32x4 v = ld16f(sp + q * 4);
            ss += (v[0] + v[1]) + (v[2] + v[3]);
        }
        ss += shfl_xor(ss, 16);
        ss += shfl_xor(ss, 32);
        rstd = rsqrtf(ss / (float)p.K + p.eps);
        xf = p.Xf + (size_t)(mvalid ? m : 0) * p.K + g * 8;
        nw = p.norm_w + g * 8;
    }
    f32x4 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    // K loop, software-pipelined over batches of U k-tiles with two register sets: the weight loads of batch b+1 are in
    // flight while the MFMAs of batch b run, and the first batch of a chunk is requested BEFORE its normalise prologue.
    constexpr int U = 8 / NT;
    u32x4 wa[NT][U], wb[NT][U];
    auto load_w = [&](u32x4 (&w)[NT][U], int kt) {
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int t = 0; t < NT; ++t) w[t][u] = ld16_stream(wp[t] + (size_t)(kt + u) * 512);
    };
    for (int cbase = 0; cbase < nkt; cbase += kc_tiles) {
        const int per = (kc_tiles + WAVES - 1) / WAVES;
        const int kt0 = cbase + wave * per, kt1 = min(cbase + kc_tiles, kt0 + per);
        const int nb = max(kt1 - kt0, 0) / U;  // full batches of this wave in this chunk (wave-uniform)
        if (nb > 0) load_w(wa, kt0);
        if constexpr (NORM && STAGE) {
            __syncthreads();  // rstd_s ready / every wave finished reading the previous chunk's rows
            // normalise: thread t owns 16-byte column groups t, t+512, ... of EVERY row; the row loop is unrolled so the
            // loads of all rows are in flight together (an un-unrolled loop pays one L2 round trip per row)
            const int cpr = p.kc >> 3, col0 = cbase * 32;
            for (int c = tid; c < cpr; c += WAVES * 64) {
                const f32x4 w0 = ld16f(p.norm_w + col0 + c * 8), w1 = ld16f(p.norm_w + col0 + c * 8 + 4);
#pragma unroll 8
                for (int r = 0; r < Mp; ++r) {
                    u32x4 v = {0u, 0u, 0u, 0u};
                    if (r < p.M) {
                        const float* xr = p.Xf + (size_t)r * p.K + col0 + c * 8;
                        const f32x4 x0 = ld16f(xr), x1 = ld16f(xr + 4);
                        const float rs = rstd_s[r];
                        v = u32x4{pack_bf2((x0[0] * rs) * w0[0], (x0[1] * rs) * w0[1]), pack_bf2((x0[2] * rs) * w0[2], (x0[3] * rs) * w0[3]),
                                  pack_bf2((x1[0] * rs) * w1[0], (x1[1] * rs) * w1[1]), pack_bf2((x1[2] * rs) * w1[2], (x1[3] * rs) * w1[3])};
                    }
                    st16(dsm + (size_t)r * row_bytes + c * 16, v);
                }
            }
            __syncthreads();
        }
        // staged fragment of k-tile kt: row m, chunk-relative column (kt - cbase)*32 + g*8
        const char* xs = dsm + (size_t)m * row_bytes + g * 16 - (size_t)cbase * 64;
        auto compute = [&](u32x4 (&w)[NT][U], int kt) {
            u32x4 xv[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if constexpr (NORM && STAGE) xv[u] = mvalid ? ld16(xs + (kt + u) * 64) : u32x4{0u, 0u, 0u, 0u};
                else if constexpr (NORM) xv[u] = mvalid ? norm_frag(xf + (kt + u) * 32, nw + (kt + u) * 32, rstd) : u32x4{0u, 0u, 0u, 0u};
                else xv[u] = mvalid ? ld16(xp + (kt + u) * 32) : u32x4{0u, 0u, 0u, 0u};
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[t] = mfma16(w[t][u], xv[u], acc[t]);
        };
        for (int b = 0; b < nb; b += 2) {
            if (b + 1 < nb) load_w(wb, kt0 + (b + 1) * U);
            compute(wa, kt0 + b * U);
            if (b + 1 < nb) {
                if (b + 2 < nb) load_w(wa, kt0 + (b + 2) * U);
                compute(wb, kt0 + (b + 1) * U);
            }
        }
        for (int kt = kt0 + nb * U; kt < kt1; ++kt) {
            u32x4 xv = {0u, 0u, 0u, 0u};
            if (mvalid) {
                if constexpr (NORM && STAGE) xv = ld16(xs + kt * 64);
                else if constexpr (NORM) xv = norm_frag(xf + kt * 32, nw + kt * 32, rstd);
                else xv = ld16(xp + kt * 32);
            }
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = mfma16(ld16_stream(wp[t] + (size_t)kt * 512), xv, acc[t]);
        }
    }
    if constexpr (STAGE) __syncthreads();  // every wave is done with the staged activations before `red` overwrites them
#pragma unroll
    for (int t = 0; t < NT; ++t) st16f(red + ((wave * NT + t) * 64 + lane) * 4, acc[t]);
    __syncthreads();
    if (wave >= NT) return;
    const int nt = nt0 + wave;  // wave t finishes tile t
    if (nt >= ntiles) return;
    f32x4 v = ld16f(red + ((0 * NT + wave) * 64 + lane) * 4);
#pragma unroll
    for (int w = 1; w < WAVES; ++w) v = v + ld16f(red + ((w * NT + wave) * 64 + lane) * 4);
    const int n = nt * 16 + g * 4;  // lane holds out[m][n..n+3]
    if constexpr (EPI == GEMV_RESID_F32) {
        float* o = reinterpret_cast<float*>(p.out) + (size_t)(mvalid ? m : 0) * p.ldo + n;
        if (mvalid) {
            v = ld16f(o) + v;
            st16f(o, v);
        }
        if (p.ssq_out) {  // sum of squares of this tile's 16 new residual values of token m (fixed order)
            float sq = (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
            sq += shfl_xor(sq, 16);
            sq += shfl_xor(sq, 32);
            if (g == 0 && mvalid) p.ssq_out[(size_t)m * p.npart + nt] = sq;
        }
        return;
    }
    if (!mvalid) return;
    if constexpr (EPI == GEMV_BF16) {
        st8(reinterpret_cast<bf16_t*>(p.out) + (size_t)m * p.ldo + n, u32x2{pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3])});
    } else if constexpr (EPI == GEMV_F32) {
        st16f(reinterpret_cast<float*>(p.out) + (size_t)m * p.ldo + n, v);
    } else if constexpr (EPI == GEMV_SWIGLU) {
        *reinterpret_cast<uint32_t*>(reinterpret_cast<bf16_t*>(p.out) + (size_t)m * p.ldo + (n >> 1)) =
            pack_bf2(silu(v[0]) * v[1], silu(v[2]) * v[3]);
    }
}

template <class K>
static void allow_big_lds(K kernel, size_t bytes) {
#ifndef VC_EMU
    if (bytes > 48 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
#endif
}

template <int WAVES, int NT, bool NORM, bool STAGE>
static void launch_gemv_w(const GemvArgs& a, int epi, size_t shmem, hipStream_t s) {
    const dim3 grid((a.N / 16 + NT - 1) / NT), block(WAVES * 64);
    switch (epi) {
        case GEMV_BF16:
            allow_big_lds(gemv_kernel<WAVES, NT, GEMV_BF16, NORM, STAGE>, shmem);
            VC_LAUNCH((gemv_kernel<WAVES, NT, GEMV_BF16, NORM, STAGE>), grid, block, shmem, s, a);
            break;
        case GEMV_F32:
            allow_big_lds(gemv_kernel<WAVES, NT, GEMV_F32, NORM, STAGE>, shmem);
            VC_LAUNCH((gemv_kernel<WAVES, NT, GEMV_F32, NORM, STAGE>), grid, block, shmem, s, a);
            break;
        case GEMV_RESID_F32:
            allow_big_lds(gemv_kernel<WAVES, NT, GEMV_RESID_F32, NORM, STAGE>, shmem);
            VC_LAUNCH((gemv_kernel<WAVES, NT, GEMV_RESID_F32, NORM, STAGE>), grid, block, shmem, s, a);
            break;
        default:
            allow_big_lds(gemv_kernel<WAVES, NT, GEMV_SWIGLU, NORM, STAGE>, shmem);
            VC_LAUNCH((gemv_kernel<WAVES, NT, GEMV_SWIGLU, NORM, STAGE>), grid, block, shmem, s, a);
            break;
    }
}

void launch_gemv(const GemvArgs& a, int epilogue, hipStream_t s) {
    if (a.Xf != nullptr) {  // fused RMSNorm prologue; 2 tiles per workgroup share every activation fragment
        // chunk K so that one chunk of normalised rows (Mp x kc bf16, 16-byte row pad) stays <= 70 KiB -> 2 workgroups
        // per CU; kc must keep every wave's k-tile share whole (multiple of 8 waves x 32)
        const size_t Mp = a.M <= 8 ? 8 : 16;
        int nch = 0;
        for (int c = 1; c <= 16 && !nch; ++c)
            if (a.K % c == 0 && (a.K / c) % 256 == 0 && Mp * ((size_t)(a.K / c) * 2 + 16) <= 70 * 1024) nch = c;
        static const int stage_ok = getenv("VC_GEMV_STAGE") ? atoi(getenv("VC_GEMV_STAGE")) : 1;
        if (stage_ok && nch) {
            GemvArgs b = a;
            b.kc = a.K / nch;
            const size_t stage = Mp * ((size_t)b.kc * 2 + 16);
            const size_t need = stage > (size_t)8 * 2 * 64 * 16 ? stage : (size_t)8 * 2 * 64 * 16;  // also holds `red`
            launch_gemv_w<8, 2, true, true>(b, epilogue, need, s);
        } else {
            launch_gemv_w<8, 2, true, false>(a, epilogue, 0, s);
        }
        return;
    }
    // plain activations: >= ~2048 waves in flight — few output tiles -> more K-splitting waves per workgroup
    if (a.N / 16 <= 512) launch_gemv_w<8, 1, false, false>(a, epilogue, 0, s);
    else launch_gemv_w<4, 1, false, false>(a, epilogue, 0, s);
}

// W [N,K] row-major -> packed fragment order (done once at weight-load time)
__global__ __launch_bounds__(256) void pack_weight_kernel(const bf16_t* W, bf16_t* Wp, int N, int K) {
    const size_t id = (size_t)blockIdx.x * 256 + threadIdx.x;  // one 16-B chunk each
    const size_t total = (size_t)N * K / 8;
    if (id >= total) return;
    const int lane = (int)(id & 63);
    const size_t tile = id >> 6;
    const int nkt = K >> 5;
    const size_t nt = tile / nkt;
    const int kt = (int)(tile % nkt);
    const u32x4 v = ld16(W + (nt * 16 + (lane & 15)) * (size_t)K + kt * 32 + (lane >> 4) * 8);
    st16(Wp + id * 8, v);
}
void launch_pack_weight(const bf16_t* W, bf16_t* Wp, int N, int K, hipStream_t s) {
    const size_t total = (size_t)N * K / 8;
    VC_LAUNCH(pack_weight_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, W, Wp, N, K);
}

__global__ __launch_bounds__(256) void interleave_rows_kernel(const bf16_t* gate, const bf16_t* up, bf16_t* out, int F,
                                                              int K) {
    const size_t id = (size_t)blockIdx.x * 256 + threadIdx.x;
    const int cpr = K / 8;
    const size_t total = (size_t)2 * F * cpr;
    if (id >= total) return;
    const size_t row = id / cpr;
    const int c = (int)(id % cpr);
    const bf16_t* src = (row & 1) ? up : gate;
    st16(out + row * K + c * 8, ld16(src + (row >> 1) * K + c * 8));
}
void launch_interleave_rows(const bf16_t* gate, const bf16_t* up, bf16_t* out, int F, int K, hipStream_t s) {
    const size_t total = (size_t)2 * F * (K / 8);
    VC_LAUNCH(interleave_rows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, gate, up, out, F, K);
}

}  // namespace vc
