// gemm.hip — bf16 MFMA GEMM for gfx950 with fused epilogues.
//
//   out[m][n] = epilogue( sum_k A[m][k] * W[n][k] + bias[n] )      A:[M,K] bf16, W:[N,K] bf16 (HF Linear layout)
//
// Replaces the torch nn.Linear / Conv2d-as-GEMM calls the reference makes through HF
// ([HF] clip/modeling_clip.py:309-311,333,346-350,
//  [HF] llama/modeling_llama.py:174-176,254-256,280; vcoder_llava/model/multimodal_projector/builder.py:42-46).
//
// Design (CDNA4): 128x128 output tile per 256-thread workgroup (4 waves, 2x2, 64x64 each = 4x4 MFMA
// 16x16x32 fragments, 64 fp32 accumulators per lane), BK=64, two LDS stages (64 KiB -> 2 workgroups/CU),
// XOR-swizzled LDS rows so the ds_read_b128 fragment reads are bank-conflict free.  Two staging forms:
// LDS-DMA (global_load_lds_dwordx4, default) and register staging (global_load_dwordx4 issued before the
// MFMAs of the current tile, ds_write_b128 after).
// The WEIGHT tile is the MFMA A operand and the ACTIVATION tile the B operand, so each lane ends up with
// 4 consecutive output features of one token: epilogues (bias, GELU, residual add, SwiGLU) are lane-local
// and stores are 8/16 bytes per lane.
#include <stdlib.h>

#include "vc_device.h"
#include <algorithm>
#include <stdexcept>

#include "kernels.h"

namespace vc {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = 128 * BK * 2;  // one operand tile: 128 rows x 128 B

// byte offset of 16-byte chunk `c` (0..7) of row `r` inside a swizzled [128][64] bf16 tile
VC_DEV int swz(int r, int c) { return r * 128 + ((c ^ (r & 7)) << 4); }

// map linear workgroup id -> (tile_m, tile_n): XCD-contiguous remap (block b runs on XCD b%8; give each
// XCD a contiguous range of tiles so neighbours share operand panels in that XCD's L2), then groups of
// 8 m-tiles sweep n so a group re-uses its activation panel while streaming weights.
VC_DEV int xcd_remap(int bid, int nblk) {
    const int q = nblk / 8, r = nblk % 8, x = bid % 8, i = bid / 8;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
}
VC_DEV void tile_from_pid(int pid, int tiles_m, int tiles_n, int& tm, int& tn, int GROUP = 8) {
    const int per_group = GROUP * tiles_n;
    const int g = pid / per_group;
    const int first_m = g * GROUP;
    const int gsz = min(tiles_m - first_m, GROUP);
    tm = first_m + (pid % per_group) % gsz;
    tn = (pid % per_group) / gsz;
}
VC_DEV void tile_coords(int bid, int nblk, int tiles_m, int tiles_n, int& tm, int& tn) {
    tile_from_pid(xcd_remap(bid, nblk), tiles_m, tiles_n, tm, tn);
}

// epilogue store of out[m][n..n+3] (shared by all kernels).  p.split_out != 0 (precision mode "split"):
// a bf16 output becomes TWO bf16 planes, hi = bf16(v) at column n and lo = bf16(v - hi) at column split_out + n of the same
// row — the K-concatenated [hi | lo] operand of the next GEMM, which contracts it against the weight matrix twice
// (GemmArgs::kwrap), i.e. with ~16 mantissa bits of the activation instead of 8.
VC_DEV u32x2 pack_bf4(f32x4 v) { return u32x2{pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3])}; }
VC_DEV f32x4 bf4_residual(f32x4 v, u32x2 hi) {
    return f32x4{v[0] - bf2f_lo(hi[0]), v[1] - bf2f_hi(hi[0]), v[2] - bf2f_lo(hi[1]), v[3] - bf2f_hi(hi[1])};
}
// FIX: the caller is the split-K fix-up kernel (a wave = one output row, 4 adjacent lanes = one 16-column tile) instead of an
// MFMA accumulator layout (lanes l, l ^ 16, l ^ 32, l ^ 48 = the 16 columns of one token)
// (the consumer side of the folded RMSNorm — GemmArgs::row_scale — is applied by the callers, which hold a row's scale in a
// register across the columns they store)
// `resid` (EPI_RESID_F32): the residual values out[m][n..n+3] when the caller requested them ahead of time (the 8-phase kernel's
// epilogue: a load inside the caller's bounds guard is waited for with vmcnt(0) at the end of the guard, one round trip per store)
template <int EPI, bool FIX = false>
VC_DEV void store_out(const GemmArgs& p, int m, int n, f32x4 v, const f32x4* resid = nullptr, const f32x4* xgw = nullptr) {
    if constexpr (EPI == EPI_BF16 || EPI == EPI_BF16_QGELU || EPI == EPI_BF16_GELU) {
        if constexpr (EPI == EPI_BF16_QGELU) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = quick_gelu(v[e]);
        }
        if constexpr (EPI == EPI_BF16_GELU) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = erf_gelu(v[e]);
        }
        const u32x2 o = pack_bf4(v);
        bf16_t* dst = reinterpret_cast<bf16_t*>(p.out) + (size_t)m * p.ldo + n;
        st8(dst, o);
        if (p.split_out) st8(dst + p.split_out, pack_bf4(bf4_residual(v, o)));
    } else if constexpr (EPI == EPI_F32) {
        st16f(reinterpret_cast<float*>(p.out) + (size_t)m * p.ldo + n, v);
    } else if constexpr (EPI == EPI_RESID_F32) {
        float* o = reinterpret_cast<float*>(p.out) + (size_t)m * p.ldo + n;
        v = (resid != nullptr ? *resid : ld16f(o)) + v;
        st16f(o, v);
        if (p.xg_out) {   // folded RMSNorm, producer side: the next GEMM's operand and the row's sum-of-squares partial
            const f32x4 t = v * (xgw != nullptr ? *xgw : ld16f(p.xg_w + n));
            const u32x2 hi = pack_bf4(t);
            bf16_t* d = p.xg_out + (size_t)m * p.ld_xg + n;
            st8(d, hi);
            if (p.xg_lo) st8(d + p.xg_lo, pack_bf4(bf4_residual(t, hi)));
            float sq = (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
            // the 16 columns of the tile sit in 4 lanes that share this lane's m (so they are all here: N % 16 == 0)
            if constexpr (FIX) {
                sq += shfl_xor(sq, 1);
                sq += shfl_xor(sq, 2);
            } else {
                sq = rows_sum(sq);
            }
            if (FIX ? (lane_id() & 3) == 0 : lane_id() < 16) p.ssq_out[(size_t)m * p.npart + (n >> 4)] = sq;
        }
    } else {  // EPI_SWIGLU: (g0,u0,g1,u1) -> 2 outputs
        const float h0 = silu(v[0]) * v[1], h1 = silu(v[2]) * v[3];
        const uint32_t o = pack_bf2(h0, h1);
        bf16_t* dst = reinterpret_cast<bf16_t*>(p.out) + (size_t)m * p.ldo + (n >> 1);
        *reinterpret_cast<uint32_t*>(dst) = o;
        if (p.split_out) *reinterpret_cast<uint32_t*>(dst + p.split_out) = pack_bf2(h0 - bf2f_lo(o), h1 - bf2f_hi(o));
    }
}

// byte offset of the WEIGHT operand's k-tile (128 bytes = 64 bf16) for k-tile `kt` of the contraction: with GemmArgs::kwrap =
// K_w / BK the weight matrix is contracted against both halves of a K-concatenated [hi | lo] activation row (K = 2 K_w), i.e. its
// k index wraps; with GemmArgs::w_lo_off a third segment (k-tiles [2 kwrap, 3 kwrap)) takes the weight's lo plane against the
// activation's hi plane (a_koff: the activation's k index wraps back to 0 there)
VC_DEV long long w_koff(int kt, const GemmArgs& p) {
    if (p.kwrap <= 0 || kt < p.kwrap) return (long long)kt * 128;
    if (kt < 2 * p.kwrap) return (long long)(kt - p.kwrap) * 128;
    return p.w_lo_off + (long long)(kt - 2 * p.kwrap) * 128;
}
VC_DEV long long a_koff(int kt, const GemmArgs& p) {
    return (long long)((p.kwrap > 0 && kt >= 2 * p.kwrap) ? kt - 2 * p.kwrap : kt) * 128;
}


// ---- EPI_QKV (kernels.h QkvEpiArgs): RoPE + head split + KV write in the QKV GEMM's epilogue ---------------------------------
// weight row behind row `row` (0..127) of X half `h` of the 256-row tile at n0: {head 2a: d 0..63 | head 2a+1: d 0..63} in half 0,
// the same heads' d 64..127 in half 1 — wave group g = row / 64 then owns ONE head in both halves
VC_DEV int qkv_wrow(int n0, int h, int row) { return n0 + (row >> 6) * 128 + h * 64 + (row & 63); }
// A row behind padded token row mp = b * Tp + t (clamped into the sample; the stores of such rows are masked)
VC_DEV int qkv_arow(const QkvEpiArgs& e, int mp) {
    const int b = min(mp / e.Tp, e.B - 1);
    return b * e.T + min(mp - b * e.Tp, e.T - 1);
}
// 4 rotate-half pairs (x = d0 + i, y = d0 + 64 + i) of token (b, t), head `head`: the projection is rounded to bf16 first (the value
// the unfused path stores in its fused rows), RoPE in fp32 on the rounded values (rope_pair, the split kernel's own function), rounded
// again; which = 0: Q rows, 1: K rows (+ the e4m3 cache row)
// cs / sn: rope_cos / rope_sin [t][dl .. dl + 3] (the GEMM's epilogue requests them ahead of its bounds guards)
VC_DEV void qkv_rope_store(const QkvEpiArgs& e, int which, int head, int b, int t, int dl, f32x4 x, f32x4 y, f32x4 cs, f32x4 sn) {
    const u32x2 xb = pack_bf4(x), yb = pack_bf4(y);
    u32x2 olo, ohi;
#pragma unroll
    for (int h2 = 0; h2 < 2; ++h2) {
        const float x0 = bf2f_lo(xb[h2]), x1 = bf2f_hi(xb[h2]), y0 = bf2f_lo(yb[h2]), y1 = bf2f_hi(yb[h2]);
        const float c0 = cs[2 * h2], c1 = cs[2 * h2 + 1], s0 = sn[2 * h2], s1 = sn[2 * h2 + 1];
        float a0, b0, a1, b1;
        rope_pair(x0, y0, c0, s0, a0, b0);
        rope_pair(x1, y1, c1, s1, a1, b1);
        olo[h2] = pack_bf2(a0, a1);
        ohi[h2] = pack_bf2(b0, b1);
    }
    const size_t bh = (size_t)b * e.H + head;
    bf16_t* dst = which == 0 ? e.q + (bh * e.q_stride + t) * 128 : e.k + (bh * e.kv_stride + t) * 128;
    st8(dst + dl, olo);
    st8(dst + 64 + dl, ohi);
    if (which == 1 && e.k8 != nullptr) {
        uint8_t* d8 = e.k8 + (bh * e.kv8_stride + t) * 128;
        *reinterpret_cast<uint32_t*>(d8 + dl) = f32x4_to_fp8x4(bf2f_lo(olo[0]), bf2f_hi(olo[0]), bf2f_lo(olo[1]), bf2f_hi(olo[1]));
        *reinterpret_cast<uint32_t*>(d8 + 64 + dl) = f32x4_to_fp8x4(bf2f_lo(ohi[0]), bf2f_hi(ohi[0]), bf2f_lo(ohi[1]), bf2f_hi(ohi[1]));
    }
}
VC_DEV void qkv_rope_store(const QkvEpiArgs& e, int which, int head, int b, int t, int dl, f32x4 x, f32x4 y) {
    qkv_rope_store(e, which, head, b, t, dl, x, y, ld16f(e.rope_cos + (size_t)t * 64 + dl), ld16f(e.rope_sin + (size_t)t * 64 + dl));
}
// 4 value features d0 .. d0 + 3 of token (b, t) into the cache row(s)
VC_DEV void qkv_v_store(const QkvEpiArgs& e, int head, int b, int t, int d0, u32x2 vb) {
    const size_t bh = (size_t)b * e.H + head;
    if (e.v != nullptr) st8(e.v + (bh * e.kv_stride + t) * 128 + d0, vb);
    if (e.v8 != nullptr)
        *reinterpret_cast<uint32_t*>(e.v8 + (bh * e.kv8_stride + t) * 128 + d0) =
            f32x4_to_fp8x4(bf2f_lo(vb[0]), bf2f_hi(vb[0]), bf2f_lo(vb[1]), bf2f_hi(vb[1]));
}
// column of key kk (0..31) of a 32-key block inside the V^T scratch row (attn.hip vt_chunk_key0: chunk c holds keys 4c..4c+3, 16+4c..16+4c+3)
VC_DEV int vt_pos32(int kk) { return ((kk & 15) >> 2) * 8 + (kk & 3) + 4 * (kk >> 4); }

template <int EPI>
__global__ __launch_bounds__(256) void gemm_bf16_kernel(GemmArgs p) {
    VC_DYNAMIC_SMEM(char, smem);  // [2 stages][W tile | A tile]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wn = wave >> 1, wm = wave & 1;
    const int tiles_m = (p.M + BM - 1) / BM, tiles_n = (p.N + BN - 1) / BN;
    int tm, tn;
    tile_coords(blockIdx.x, tiles_m * tiles_n, tiles_m, tiles_n, tm, tn);
    const int m0 = tm * BM, n0 = tn * BN;

    // per-thread global source rows for the 4 staged chunks of each operand
    const char* a_src[4];
    const char* w_src[4];
    int lds_off[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = tid + i * 256, row = c >> 3, chunk = c & 7;
        const int am = min(m0 + row, p.M - 1), wr = min(n0 + row, p.N - 1);
        a_src[i] = reinterpret_cast<const char*>(p.A + (size_t)am * p.lda) + chunk * 16;
        w_src[i] = reinterpret_cast<const char*>(p.W + (size_t)wr * p.ldw) + chunk * 16;
        lds_off[i] = swz(row, chunk);
    }
    u32x4 ra[4], rw[4];
    auto load_tile = [&](int kt) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            rw[i] = ld16(w_src[i] + w_koff(kt, p));
            ra[i] = ld16(a_src[i] + a_koff(kt, p));
        }
    };
    auto store_tile = [&](int stage) {
        char* ws = smem + stage * (2 * TILE_BYTES);
        char* as = ws + TILE_BYTES;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            st16(ws + lds_off[i], rw[i]);
            st16(as + lds_off[i], ra[i]);
        }
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk = p.K / BK;
    load_tile(0);
    store_tile(0);
    __syncthreads();
    const int frow = lane & 15, fchunk = lane >> 4;
    for (int kt = 0; kt < nk; ++kt) {
        const bool more = kt + 1 < nk;
        if (more) load_tile(kt + 1);
        const char* ws = smem + (kt & 1) * (2 * TILE_BYTES);
        const char* as = ws + TILE_BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            u32x4 fw[4], fa[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                fw[i] = ld16(ws + swz(wn * 64 + i * 16 + frow, ks * 4 + fchunk));
                fa[i] = ld16(as + swz(wm * 64 + i * 16 + frow, ks * 4 + fchunk));
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = mfma16(fw[i], fa[j], acc[i][j]);
        }
        if (more) store_tile((kt + 1) & 1);
        __syncthreads();
    }

    // ---- epilogue: lane holds out[m][n..n+3], m = .. + (lane&15), n = .. + (lane>>4)*4
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int n = n0 + wn * 64 + i * 16 + (lane >> 4) * 4;
        if (n >= p.N) continue;
        f32x4 bv = f32x4{0.f, 0.f, 0.f, 0.f};
        if (p.bias) bv = ld16f(p.bias + n);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int m = m0 + wm * 64 + j * 16 + (lane & 15);
            if (m >= p.M) continue;
            store_out<EPI>(p, m, n, (p.row_scale ? acc[i][j] * p.row_scale[m] : acc[i][j]) + bv);
        }
    }
}

// ---- variant with LDS-DMA staging: same tile / swizzle / MFMA schedule, but the next k-tile is copied global->LDS
// by global_load_lds_dwordx4 (no staging VGPRs, no ds_write_b128 — the ds_write pass was the larger half of this
// kernel's LDS time) while the MFMAs of the current tile run; the compiler's vmcnt(0) before the tile's closing
// barrier retires it.  One wave-instruction fills 8 swizzled rows (1 KiB): lane p writes slot p%8 of row p/8, so it
// FETCHES chunk (p%8)^(row&7) of that row.
// Geometry: WN x WM waves, each owning FI x FJ MFMA tiles (16 features x 16 tokens) -> workgroup tile
// BNt = WN*FI*16 features x BMt = WM*FJ*16 tokens.  <2,2,4,4> = 128x128 / 4 waves / 64 KiB LDS (2 workgroups per CU);
// <2,4,8,4> = 256x256 / 8 waves / 128 KiB LDS (1 workgroup per CU): twice the MFMAs per fragment read and a k-tile
// compute phase (64 MFMAs per wave) long enough to cover the LDS-DMA's HBM latency with a one-tile prefetch distance.
template <int EPI, int WN, int WM, int FI, int FJ>
__global__ __launch_bounds__(WN * WM * 64) void gemm_bf16_dma_kernel(GemmArgs p) {
    constexpr int BNt = WN * FI * 16, BMt = WM * FJ * 16, NW = WN * WM;
    constexpr int W_BYTES = BNt * 128, A_BYTES = BMt * 128, STAGE = W_BYTES + A_BYTES;
    constexpr int WP = BNt / 8 / NW, AP = BMt / 8 / NW;  // 1-KiB DMA pieces per wave per operand tile
    VC_DYNAMIC_SMEM(char, smem);  // [2 stages][W tile | A tile]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wn = wave / WM, wm = wave % WM;
    const int tiles_m = (p.M + BMt - 1) / BMt, tiles_n = (p.N + BNt - 1) / BNt;
    int tm, tn;
    tile_coords(blockIdx.x, tiles_m * tiles_n, tiles_m, tiles_n, tm, tn);
    const int m0 = tm * BMt, n0 = tn * BNt;
    const char* a_src[AP];
    const char* w_src[WP];
    // a 1-KiB piece = 8 swizzled rows; lane p fills slot p%8 of row p/8, i.e. fetches chunk (p%8)^(row&7) of that row
#pragma unroll
    for (int i = 0; i < WP; ++i) {
        const int row = (i * NW + wave) * 8 + (lane >> 3);
        const int wr = min(n0 + row, p.N - 1);
        w_src[i] = reinterpret_cast<const char*>(p.W + (size_t)wr * p.ldw) + (((lane & 7) ^ (row & 7)) << 4);
    }
#pragma unroll
    for (int i = 0; i < AP; ++i) {
        const int row = (i * NW + wave) * 8 + (lane >> 3);
        const int am = min(m0 + row, p.M - 1);
        a_src[i] = reinterpret_cast<const char*>(p.A + (size_t)am * p.lda) + (((lane & 7) ^ (row & 7)) << 4);
    }
    auto issue_tile = [&](int kt, int stage) {
        char* ws = smem + stage * STAGE;
        char* as = ws + W_BYTES;
#pragma unroll
        for (int i = 0; i < WP; ++i) glds16(w_src[i] + w_koff(kt, p), ws + (i * NW + wave) * 1024);
#pragma unroll
        for (int i = 0; i < AP; ++i) glds16(a_src[i] + a_koff(kt, p), as + (i * NW + wave) * 1024);
    };
    f32x4 acc[FI][FJ];
#pragma unroll
    for (int i = 0; i < FI; ++i)
#pragma unroll
        for (int j = 0; j < FJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int nk = p.K / BK;
    issue_tile(0, 0);
    __syncthreads();
    const int frow = lane & 15, fchunk = lane >> 4;
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) issue_tile(kt + 1, (kt + 1) & 1);
        const char* ws = smem + (kt & 1) * STAGE;
        const char* as = ws + W_BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            u32x4 fw[FI], fa[FJ];
#pragma unroll
            for (int i = 0; i < FI; ++i) fw[i] = ld16(ws + swz(wn * (FI * 16) + i * 16 + frow, ks * 4 + fchunk));
#pragma unroll
            for (int j = 0; j < FJ; ++j) fa[j] = ld16(as + swz(wm * (FJ * 16) + j * 16 + frow, ks * 4 + fchunk));
#pragma unroll
            for (int i = 0; i < FI; ++i)
#pragma unroll
                for (int j = 0; j < FJ; ++j) acc[i][j] = mfma16(fw[i], fa[j], acc[i][j]);
        }
        __syncthreads();
    }

    // ---- epilogue: lane holds out[m][n..n+3], m = .. + (lane&15), n = .. + (lane>>4)*4
#pragma unroll
    for (int i = 0; i < FI; ++i) {
        const int n = n0 + wn * (FI * 16) + i * 16 + (lane >> 4) * 4;
        if (n >= p.N) continue;
        f32x4 bv = f32x4{0.f, 0.f, 0.f, 0.f};
        if (p.bias) bv = ld16f(p.bias + n);
#pragma unroll
        for (int j = 0; j < FJ; ++j) {
            const int m = m0 + wm * (FJ * 16) + j * 16 + (lane & 15);
            if (m >= p.M) continue;
            store_out<EPI>(p, m, n, (p.row_scale ? acc[i][j] * p.row_scale[m] : acc[i][j]) + bv);
        }
    }
}


// ---- 256 x 256 tile, 8 waves, counted-vmcnt "8-phase" schedule ----------------------------------------------------------
// The one-barrier loop above drains the LDS-DMA queue (vmcnt(0)) at every k-tile barrier, so HBM/L2 latency is exposed
// once per tile.  Here the DMA stream runs 3 half-tiles ahead and is never drained inside the loop:
//   * a k-tile (BK = 64) is staged as four 16-KiB half-tiles  Y0 | X0 | Y1 | X1  (X = weight rows n, Y = activation rows
//     m; half h = tile rows [128h, 128h+128)), double-buffered: 2 x 64 KiB;
//   * wave (g, q) = (wave/4, wave%4) owns n rows {64g..64g+63} of BOTH X halves and m rows {32q..32q+31} of BOTH Y halves,
//     i.e. a 128 x 64 output made of four 64 x 32 quadrants; one quadrant x BK = 16 MFMAs = one phase;
//   * phase r of k-tile t:  r=0 reads Y0,X0 (12 fragments) and stages X1(t+1);  r=1 reads Y1, stages Y0(t+2);
//     r=2 reads X1, stages X0(t+2);  r=3 reads nothing, stages Y1(t+2) and waits vmcnt(6) — everything except the three
//     most recent half-tiles has landed, which is exactly k-tile t+1.  Every restage targets a half-tile whose last
//     fragment read retired at least one full barrier earlier (Y0: lgkmcnt(8) before the r=0 barrier);
//   * the two wave groups g = 0/1 (the two waves of every SIMD) run one barrier apart, so one group's 16 MFMAs (under
//     s_setprio 1) overlap the other group's fragment reads + DMA issue; barriers are bare s_barrier.
// LDS-DMA data is ordered for a ds_read only by the issuing wave's vmcnt wait followed by a barrier the reader passes;
// the r=3 wait precedes that phase's first barrier and the first read of the retired buffer is a phase later.
//
// F8 = true: the same schedule over OCP e4m3 operands (activations quantised per token row, weights per output row, both
// with power-of-two scales): a 128-byte tile row is now ONE K = 128 step of v_mfma_scale_f32_16x16x128_f8f6f4 (unit
// block scales; twice the bf16 MFMA rate) instead of two K = 32 bf16 steps — identical bytes, LDS layout, DMA pattern
// and MFMA cycles per k-tile, half the k-tiles.  A lane feeds the instruction chunks g and 4+g of its row (g = lane/16) —
// the fragment reads of the bf16 form, which are bank-conflict free; reading the "natural" chunks 2g, 2g+1 measured 50 %
// conflict cycles.  Which 32 of the row's 128 k a lane contributes is free as long as both operands agree: the
// instruction sums over all of them.  The epilogue multiplies the fp32 accumulator by a_scale[m] * w_scale[n] (exact: powers of two).
// KWRAP (precision mode "split"): the weight's k-tile index wraps after p.kwrap tiles (compile-time flag: the default
// instantiation keeps the weight source a plain `base + kt * 128`, exactly the round-2 loop)
// M32 (round 6): the same schedule on v_mfma_f32_32x32x16_bf16.  A wave's 128 x 64 output is 4 x 2 accumulators of 32 x 32 instead
// of 8 x 4 of 16 x 16; a quadrant (64 n x 32 m x BK) is 8 MFMAs of 32 cycles instead of 16 of 16, fed by the same number of
// ds_read_b128 (a fragment = 32 rows x 16 k: lane l reads row l % 32, chunk 2 s + l / 32 of k-step s).  Half the MFMA issues per
// flop, an exact 32-cycle back-to-back cadence (the 16 x 16 x 32 form issues at ~17 instead of 16), and half the operand-register
// reads per flop.  The tile rows are swizzled by (row >> 1) & 7 instead of row & 7: the 16 lanes of a ds_read_b128 group then
// cover rows {0-3, 12-15, 20-27} (or {4-11, 16-19, 28-31}) of ONE chunk column — 8 row pairs with distinct (row >> 1) & 7, two
// parities each = 16 distinct 16-byte bank slots.  C/D: lane l holds token l % 32 and, per register group rg = reg / 4, the 4
// consecutive features 8 rg + 4 (l / 32) + reg % 4 (cdna_hip_programming.md section 3) — the lane-local epilogues are unchanged.
template <bool M32> VC_DEV int swz8(int r, int c) { return r * 128 + ((c ^ ((M32 ? r >> 1 : r) & 7)) << 4); }

template <int EPI, bool F8 = false, bool KWRAP = false, bool M32 = false>
__global__ __launch_bounds__(512) void gemm_bf16_8phase_kernel(GemmArgs p) {
    static_assert(!(F8 && M32), "the 32 x 32 form is instantiated for bf16 operands");
    static_assert(!(EPI == EPI_QKV && (M32 || KWRAP)), "the fused QKV epilogue is built on the 16 x 16 accumulator layout of the bf16 / e4m3 forms");
    constexpr int HALF = 128 * 128, TILE = 4 * HALF;  // bytes
    constexpr int ES = F8 ? 1 : 2;                    // bytes per operand element; a k-tile is 128 bytes of every row
    constexpr int SY0 = 0, SX0 = HALF, SY1 = 2 * HALF, SX1 = 3 * HALF;
    VC_DYNAMIC_SMEM(char, smem);  // [2][Y0 | X0 | Y1 | X1]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = wave >> 2, q = wave & 3;
    const int tiles_m = (p.M + 255) / 256, tiles_n = (p.N + 255) / 256;
    // split-K remainder round: workgroups [sk_full, ...) are sk_ks K-slices of each of the last tiles
    int pid, ks = 0, KS = 1;
    if (p.sk_ks > 1 && (int)blockIdx.x >= p.sk_full) {
        const int r = (int)blockIdx.x - p.sk_full;
        pid = p.sk_full + r / p.sk_ks;
        ks = r % p.sk_ks;
        KS = p.sk_ks;
    } else {
        pid = p.xcd_remap_on ? xcd_remap(blockIdx.x, p.sk_ks > 1 ? p.sk_full : tiles_m * tiles_n) : (int)blockIdx.x;
    }
    int tm, tn;
    tile_from_pid(pid, tiles_m, tiles_n, tm, tn, p.tile_group);
    const int m0 = tm * 256, n0 = tn * 256;
    const int nk_all = p.K * ES / 128;
    const int kt_first = (int)((long)ks * nk_all / KS);
    // DMA sources: every wave moves pieces {wave, 8 + wave} (8 swizzled rows = 1 KiB each) of every half-tile
    const char* x_src[2][2];
    const char* y_src[2][2];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = (i * 8 + wave) * 8 + (lane >> 3);
            const int sw = ((lane & 7) ^ ((M32 ? row >> 1 : row) & 7)) << 4;
            const int wrow = EPI == EPI_QKV ? qkv_wrow(n0, h, row) : n0 + h * 128 + row;
            int arow = min(m0 + h * 128 + row, p.M - 1);
            if constexpr (EPI == EPI_QKV) arow = qkv_arow(p.qe, arow);
            x_src[h][i] = reinterpret_cast<const char*>(p.W) + (size_t)min(wrow, p.N - 1) * p.ldw * ES + sw +
                          (KWRAP ? (size_t)0 : (size_t)kt_first * 128);
            y_src[h][i] = reinterpret_cast<const char*>(p.A) + (size_t)arow * p.lda * ES + sw +
                          (KWRAP ? (size_t)0 : (size_t)kt_first * 128);
        }
    const int nk = (int)((long)(ks + 1) * nk_all / KS) - kt_first;  // k-tiles of this workgroup
    auto stage_x = [&](int h, int kt) {
        if (kt >= nk) return;
        char* dst = smem + (kt & 1) * TILE + (h ? SX1 : SX0) + wave * 1024;
        const long long wk = KWRAP ? w_koff(kt_first + kt, p) : (long long)kt * 128;
        glds16(x_src[h][0] + wk, dst);
        glds16(x_src[h][1] + wk, dst + 8192);
    };
    auto stage_y = [&](int h, int kt) {
        if (kt >= nk) return;
        char* dst = smem + (kt & 1) * TILE + (h ? SY1 : SY0) + wave * 1024;
        const long long ak = KWRAP ? a_koff(kt_first + kt, p) : (long long)kt * 128;
        glds16(y_src[h][0] + ak, dst);
        glds16(y_src[h][1] + ak, dst + 8192);
    };
    f32x4 acc[2][4][2][2];  // [x half][i][y half][j]
    f32x16 acc32[2][2][2];  // M32: [x half][32-row fragment][y half]
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    if constexpr (M32) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc32[a][i >> 1][b][(i & 1) * 8 + j * 4 + e] = 0.f;
                    } else {
                        acc[a][i][b][j] = f32x4{0.f, 0.f, 0.f, 0.f};
                    }
                }
    const int frow = M32 ? lane & 31 : lane & 15, fchunk = M32 ? lane >> 5 : lane >> 4;
    // X0 is dead after phase 1 and X1 is first read in phase 2: one register set serves both halves
    u32x4 fx[4][2], fy[2][2][2];  // [fragment][ks], [half][fragment][ks]; M32: fx[2 fi + s / 2][s & 1], fy[half][s / 2][s & 1]
    auto read_x = [&](const char* base, int h) {
        if constexpr (M32) {
#pragma unroll
            for (int fi = 0; fi < 2; ++fi)
#pragma unroll
                for (int s_ = 0; s_ < 4; ++s_)
                    fx[2 * fi + (s_ >> 1)][s_ & 1] = ld16(base + (h ? SX1 : SX0) + swz8<true>(g * 64 + fi * 32 + frow, 2 * s_ + fchunk));
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
                    fx[i][ks] = ld16(base + (h ? SX1 : SX0) + swz(g * 64 + i * 16 + frow, ks * 4 + fchunk));
        }
    };
    auto read_y = [&](const char* base, int h) {
        if constexpr (M32) {
#pragma unroll
            for (int s_ = 0; s_ < 4; ++s_)
                fy[h][s_ >> 1][s_ & 1] = ld16(base + (h ? SY1 : SY0) + swz8<true>(q * 32 + frow, 2 * s_ + fchunk));
        } else {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
                    fy[h][j][ks] = ld16(base + (h ? SY1 : SY0) + swz(q * 32 + j * 16 + frow, ks * 4 + fchunk));
        }
    };
    auto quadrant = [&](int hx, int hy) {
        set_prio<1>();
        if constexpr (M32) {
#pragma unroll
            for (int s_ = 0; s_ < 4; ++s_)
#pragma unroll
                for (int fi = 0; fi < 2; ++fi)
                    acc32[hx][fi][hy] = mfma32(fx[2 * fi + (s_ >> 1)][s_ & 1], fy[hy][s_ >> 1][s_ & 1], acc32[hx][fi][hy]);
        } else if constexpr (F8) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[hx][i][hy][j] = mfma16_f8(fx[i][0], fx[i][1], fy[hy][j][0], fy[hy][j][1], acc[hx][i][hy][j]);
            // the scaled-MFMA intrinsic is sunk by the optimiser (all 32 of a k-tile end up behind the loop's last barrier,
            // out of their s_setprio brackets) unless its results are pinned here
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) pin_vgprs(acc[hx][i][hy][j]);
        } else {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[hx][i][hy][j] = mfma16(fx[i][ks], fy[hy][j][ks], acc[hx][i][hy][j]);
        }
        set_prio<0>();
    };
    // prologue: k-tile 0 complete, the first three half-tiles of k-tile 1 in flight
    stage_y(0, 0); stage_x(0, 0); stage_y(1, 0); stage_x(1, 0);
    stage_y(0, 1); stage_x(0, 1); stage_y(1, 1);
    if (nk > 1) wait_vmcnt<6>();
    else wait_vmcnt<0>();
    wg_barrier_raw();
    if (g == 1) wg_barrier_raw();  // group 1 runs one barrier behind group 0
    for (int kt = 0; kt < nk; ++kt) {
        const char* base = smem + (kt & 1) * TILE;
        // r = 0
        read_y(base, 0);
        sched_fence();
        read_x(base, 0);
        stage_x(1, kt + 1);
        wait_lgkmcnt<8>();  // the Y0 reads (issued first) have retired: Y0 may be restaged next phase
        wg_barrier_raw();
        wait_lgkmcnt<0>();
        quadrant(0, 0);
        wg_barrier_raw();
        // r = 1
        read_y(base, 1);
        stage_y(0, kt + 2);
        wg_barrier_raw();
        wait_lgkmcnt<0>();
        quadrant(0, 1);
        wg_barrier_raw();
        // r = 2
        read_x(base, 1);
        stage_x(0, kt + 2);
        wg_barrier_raw();
        wait_lgkmcnt<0>();
        quadrant(1, 1);
        wg_barrier_raw();
        // r = 3
        stage_y(1, kt + 2);
        if (kt + 2 < nk) wait_vmcnt<6>();  // all of k-tile kt+1 has landed (3 newer half-tiles may be in flight)
        else wait_vmcnt<0>();
        wg_barrier_raw();
        quadrant(1, 0);
        wg_barrier_raw();
    }
    if (g == 0) wg_barrier_raw();

    if constexpr (M32) {
        // lane holds out[m][n .. n + 3] with m = .. + lane % 32 and n = .. + fi * 32 + rg * 8 + (lane / 32) * 4
        float* wsb = KS > 1 ? p.ws + ((size_t)(pid - p.sk_full) * KS + ks) * 65536 : nullptr;
        float rs32[2] = {1.f, 1.f};
        if (p.row_scale && KS == 1) {
#pragma unroll
            for (int hy = 0; hy < 2; ++hy) rs32[hy] = p.row_scale[min(m0 + hy * 128 + q * 32 + (lane & 31), p.M - 1)];
        }
#pragma unroll
        for (int hx = 0; hx < 2; ++hx)
#pragma unroll
            for (int fi = 0; fi < 2; ++fi)
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) {
                    const int nl = hx * 128 + g * 64 + fi * 32 + rg * 8 + (lane >> 5) * 4;
                    f32x4 bv = f32x4{0.f, 0.f, 0.f, 0.f};
                    if (KS == 1 && p.bias && n0 + nl < p.N) bv = ld16f(p.bias + n0 + nl);
#pragma unroll
                    for (int hy = 0; hy < 2; ++hy) {
                        const int ml = hy * 128 + q * 32 + (lane & 31);
                        const f32x4 v = f32x4{acc32[hx][fi][hy][rg * 4], acc32[hx][fi][hy][rg * 4 + 1], acc32[hx][fi][hy][rg * 4 + 2],
                                              acc32[hx][fi][hy][rg * 4 + 3]};
                        if (KS > 1) st16f(wsb + ml * 256 + nl, v);
                        else if constexpr (EPI != EPI_QKV) {
                            if (n0 + nl < p.N && m0 + ml < p.M) store_out<EPI>(p, m0 + ml, n0 + nl, v * rs32[hy] + bv);
                        }
                    }
                }
        return;
    }
    if (KS > 1) {  // partial tile -> workspace [remainder tile][slice][256 m][256 n]; the fix-up launch finishes it
        float* base = p.ws + ((size_t)(pid - p.sk_full) * KS + ks) * 65536;
#pragma unroll
        for (int hx = 0; hx < 2; ++hx)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int hy = 0; hy < 2; ++hy)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        st16f(base + (hy * 128 + q * 32 + j * 16 + (lane & 15)) * 256 + hx * 128 + g * 64 + i * 16 + (lane >> 4) * 4,
                              acc[hx][i][hy][j]);
        return;
    }
    if constexpr (EPI == EPI_QKV) {
        // lane holds, for its wave group's head, features d = hx * 64 + i * 16 + G * 4 + (0..3) of tokens hy * 128 + q * 32 + j * 16 + l15
        const QkvEpiArgs& e = p.qe;
        const int D = e.H * 128;
        const int which = n0 / D, head = (n0 - which * D) / 128 + g;
        const int G = lane >> 4, l15 = lane & 15;
        int tb[2][2], tt[2][2];
        bool ok[2][2];
        float rs[2][2];
#pragma unroll
        for (int hy = 0; hy < 2; ++hy)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int mp = m0 + hy * 128 + q * 32 + j * 16 + l15;
                tb[hy][j] = mp / e.Tp;
                tt[hy][j] = mp - tb[hy][j] * e.Tp;
                ok[hy][j] = mp < p.M && tt[hy][j] < e.T;
                const int src = qkv_arow(e, min(mp, p.M - 1));
                rs[hy][j] = F8 ? p.a_scale[src] : (p.row_scale ? p.row_scale[src] : 1.f);
            }
        if (which < 2) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int dl = i * 16 + G * 4, nx = n0 + g * 128 + dl;   // nx: the weight row (= output column) of x; y is 64 further
                f32x4 bx = f32x4{0.f, 0.f, 0.f, 0.f}, by = bx, swx = f32x4{1.f, 1.f, 1.f, 1.f}, swy = swx;
                if (p.bias) {
                    bx = ld16f(p.bias + nx);
                    by = ld16f(p.bias + nx + 64);
                }
                if constexpr (F8) {
                    swx = ld16f(p.w_scale + nx);
                    swy = ld16f(p.w_scale + nx + 64);
                }
                // the cos / sin rows of the lane's four tokens: requested together at a clamped token and waited for OUTSIDE the
                // guards (round 6, as the RESID epilogue below: inside the guard every store paid its own table round trip and the
                // drain of the stores before it)
                f32x4 csv[2][2], snv[2][2];
#pragma unroll
                for (int hy = 0; hy < 2; ++hy)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const size_t tc = (size_t)min(tt[hy][j], e.T - 1) * 64 + dl;
                        csv[hy][j] = ld16f(e.rope_cos + tc);
                        snv[hy][j] = ld16f(e.rope_sin + tc);
                    }
#pragma unroll
                for (int hy = 0; hy < 2; ++hy)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        pin_vgprs(csv[hy][j]);
                        pin_vgprs(snv[hy][j]);
                    }
                pin_vgprs(bx);
                pin_vgprs(by);
                if constexpr (F8) {
                    pin_vgprs(swx);
                    pin_vgprs(swy);
                }
#pragma unroll
                for (int hy = 0; hy < 2; ++hy)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        if (ok[hy][j])
                            qkv_rope_store(e, which, head, tb[hy][j], tt[hy][j], dl, acc[0][i][hy][j] * (swx * rs[hy][j]) + bx,
                                           acc[1][i][hy][j] * (swy * rs[hy][j]) + by, csv[hy][j], snv[hy][j]);
            }
        } else {
#pragma unroll
            for (int hx = 0; hx < 2; ++hx)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int d0 = hx * 64 + i * 16 + G * 4, nv = n0 + g * 128 + d0;
                    f32x4 bv = f32x4{0.f, 0.f, 0.f, 0.f}, sw = f32x4{1.f, 1.f, 1.f, 1.f};
                    if (p.bias) bv = ld16f(p.bias + nv);
                    if constexpr (F8) sw = ld16f(p.w_scale + nv);
                    pin_vgprs(bv);
                    if constexpr (F8) pin_vgprs(sw);
#pragma unroll
                    for (int hy = 0; hy < 2; ++hy) {
                        u32x2 vb[2];
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            vb[j] = pack_bf4(acc[hx][i][hy][j] * (sw * rs[hy][j]) + bv);
                            if (ok[hy][j]) qkv_v_store(e, head, tb[hy][j], tt[hy][j], d0, vb[j]);
                            else vb[j] = u32x2{0u, 0u};   // keys behind the sequence are zeros in the V^T scratch
                        }
                        // P[f] = feature d0 + f of the lane's two tokens (keys kk = l15 and 16 + l15 of the 32-key block); after the
                        // quad transpose lane r = l15 & 3 holds feature d0 + r of keys 4c .. 4c + 3 and 16 + 4c .. 16 + 4c + 3
                        // (c = l15 >> 2): chunk c of the block in the flash kernel's key order — one 16-byte store
                        uint32_t P[4] = {(vb[0][0] & 0xFFFFu) | (vb[1][0] << 16), (vb[0][0] >> 16) | (vb[1][0] & 0xFFFF0000u),
                                         (vb[0][1] & 0xFFFFu) | (vb[1][1] << 16), (vb[0][1] >> 16) | (vb[1][1] & 0xFFFF0000u)};
                        quad_transpose4(P, l15 & 3);
                        const int base = m0 + hy * 128 + q * 32, bb = base / e.Tp, t0 = base - bb * e.Tp, c = l15 >> 2;
                        if (base < p.M && t0 + 4 * c < e.T)
                            st16(e.vt + (((size_t)bb * e.H + head) * 128 + d0 + (l15 & 3)) * e.vt_stride + t0 + 8 * c,
                                 u32x4{(P[0] & 0xFFFFu) | (P[1] << 16), (P[2] & 0xFFFFu) | (P[3] << 16), (P[0] >> 16) | (P[1] & 0xFFFF0000u),
                                       (P[2] >> 16) | (P[3] & 0xFFFF0000u)});
                    }
                }
        }
        return;
    }
    // ---- epilogue: lane holds out[m][n..n+3]
    float rsc[2][2] = {{1.f, 1.f}, {1.f, 1.f}};   // folded RMSNorm, consumer side: the row scales of this lane's four token rows
    if (p.row_scale) {
#pragma unroll
        for (int hy = 0; hy < 2; ++hy)
#pragma unroll
            for (int j = 0; j < 2; ++j) rsc[hy][j] = p.row_scale[min(m0 + hy * 128 + q * 32 + j * 16 + (lane & 15), p.M - 1)];
    }
    // Everything the stores need from memory is requested AHEAD of the bounds guards, unconditionally at clamped addresses and in
    // batches (round 6, ISA): a load inside a guard is waited for with vmcnt(0) where the guard ends, so the RESID epilogue of
    // rounds 1-5 paid 32 DEPENDENT residual round trips per lane (a quarter of the o_proj launch: MFMA-busy 49 % where the other
    // epilogues' kernels show 60-67 %), the bias / weight-scale loads 8.  Now: the column operands of a tile half together, the
    // residual values of two column groups (8 loads) together.
    float asc[2][2] = {{1.f, 1.f}, {1.f, 1.f}};   // W8A8: the activation rows' scales
    if constexpr (F8) {
#pragma unroll
        for (int hy = 0; hy < 2; ++hy)
#pragma unroll
            for (int j = 0; j < 2; ++j) asc[hy][j] = p.a_scale[min(m0 + hy * 128 + q * 32 + j * 16 + (lane & 15), p.M - 1)];
    }
#pragma unroll
    for (int hx = 0; hx < 2; ++hx) {
        f32x4 bvh[4], swh[4], gwh[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int nc = min(n0 + hx * 128 + g * 64 + i * 16 + (lane >> 4) * 4, p.N - 4);
            bvh[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            swh[i] = f32x4{1.f, 1.f, 1.f, 1.f};
            if (p.bias) bvh[i] = ld16f(p.bias + nc);          // (wave-uniform test: the four loads of the half go out together)
            if constexpr (F8) swh[i] = ld16f(p.w_scale + nc);
            if constexpr (EPI == EPI_RESID_F32) {             // folded-RMSNorm producer: the consumer's norm weights of these columns
                gwh[i] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (p.xg_out) gwh[i] = ld16f(p.xg_w + nc);
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            pin_vgprs(bvh[i]);
            if constexpr (F8) pin_vgprs(swh[i]);
            if constexpr (EPI == EPI_RESID_F32) pin_vgprs(gwh[i]);
        }
#pragma unroll
        for (int ip = 0; ip < 2; ++ip) {
            f32x4 res[2][2][2];
            if constexpr (EPI == EPI_RESID_F32) {
#pragma unroll
                for (int ii = 0; ii < 2; ++ii)
#pragma unroll
                    for (int hy = 0; hy < 2; ++hy)
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            const int nc = min(n0 + hx * 128 + g * 64 + (ip * 2 + ii) * 16 + (lane >> 4) * 4, p.N - 4);
                            const int mc = min(m0 + hy * 128 + q * 32 + j * 16 + (lane & 15), p.M - 1);
                            res[ii][hy][j] = ld16f(reinterpret_cast<const float*>(p.out) + (size_t)mc * p.ldo + nc);
                        }
                // ... and waited for HERE, outside the guards (an empty asm that takes the registers): a first use inside a guard
                // leaves them pending on the path around it, and every later guarded use then drains the stores issued since
#pragma unroll
                for (int ii = 0; ii < 2; ++ii)
#pragma unroll
                    for (int hy = 0; hy < 2; ++hy)
#pragma unroll
                        for (int j = 0; j < 2; ++j) pin_vgprs(res[ii][hy][j]);
            }
#pragma unroll
            for (int ii = 0; ii < 2; ++ii) {
                const int i = ip * 2 + ii;
                const int n = n0 + hx * 128 + g * 64 + i * 16 + (lane >> 4) * 4;
                if (n >= p.N) continue;
#pragma unroll
                for (int hy = 0; hy < 2; ++hy)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const int m = m0 + hy * 128 + q * 32 + j * 16 + (lane & 15);
                        if (m >= p.M) continue;
                        if constexpr (EPI == EPI_QKV) {
                        } else if constexpr (F8) store_out<EPI>(p, m, n, acc[hx][i][hy][j] * (swh[i] * asc[hy][j]) + bvh[i]);
                        else if constexpr (EPI == EPI_RESID_F32) store_out<EPI>(p, m, n, acc[hx][i][hy][j] * rsc[hy][j] + bvh[i], &res[ii][hy][j], &gwh[i]);
                        else store_out<EPI>(p, m, n, acc[hx][i][hy][j] * rsc[hy][j] + bvh[i]);
                    }
            }
        }
    }
}

// sums the K-slices of the split remainder tiles in k order and applies the epilogue (one thread = out[m][n..n+3])
template <int EPI>
__global__ __launch_bounds__(256) void gemm_splitk_fixup_kernel(GemmArgs p) {
    const int tiles_m = (p.M + 255) / 256, tiles_n = (p.N + 255) / 256;
    const int r = blockIdx.x >> 6;
    const int idx = ((int)(blockIdx.x & 63) << 8) + threadIdx.x;
    const int ml = idx >> 6, nl = (idx & 63) << 2;
    int tm, tn;
    tile_from_pid(p.sk_full + r, tiles_m, tiles_n, tm, tn, p.tile_group);
    const int m = tm * 256 + ml, n = tn * 256 + nl;
    if (m >= p.M || n >= p.N) return;
    const float* base = p.ws + (size_t)r * p.sk_ks * 65536 + ml * 256 + nl;
    f32x4 v = ld16f(base);
    for (int k = 1; k < p.sk_ks; ++k) v = v + ld16f(base + (size_t)k * 65536);
    if (p.f8) v = v * (ld16f(p.w_scale + n) * p.a_scale[m]);
    if (p.row_scale) v = v * p.row_scale[m];
    if (p.bias) v = v + ld16f(p.bias + n);
    store_out<EPI, true>(p, m, n, v);
}


// ... and of the fused-QKV epilogue: one thread = the rotate-half pairs (x at tile column c, y at column 128 + c) of 4 features of
// one token of a remainder tile; V^T elements are stored one by one here (a few remainder tiles per launch)
__global__ __launch_bounds__(256) void gemm_splitk_fixup_qkv_kernel(GemmArgs p) {
    const QkvEpiArgs& e = p.qe;
    const int tiles_m = (p.M + 255) / 256, tiles_n = (p.N + 255) / 256;
    const int r = blockIdx.x >> 5;
    const int idx = ((int)(blockIdx.x & 31) << 8) + threadIdx.x;
    const int ml = idx >> 5, cx = (idx & 31) << 2;   // cx: tile column of x (0..124) = g * 64 + dl
    int tm, tn;
    tile_from_pid(p.sk_full + r, tiles_m, tiles_n, tm, tn, p.tile_group);
    const int mp = tm * 256 + ml, n0 = tn * 256;
    const int b = mp / e.Tp, t = mp - b * e.Tp;
    if (mp >= p.M || t >= e.T) return;
    const float* base = p.ws + (size_t)r * p.sk_ks * 65536 + ml * 256 + cx;
    f32x4 x = ld16f(base), y = ld16f(base + 128);
    for (int k = 1; k < p.sk_ks; ++k) {
        x = x + ld16f(base + (size_t)k * 65536);
        y = y + ld16f(base + (size_t)k * 65536 + 128);
    }
    const int D = e.H * 128, which = n0 / D, g = cx >> 6, dl = cx & 63, head = (n0 - which * D) / 128 + g, nx = n0 + g * 128 + dl;
    const int src = b * e.T + t;
    if (p.f8) {
        x = x * (ld16f(p.w_scale + nx) * p.a_scale[src]);
        y = y * (ld16f(p.w_scale + nx + 64) * p.a_scale[src]);
    }
    if (p.row_scale) {
        x = x * p.row_scale[src];
        y = y * p.row_scale[src];
    }
    if (p.bias) {
        x = x + ld16f(p.bias + nx);
        y = y + ld16f(p.bias + nx + 64);
    }
    if (which < 2) {
        qkv_rope_store(e, which, head, b, t, dl, x, y);
        return;
    }
    const u32x2 vx = pack_bf4(x), vy = pack_bf4(y);
    qkv_v_store(e, head, b, t, dl, vx);
    qkv_v_store(e, head, b, t, 64 + dl, vy);
    bf16_t* vt = e.vt + (((size_t)b * e.H + head) * 128) * e.vt_stride + (t & ~31) + vt_pos32(t & 31);
#pragma unroll
    for (int f = 0; f < 4; ++f) {
        vt[(size_t)(dl + f) * e.vt_stride] = (bf16_t)(f & 1 ? vx[f >> 1] >> 16 : vx[f >> 1] & 0xFFFFu);
        vt[(size_t)(64 + dl + f) * e.vt_stride] = (bf16_t)(f & 1 ? vy[f >> 1] >> 16 : vy[f >> 1] & 0xFFFFu);
    }
    if (t == e.T - 1)   // the keys behind the sequence inside its last 32-key block are zeros (as the tile epilogue writes them)
        for (int kk = (t & 31) + 1; kk < 32; ++kk)
#pragma unroll
            for (int f = 0; f < 4; ++f) {
                e.vt[(((size_t)b * e.H + head) * 128 + dl + f) * e.vt_stride + (t & ~31) + vt_pos32(kk)] = 0;
                e.vt[(((size_t)b * e.H + head) * 128 + 64 + dl + f) * e.vt_stride + (t & ~31) + vt_pos32(kk)] = 0;
            }
}

// rstd of every row from the sum-of-squares partials a RESID epilogue published (one wave per row, fixed order)
__global__ __launch_bounds__(256) void rstd_from_partials_kernel(const float* ssq, int npart, int nparts, float* rstd, int rows, int D,
                                                                 float eps) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    float s = 0.f;
    for (int i = lane; i < nparts; i += 64) s += ssq[(size_t)row * npart + i];
    s = wave_sum(s);
    if (lane == 0) rstd[row] = rsqrtf(s / (float)D + eps);
}
void launch_rstd_from_partials(const float* ssq, int npart, int nparts, float* rstd, int rows, int D, float eps, hipStream_t s) {
    VC_LAUNCH(rstd_from_partials_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, ssq, npart, nparts, rstd, rows, D, eps);
}

template <class K>
static void allow_big_lds_gemm(K kernel, size_t bytes) {
#ifndef VC_EMU
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
#endif
}

// e4m3 x e4m3 (GemmArgs::f8): always the 8-phase kernel, any size (rows are clamped, stores masked)
static void launch_gemm_f8(const GemmArgs& a, int epilogue, hipStream_t s) {
    const long t256 = (long)((a.M + 255) / 256) * ((a.N + 255) / 256);
    const size_t sh2 = 2 * (256 * 128 + 256 * 128);
    static const int sk_on = getenv("VC_GEMM_SPLITK") ? atoi(getenv("VC_GEMM_SPLITK")) : 1;
    constexpr int tile_group = 4, xcd_on = 1;   // as in launch_gemm below
    GemmArgs ask = a;
    ask.tile_group = tile_group;
    ask.xcd_remap_on = xcd_on;
    ask.sk_full = (int)t256;
    ask.sk_ks = 1;
    const long rem = t256 % 256;
    if (sk_on && a.ws && t256 > 256 && rem > 0 && rem <= 128) {
        int ks = (int)std::min<long>(256 / rem, 8);
        ks = std::min(ks, a.K / 128);
        while (ks > 1 && (size_t)rem * ks * 65536 * 4 > a.ws_bytes) --ks;
        if (ks > 1) {
            ask.sk_full = (int)(t256 - rem);
            ask.sk_ks = ks;
        }
    }
    const dim3 gsk((unsigned)(ask.sk_ks > 1 ? ask.sk_full + rem * ask.sk_ks : t256)), b2(512);
#define VC_G8(E)                                                                                       \
    do {                                                                                               \
        static bool once = false;                                                                      \
        if (!once) {                                                                                   \
            allow_big_lds_gemm(gemm_bf16_8phase_kernel<E, true>, sh2);                                 \
            once = true;                                                                               \
        }                                                                                              \
        VC_LAUNCH((gemm_bf16_8phase_kernel<E, true>), gsk, b2, sh2, s, ask);                           \
        if (ask.sk_ks > 1)                                                                             \
            VC_LAUNCH((gemm_splitk_fixup_kernel<E>), dim3((unsigned)(rem * 64)), dim3(256), 0, s, ask); \
    } while (0)
    if (epilogue == EPI_QKV) {
        static bool once = false;
        if (!once) {
            allow_big_lds_gemm(gemm_bf16_8phase_kernel<EPI_QKV, true>, sh2);
            once = true;
        }
        VC_LAUNCH((gemm_bf16_8phase_kernel<EPI_QKV, true>), gsk, b2, sh2, s, ask);
        if (ask.sk_ks > 1) VC_LAUNCH(gemm_splitk_fixup_qkv_kernel, dim3((unsigned)(rem * 32)), dim3(256), 0, s, ask);
        return;
    }
    switch (epilogue) {
        case EPI_BF16: VC_G8(EPI_BF16); break;
        case EPI_RESID_F32: VC_G8(EPI_RESID_F32); break;
        case EPI_SWIGLU: VC_G8(EPI_SWIGLU); break;
        default: throw std::runtime_error("fp8 GEMM: epilogue not instantiated");
    }
#undef VC_G8
}

// test / benchmark hook (vck_set_gemm_variant): overrides VC_GEMM_VARIANT inside one process; < 0 = the environment's
static int g_gemm_variant = -1;
void set_gemm_variant(int v) { g_gemm_variant = v; }

void launch_gemm(const GemmArgs& a, int epilogue, hipStream_t s) {
    if (a.xg_out && (epilogue != EPI_RESID_F32 || a.N % 16 != 0 || !a.xg_w || !a.ssq_out || a.npart < a.N / 16))
        throw std::runtime_error("gemm: the folded-RMSNorm producer needs EPI_RESID_F32, N % 16 == 0, xg_w, ssq_out, npart >= N / 16");
    if (epilogue == EPI_QKV) {
        const QkvEpiArgs& e = a.qe;
        if (a.kwrap > 0 || a.xg_out || e.H <= 0 || (e.H * 128) % 256 != 0 || a.N != 3 * e.H * 128 || e.Tp % 32 != 0 || e.Tp < e.T ||
            a.M != e.B * e.Tp || !e.q || !e.k || !e.vt || !e.rope_cos || !e.rope_sin || (!e.v && !e.v8))
            throw std::runtime_error("gemm: EPI_QKV needs hd 128, D % 256 == 0, N = 3 D, M = B * Tp with Tp = rup(T, 32), q / k / v / vt and the RoPE tables");
    }
    if (a.f8) return launch_gemm_f8(a, epilogue, s);
    if (epilogue == EPI_QKV) {   // always the 8-phase 256 x 256 kernel on the 16 x 16 x 32 MFMA (any size: rows are clamped, stores masked)
        const long t256 = (long)((a.M + 255) / 256) * ((a.N + 255) / 256);
        const size_t sh2 = 2 * (256 * 128 + 256 * 128);
        static const int sk_on = getenv("VC_GEMM_SPLITK") ? atoi(getenv("VC_GEMM_SPLITK")) : 1;
        GemmArgs ask = a;
        ask.tile_group = 4;
        ask.xcd_remap_on = 1;
        ask.sk_full = (int)t256;
        ask.sk_ks = 1;
        const long rem = t256 % 256;
        if (sk_on && a.ws && t256 > 256 && rem > 0 && rem <= 128) {
            int ks = (int)std::min<long>(256 / rem, 8);
            ks = std::min(ks, a.K / BK);
            while (ks > 1 && (size_t)rem * ks * 65536 * 4 > a.ws_bytes) --ks;
            if (ks > 1) {
                ask.sk_full = (int)(t256 - rem);
                ask.sk_ks = ks;
            }
        }
        static bool once = false;
        if (!once) {
            allow_big_lds_gemm(gemm_bf16_8phase_kernel<EPI_QKV>, sh2);
            once = true;
        }
        VC_LAUNCH((gemm_bf16_8phase_kernel<EPI_QKV>), dim3((unsigned)(ask.sk_ks > 1 ? ask.sk_full + rem * ask.sk_ks : t256)), dim3(512), sh2, s, ask);
        if (ask.sk_ks > 1) VC_LAUNCH(gemm_splitk_fixup_qkv_kernel, dim3((unsigned)(rem * 32)), dim3(256), 0, s, ask);
        return;
    }
    const int tiles = ((a.M + BM - 1) / BM) * ((a.N + BN - 1) / BN);
    const dim3 grid(tiles), block(256);
    const size_t shmem = 4 * TILE_BYTES;
    // default (1): 256x256 counted-vmcnt 8-phase kernel for large problems, 128x128 LDS-DMA kernel for small ones.
    // Tuning / regression knobs (VC_GEMM_VARIANT): 0 register-staged 128x128 (8-16 % slower than the DMA form of the
    // same geometry); 2 force the one-barrier 256x256 DMA kernel (947-1087 TFLOP/s where the 8-phase schedule reaches
    // 1068-1281); 3 force DMA 128x128; 4 force 256x256 as 4 waves x (128x128) (1 wave per SIMD: 20-25 % slower than 2
    // with this simple loop); 5 force the 8-phase kernel for every size
    static const int env_variant = getenv("VC_GEMM_VARIANT") ? atoi(getenv("VC_GEMM_VARIANT")) : 1;
    int variant = g_gemm_variant >= 0 ? g_gemm_variant : env_variant;
    // 6 / 7: the 8-phase kernel on the 32 x 32 x 16 MFMA for large problems / for every size (the folded-RMSNorm producer's row
    // sums assume the 16 x 16 accumulator layout: it keeps the 16 x 16 form)
    const bool m32 = (variant == 6 || variant == 7) && !a.xg_out;
    if (variant == 6) variant = 1;
    if (variant == 7) variant = 5;
    if (variant >= 1) {
        const long t256 = (long)((a.M + 255) / 256) * ((a.N + 255) / 256);
        const bool big = variant == 2 || variant == 4 || variant == 5 || (variant == 1 && a.M >= 1024 && a.N >= 512);
        if (big) {
            const size_t sh2 = 2 * (256 * 128 + 256 * 128);
            const bool wide = variant == 4;  // 4 waves x (128 x 128): fewer LDS fragment reads per MFMA, 1 wave per SIMD
            const bool phased = variant == 5 || variant == 1;  // counted-vmcnt 8-phase schedule
            const dim3 g2((unsigned)t256), b2(wide ? 256 : 512);
            // split-K for a short last round of the 8-phase kernel (VC_GEMM_SPLITK=0 disables): rem tiles left over
            // after the full rounds of 256 are cut into ks = 256 / rem K-slices each, so the round is ~1/ks as long
            static const int sk_on = getenv("VC_GEMM_SPLITK") ? atoi(getenv("VC_GEMM_SPLITK")) : 1;
            // m-tiles per sweep group (measured: 4 and 2 beat 8 / 16 / 38 by 2-6 %) and the XCD remap of the tile order (on: +-3 %)
            constexpr int tile_group = 4, xcd_on = 1;
            GemmArgs ask = a;
            ask.tile_group = tile_group;
            ask.xcd_remap_on = xcd_on;
            ask.sk_full = (int)t256;
            ask.sk_ks = 1;
            long rem = t256 % 256;
            if (phased && sk_on && a.ws && t256 > 256 && rem > 0 && rem <= 128) {
                int ks = (int)std::min<long>(256 / rem, 8);
                ks = std::min(ks, a.K / BK);
                while (ks > 1 && (size_t)rem * ks * 65536 * 4 > a.ws_bytes) --ks;
                if (ks > 1) {
                    ask.sk_full = (int)(t256 - rem);
                    ask.sk_ks = ks;
                }
            }
            const dim3 gsk((unsigned)(ask.sk_ks > 1 ? ask.sk_full + rem * ask.sk_ks : t256));
#define VC_G256(E)                                                                                         \
    do {                                                                                                   \
        static bool once = false;                                                                          \
        if (!once) {                                                                                       \
            allow_big_lds_gemm(gemm_bf16_dma_kernel<E, 2, 4, 8, 4>, sh2);                                  \
            allow_big_lds_gemm(gemm_bf16_dma_kernel<E, 2, 2, 8, 8>, sh2);                                  \
            allow_big_lds_gemm(gemm_bf16_8phase_kernel<E>, sh2);                                           \
            allow_big_lds_gemm(gemm_bf16_8phase_kernel<E, false, true>, sh2);                              \
            allow_big_lds_gemm(gemm_bf16_8phase_kernel<E, false, false, true>, sh2);                       \
            allow_big_lds_gemm(gemm_bf16_8phase_kernel<E, false, true, true>, sh2);                        \
            once = true;                                                                                   \
        }                                                                                                  \
        if (phased && m32) {                                                                               \
            if (ask.kwrap > 0) VC_LAUNCH((gemm_bf16_8phase_kernel<E, false, true, true>), gsk, b2, sh2, s, ask);   \
            else VC_LAUNCH((gemm_bf16_8phase_kernel<E, false, false, true>), gsk, b2, sh2, s, ask);        \
            if (ask.sk_ks > 1)                                                                             \
                VC_LAUNCH((gemm_splitk_fixup_kernel<E>), dim3((unsigned)(rem * 64)), dim3(256), 0, s, ask); \
        } else if (phased && ask.kwrap > 0) {                                                              \
            VC_LAUNCH((gemm_bf16_8phase_kernel<E, false, true>), gsk, b2, sh2, s, ask);                    \
            if (ask.sk_ks > 1)                                                                             \
                VC_LAUNCH((gemm_splitk_fixup_kernel<E>), dim3((unsigned)(rem * 64)), dim3(256), 0, s, ask); \
        } else if (phased) {                                                                               \
            VC_LAUNCH((gemm_bf16_8phase_kernel<E>), gsk, b2, sh2, s, ask);                                 \
            if (ask.sk_ks > 1)                                                                             \
                VC_LAUNCH((gemm_splitk_fixup_kernel<E>), dim3((unsigned)(rem * 64)), dim3(256), 0, s, ask); \
        }                                                                                                  \
        else if (wide) VC_LAUNCH((gemm_bf16_dma_kernel<E, 2, 2, 8, 8>), g2, b2, sh2, s, a);                \
        else VC_LAUNCH((gemm_bf16_dma_kernel<E, 2, 4, 8, 4>), g2, b2, sh2, s, a);                          \
    } while (0)
            switch (epilogue) {
                case EPI_BF16: VC_G256(EPI_BF16); break;
                case EPI_BF16_QGELU: VC_G256(EPI_BF16_QGELU); break;
                case EPI_BF16_GELU: VC_G256(EPI_BF16_GELU); break;
                case EPI_F32: VC_G256(EPI_F32); break;
                case EPI_RESID_F32: VC_G256(EPI_RESID_F32); break;
                default: VC_G256(EPI_SWIGLU); break;
            }
#undef VC_G256
            return;
        }
        switch (epilogue) {
            case EPI_BF16: VC_LAUNCH((gemm_bf16_dma_kernel<EPI_BF16, 2, 2, 4, 4>), grid, block, shmem, s, a); break;
            case EPI_BF16_QGELU: VC_LAUNCH((gemm_bf16_dma_kernel<EPI_BF16_QGELU, 2, 2, 4, 4>), grid, block, shmem, s, a); break;
            case EPI_BF16_GELU: VC_LAUNCH((gemm_bf16_dma_kernel<EPI_BF16_GELU, 2, 2, 4, 4>), grid, block, shmem, s, a); break;
            case EPI_F32: VC_LAUNCH((gemm_bf16_dma_kernel<EPI_F32, 2, 2, 4, 4>), grid, block, shmem, s, a); break;
            case EPI_RESID_F32: VC_LAUNCH((gemm_bf16_dma_kernel<EPI_RESID_F32, 2, 2, 4, 4>), grid, block, shmem, s, a); break;
            default: VC_LAUNCH((gemm_bf16_dma_kernel<EPI_SWIGLU, 2, 2, 4, 4>), grid, block, shmem, s, a); break;
        }
        return;
    }
    switch (epilogue) {
        case EPI_BF16: VC_LAUNCH((gemm_bf16_kernel<EPI_BF16>), grid, block, shmem, s, a); break;
        case EPI_BF16_QGELU: VC_LAUNCH((gemm_bf16_kernel<EPI_BF16_QGELU>), grid, block, shmem, s, a); break;
        case EPI_BF16_GELU: VC_LAUNCH((gemm_bf16_kernel<EPI_BF16_GELU>), grid, block, shmem, s, a); break;
        case EPI_F32: VC_LAUNCH((gemm_bf16_kernel<EPI_F32>), grid, block, shmem, s, a); break;
        case EPI_RESID_F32: VC_LAUNCH((gemm_bf16_kernel<EPI_RESID_F32>), grid, block, shmem, s, a); break;
        default: VC_LAUNCH((gemm_bf16_kernel<EPI_SWIGLU>), grid, block, shmem, s, a); break;
    }
}

}  // namespace vc
