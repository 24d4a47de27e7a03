// decode.hip — weight-streaming skinny GEMM for the decode steps (M <= 32 token rows per weight pass: a session's batch
// or the rows of the shared decode pool), the fp8 quantisers of the W8A16 / W8A8 formats, weight packing.
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
#include <stdio.h>
#include <stdlib.h>

#include <atomic>
#include <stdexcept>
#include <string>

#include "vc_device.h"
#include "kernels.h"

namespace vc {


// shared epilogue: `v` = out[m][n..n+3] partial sums already reduced over the waves of the workgroup
// `m` = token row (0 .. 16*MG-1), `SSW` = row slots per wave in ss_part (16 * MG)
// `pre` (RESID epilogue): the residual values out[m][n..n+3] and the norm weights xg_w[n..n+3], requested by the caller ahead of time
struct GemvResidPre {
    f32x4 o, gw;
    bool have;
};
template <int WAVES, int EPI, bool FP8, int SSW = 16>
VC_DEV void gemv_epilogue(const GemvArgs& p, f32x4 v, const float* ss_part /*[WAVES][SSW]*/, int nt, int m, int g, bool mvalid,
                          const GemvResidPre* pre = nullptr) {
    const int n = nt * 16 + g * 4;  // lane holds out[m][n..n+3]
    if constexpr (FP8) v = v * ld16f(p.wscale + n);
    if (p.ssq_in != nullptr) {
        float ss = ss_part[m];
#pragma unroll
        for (int w = 1; w < WAVES; ++w) ss += ss_part[w * SSW + m];
        v = v * rsqrtf(ss / (float)p.K + p.eps);
    }
    if constexpr (EPI == GEMV_RESID_F32) {
        float* o = reinterpret_cast<float*>(p.out) + (size_t)(mvalid ? m : 0) * p.ldo + n;
        if (mvalid) {
            v = (pre != nullptr && pre->have ? pre->o : ld16f(o)) + v;
            st16f(o, v);
            if (p.xg_out) {  // the consumer's operand: bf16(x * g) of the updated residual values
                const f32x4 gw = pre != nullptr && pre->have ? pre->gw : ld16f(p.xg_w + n);
                const f32x4 t = {v[0] * gw[0], v[1] * gw[1], v[2] * gw[2], v[3] * gw[3]};
                const u32x2 hi = {pack_bf2(t[0], t[1]), pack_bf2(t[2], t[3])};
                st8(p.xg_out + (size_t)m * p.N + n, hi);
                if (p.split_rows)  // split mode: the lo row of the stacked group, bf16(t - hi)
                    st8(p.xg_out + (size_t)(m + p.split_rows) * p.N + n,
                        u32x2{pack_bf2(t[0] - bf2f_lo(hi[0]), t[1] - bf2f_hi(hi[0])), pack_bf2(t[2] - bf2f_lo(hi[1]), t[3] - bf2f_hi(hi[1]))});
            }
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
        const u32x2 hi = {pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3])};
        st8(reinterpret_cast<bf16_t*>(p.out) + (size_t)m * p.ldo + n, hi);
        if (p.split_rows)
            st8(reinterpret_cast<bf16_t*>(p.out) + (size_t)(m + p.split_rows) * p.ldo + n,
                u32x2{pack_bf2(v[0] - bf2f_lo(hi[0]), v[1] - bf2f_hi(hi[0])), pack_bf2(v[2] - bf2f_lo(hi[1]), v[3] - bf2f_hi(hi[1]))});
    } else if constexpr (EPI == GEMV_F32) {
        st16f(reinterpret_cast<float*>(p.out) + (size_t)m * p.ldo + n, v);
    } else if constexpr (EPI == GEMV_SWIGLU) {
        const float h0 = silu(v[0]) * v[1], h1 = silu(v[2]) * v[3];
        const uint32_t hi = pack_bf2(h0, h1);
        *reinterpret_cast<uint32_t*>(reinterpret_cast<bf16_t*>(p.out) + (size_t)m * p.ldo + (n >> 1)) = hi;
        if (p.split_rows)
            *reinterpret_cast<uint32_t*>(reinterpret_cast<bf16_t*>(p.out) + (size_t)(m + p.split_rows) * p.ldo + (n >> 1)) =
                pack_bf2(h0 - bf2f_lo(hi), h1 - bf2f_hi(hi));
    }
}

// RMSNorm is folded WITHOUT a prologue: out = rstd[m] * sum_k (x[m][k] * g[k]) * W[n][k], so the kernel that produces
// the residual row also writes xg = bf16(x * g) for its consumer (RESID epilogue above / the embedding kernels) together
// with deterministic sum-of-squares partials, and the consumer multiplies its fp32 accumulator by
// rstd[m] = rsqrt(sum_p ssq[m][p] / K + eps) in the epilogue ([HF] llama/modeling_llama.py:62-67 with the scalar factor
// moved out of the dot product).  The partials are fetched while the weights stream.
//
// FP8 (W8A16): the packed weights are e4m3 bytes, one 16-byte load per lane = 16 consecutive k of one output row, i.e. a
// 64-wide k "super tile" per wave-instruction ([N/16][K/64][64 lanes][16 B]; lane = n%16 + 16*((k%64)/16)).  The bytes
// are widened to bf16 fragments in registers (exact) and fed to two MFMAs whose activation fragments use the same k
// assignment; the per-output-row power-of-two scale multiplies the fp32 accumulator in the epilogue (exact).

// the tail shared by the per-wave-ring kernels: cross-wave reduction of the K shares through the (finished) ring, the split-mode and
// split-K combinations, the epilogue.  acc[t][q] = this wave's partial sums of tile nt0 + t, row group q.
template <int WAVES, int NT, int MG, int EPI, bool FP8>
VC_DEV void gemv_ring_finish(const GemvArgs& p, f32x4 (&acc)[NT][MG], char* ring, const float* ss_part /*[WAVES][16 * MG]*/, int nt0, int ks, int KS,
                             const bool (&ovalid)[MG]) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m = lane & 15, g = lane >> 4;
    const int ntiles = p.N >> 4;
    const int SR = p.split_rows;
    // RESID epilogue, one unit per wave: the residual values and norm weights the epilogue adds are requested BEFORE the cross-wave
    // reduction (two barriers and the LDS sums then overlap their round trip; round 6).  Unconditional loads at clamped addresses —
    // a guarded load would be waited for at the end of its guard
    GemvResidPre pre{f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}, false};
    if constexpr (EPI == GEMV_RESID_F32 && NT * MG <= WAVES) {
        if (wave < NT * MG && p.xg_w != nullptr) {   // (wave-uniform)
            const int ft = wave % NT, fq = wave / NT;
            const int nt = min(nt0 + ft, ntiles - 1), mrow = ovalid[fq] ? m + 16 * fq : 0;
            const int n = nt * 16 + g * 4;
            pre.o = ld16f(reinterpret_cast<const float*>(p.out) + (size_t)mrow * p.ldo + n);
            pre.gw = ld16f(p.xg_w + n);
            pre.have = true;
        }
    }
    __syncthreads();  // every wave is done with its ring before `red` overwrites it
    float* red = reinterpret_cast<float*>(ring);  // [WAVES][NT][MG][64][4]
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int q = 0; q < MG; ++q) st16f(red + (((wave * NT + t) * MG + q) * 64 + lane) * 4, acc[t][q]);
    __syncthreads();
    // one finishing wave per (tile, row group) unit; with more units than waves (NT * MG > WAVES) a wave takes several
#pragma unroll
    for (int u0 = 0; u0 < (NT * MG + WAVES - 1) / WAVES; ++u0) {
    const int unit_w = u0 * WAVES + wave;
    if (unit_w >= NT * MG) continue;
    const int ft = unit_w % NT, fq = unit_w / NT;  // this wave finishes tile ft for row group fq
    const int nt = nt0 + ft;
    if (nt >= ntiles) continue;
    if (SR == 16 && fq > 0) continue;              // split mode: row group 1 holds the lo parts of group 0's rows
    f32x4 v = ld16f(red + (((0 * NT + ft) * MG + fq) * 64 + lane) * 4);
#pragma unroll
    for (int w = 1; w < WAVES; ++w) v = v + ld16f(red + (((w * NT + ft) * MG + fq) * 64 + lane) * 4);
    if (SR) {  // hi . W + lo . W, each summed over the waves in the fixed order above: the same bits whichever form ran
        if constexpr (MG > 1) {
            if (SR == 16) {
                f32x4 u = ld16f(red + (((0 * NT + ft) * MG + 1) * 64 + lane) * 4);
#pragma unroll
                for (int w = 1; w < WAVES; ++w) u = u + ld16f(red + (((w * NT + ft) * MG + 1) * 64 + lane) * 4);
                v = v + u;
            }
        }
        if (SR == 8) {
            f32x4 u;
#pragma unroll
            for (int e = 0; e < 4; ++e) u[e] = shfl_xor(v[e], 8);   // token slot m + 8 (same output features)
            v = v + u;
        }
    }
    if (KS > 1) {
        // hand the partial to whoever finishes this (tile, row group) last; it adds the KS partials in k order.  Write-through
        // (sc1) stores, drained, then a relaxed agent-scope arrival count; the finisher reads with cache-bypassing (sc1)
        // loads: the "sc1 stores and loads on both sides" form of the CDNA hand-off rules (cdna_hip_programming.md §6 G16 /
        // MI355X_MICROARCH.md "valid forms") — an agent-scope fence here writes back / invalidates the XCD's whole L2 and
        // measured 4-8x slower launches.  The scratch is laid out for two row groups whatever MG is.
        const size_t unit = (size_t)nt * 2 + fq;
        auto entry = [&](int e) { return p.sk_scratch + (((size_t)e * ntiles * 2 + unit) * 64 + lane) * 4; };
        float* mine = entry(ks);
#pragma unroll
        for (int e = 0; e < 4; ++e) st_agent(mine + e, v[e]);
        wait_vmcnt<0>();  // the write-through stores have been acknowledged before the arrival is counted
        wave_lds_fence(); // (a wave issues as one on the hardware; the emulator's lane fibers must all have stored before lane 0 counts)
        unsigned arrived = 0;
        if (lane == 0) arrived = atomic_inc_agent(&p.sk_counters[unit]);
        arrived = shfl(arrived, 0);
        if (arrived != (unsigned)(KS - 1)) continue;
        v = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int k = 0; k < KS; ++k) {
            const float* q = entry(k);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += ld_agent(q + e);
        }
        if (lane == 0) st_agent_u32(&p.sk_counters[unit], 0u);  // re-armed for the next launch (stream order)
    }
    gemv_epilogue<WAVES, EPI, FP8, 16 * MG>(p, v, ss_part, nt, m + 16 * fq, g, ovalid[fq], &pre);
    }
}

// ---- the per-wave-ring kernel (LDS-DMA) ----------------------------------------------------------------------------------
// WAVES waves split K (k-lines interleaved: wave w takes lines w, w + WAVES, ...); a workgroup owns NT consecutive 16-output
// tiles so one activation fragment feeds NT weight tiles.  Nothing of the stream ever sits in VGPRs: every wave owns a private
// ring of R slots in LDS; one slot = the NT packed 1-KiB weight blocks of a k-tile (non-temporal global_load_lds) plus the
// matching activation fragment piece(s) (one 16-byte gather per lane, default cache policy: every workgroup re-reads
// them from L2).  All vector-memory operations of the loop are DMAs issued in program order, OPS per k-tile, so the
// oldest k-tile in flight has landed exactly when vmcnt <= (R-1)*OPS — a counted wait, never a drain; the slot is then
// read back with ds_read_b128 (lane-linear, conflict-free), fed to the MFMAs and re-armed for k-tile i+R.  A wave only
// ever reads LDS bytes that it DMA'd itself, so its own vmcnt wait is the only ordering needed (no barrier in the loop).
// ~40 VGPRs per wave: occupancy is set by the ring size alone.  A pure stream of this shape measures 5.8-6.1 TB/s at the
// qkv / gate-up launch sizes (tools/experiments/corun.hip) against 5.0-5.3 for a register-staged loop (rounds 1-4 kept one as a
// regression variant; removed in round 5).
// XP = 8-row pieces of the activation operand per ring slot (1..4: M <= 8 / 16 / 24 / 32); MG = (XP + 1) / 2 MFMA row groups
// of 16 token rows, each weight block read from LDS feeding MG MFMAs.  The assignment of k-tiles to waves and K-slices does
// not depend on XP (nor on NT / R), so a row's sum is formed in the same order whichever variant serves it: the decode
// pool's 32-row steps give every row bit-for-bit what an 8-row step gives it.
//
// Activation operand: every workgroup re-reads X [M, K] from L2 — M / (16 NT) bytes of it per weight byte, more than the
// weight stream itself from M = 24 on — so it is fetched in FULL 128-byte lines: one DMA instruction = 8 rows x 128 B
// (64 k of bf16 = the k-tile pair of a bf16 slot / the 64-wide super-tile of a W8A16 slot), half the line requests of a
// fragment-shaped gather (16 rows x 64 B per instruction).  The LDS image of a piece is lane-linear [8 rows][8 chunks of
// 16 B]; the chunk a lane fetches is XOR-swizzled by (row >> 1) & 7 on the SOURCE side (the DMA writes linearly), so the
// 16 lanes of a ds_read_b128 group — 16 rows, one chunk column — hit 16 distinct 16-byte bank slots.
template <int WAVES, int NT, int R, int EPI, bool FP8, int XP = 2>
__global__ __launch_bounds__(WAVES * 64) void gemv_dma_kernel(GemvArgs p) {
    constexpr int MG = (XP + 1) / 2;
    constexpr int KSH = FP8 ? 6 : 5;
    // k-tiles per ring slot: a slot always spans 64 k = one 128-byte line of every activation row: a PAIR of 32-wide bf16
    // k-tiles, or one 64-wide W8A16 super-tile
    constexpr int KPI = FP8 ? 1 : 2;
    constexpr int WB = KPI * NT;             // 1-KiB weight blocks per slot
    constexpr int OPS = WB + XP;            // DMA instructions per slot
    constexpr int SLOT = OPS * 1024;
    static_assert(WAVES * R * SLOT >= WAVES * NT * MG * 1024, "the ring is re-used for the cross-wave reduction");
    stamp_begin(p.stamp, blockIdx.x);
    VC_DYNAMIC_SMEM(char, ring);             // [WAVES][R][SLOT]
    __shared__ float ss_part[WAVES][16 * MG];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ntiles = p.N >> 4;
    const int KS = p.ksplit > 1 ? p.ksplit : 1;
    const int ks = (int)blockIdx.x % KS;
    const int nt0 = ((int)blockIdx.x / KS) * NT;
    const int nkt = p.K >> KSH;
    const int nit = (nkt + KPI - 1) / KPI;   // slots' worth of k-tiles in the matrix
    // this workgroup's share of K: a contiguous slice (ksplit)
    const int it0 = (int)((long)ks * nit / KS), it1 = (int)((long)(ks + 1) * nit / KS);
    const int m = lane & 15, g = lane >> 4;
    // precision mode "split" (p.split_rows = G in {8, 16}): X holds G + M rows — rows [0, M) the bf16 hi parts of the M
    // activation rows, rows [G, G + M) their lo parts (x = hi + lo) — and the two partial products of a row meet in the
    // epilogue: token slots m and m + 8 of one MFMA row group (G = 8), or the two row groups (G = 16)
    const int SR = p.split_rows;
    const int Mx = SR ? SR + p.M : p.M;      // rows of X
    bool mvalid[MG], ovalid[MG];             // slot holds a row of X / slot is a row of the OUTPUT (norm partials, stores)
#pragma unroll
    for (int q = 0; q < MG; ++q) {
        mvalid[q] = m + 16 * q < Mx;
        ovalid[q] = m + 16 * q < p.M;
    }
    char* my = ring + wave * (R * SLOT);
    const char* wsrc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
        wsrc[t] = reinterpret_cast<const char*>(p.Wp) + ((size_t)min(nt0 + t, ntiles - 1) * nkt * 64 + lane) * 16;
    // activation pieces: lane = (row r of the piece, slot s of the line); it fetches chunk `xc` of its row
    const int xr = lane >> 3, xs = lane & 7;
    const char* xsrc[XP];
    int xc[XP];
#pragma unroll
    for (int x = 0; x < XP; ++x) {
        const int row = 8 * x + xr;
        const int pc = xs ^ ((row >> 1) & 7);                               // physical chunk held by slot s of this row
        xc[x] = FP8 ? (((pc & 3) << 1) | (pc >> 2)) : pc;                   // W8A16: physical p = logical (c >> 1) | ((c & 1) << 2)
        xsrc[x] = reinterpret_cast<const char*>(p.X + (size_t)(row < Mx ? row : 0) * p.K);
    }
    const int kline_last = (p.K * 2 + 127) / 128 - 1;                        // last (possibly half) 128-byte line of a row
    const bool half_line = (p.K * 2) % 128 != 0;                             // bf16, odd k-tile count: 64 valid bytes in it
    auto issue = [&](int i, int slot) {
        char* dst = my + slot * SLOT;
        const int is = it0 + wave + i * WAVES;                            // slot index along K = line index of X
#pragma unroll
        for (int kk = 0; kk < KPI; ++kk) {
            const size_t kt = (size_t)min(is * KPI + kk, nkt - 1);           // an odd tail re-reads the last tile
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                glds16_nt(wsrc[t] + kt * 1024, dst + (kk * NT + t) * 1024);
            }
        }
#pragma unroll
        for (int x = 0; x < XP; ++x) {
            // the half line at the end of an odd-k-tile row holds only chunks 0..3: the other lanes re-read a valid chunk
            // (their values meet zeroed weights' partner: the consumer zeroes the activation fragment of that k-tile)
            const int c = (half_line && is >= kline_last) ? (xc[x] & 3) : xc[x];
            glds16(xsrc[x] + (size_t)min(is, kline_last) * 128 + c * 16, dst + (WB + x) * 1024);
        }
    };
    f32x4 acc[NT][MG];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int q = 0; q < MG; ++q) acc[t][q] = f32x4{0.f, 0.f, 0.f, 0.f};
    // LDS offset of logical chunk c of row 16q + m inside the slot's activation area
    auto xoff = [&](int q, int c) {
        const int row = 16 * q + m;
        const int pc = FP8 ? ((c >> 1) | ((c & 1) << 2)) : c;
        return (WB + (row >> 3)) * 1024 + ((row & 7) * 8 + (pc ^ ((row >> 1) & 7))) * 16;
    };
    auto consume = [&](int i, int slot) {
        const char* s = my + slot * SLOT;
        const char* sl = s + lane * 16;
        // the activation pieces are read across lanes (a lane reads bytes other lanes of its wave DMA'd): the wave's counted
        // vmcnt wait covers them on the hardware (the wave issues as one); the emulator's lane fibers need the rendezvous
        wave_lds_fence();
#pragma unroll
        for (int kk = 0; kk < KPI; ++kk) {
            u32x4 w[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) w[t] = ld16(sl + (kk * NT + t) * 1024);
            const bool tail = KPI > 1 && (it0 + wave + i * WAVES) * KPI + kk >= nkt;
            u32x4 x0[MG], x1[MG];
#pragma unroll
            for (int q = 0; q < MG; ++q) {
                const bool have = 2 * q + (m >> 3) < XP;                    // the piece holding this row was fetched
                x0[q] = x1[q] = u32x4{0u, 0u, 0u, 0u};
                if (have) {
                    x0[q] = ld16(s + xoff(q, FP8 ? 2 * g : kk * 4 + g));
                    if constexpr (FP8) x1[q] = ld16(s + xoff(q, 2 * g + 1));
                }
                if (!mvalid[q] || tail) x0[q] = x1[q] = u32x4{0u, 0u, 0u, 0u};
            }
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                if constexpr (FP8) {
                    const u32x2 b0 = fp8x4_to_bf16x4(w[t][0]), b1 = fp8x4_to_bf16x4(w[t][1]);
                    const u32x2 b2 = fp8x4_to_bf16x4(w[t][2]), b3 = fp8x4_to_bf16x4(w[t][3]);
#pragma unroll
                    for (int q = 0; q < MG; ++q) {
                        acc[t][q] = mfma16(u32x4{b0[0], b0[1], b1[0], b1[1]}, x0[q], acc[t][q]);
                        acc[t][q] = mfma16(u32x4{b2[0], b2[1], b3[0], b3[1]}, x1[q], acc[t][q]);
                    }
                } else {
#pragma unroll
                    for (int q = 0; q < MG; ++q) acc[t][q] = mfma16(w[t], x0[q], acc[t][q]);
                }
            }
        }
        wait_lgkmcnt<0>();  // the slot's fragment reads have retired before it is re-armed
        wave_lds_fence();
    };
    const int cnt = max(0, (it1 - it0 - wave + WAVES - 1) / WAVES);  // slots of this wave (wave-uniform)
    const int primed = min(cnt, R);
    // 1/rms of the rows from the producer's partials.  The loads are issued AFTER the ring is primed, unconditionally (clamped
    // index, masked where summed) so that they go out back to back, and their round trip overlaps the first weight slots'.
    // (ISA, round 6: rounds 1-5 issued them guarded and BEFORE the ring — the compiler ended every guarded load with its own
    // s_waitcnt vmcnt(0), up to 2 * SQ dependent round trips before the first weight DMA of a launch; and a register load that is
    // outstanding together with LDS-DMAs is always waited for with vmcnt(0) — LLVM treats the mixed queue as unordered — so "before
    // the ring, waited for behind it" cannot be had from C++.  After the ring, that vmcnt(0) only waits for the slots the loop
    // is about to consume anyway.)
    constexpr int SQ = 6;
    const int nq = p.npart >> 2;
    auto ssq_reduce = [&]() {
        if (p.ssq_in == nullptr) return;
        f32x4 sq[MG][SQ];
#pragma unroll
        for (int q = 0; q < MG; ++q) {
            const float* sp = p.ssq_in + (size_t)(ovalid[q] ? m + 16 * q : 0) * p.npart;
#pragma unroll
            for (int j = 0; j < SQ; ++j) sq[q][j] = ld16f(sp + min(wave * 4 + g + j * WAVES * 4, nq - 1) * 4);
        }
#pragma unroll
        for (int q = 0; q < MG; ++q) {
#pragma unroll
            for (int j = 0; j < SQ; ++j)
                if (wave * 4 + g + j * WAVES * 4 >= nq) sq[q][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            float s0 = 0.f, s1 = 0.f;
#pragma unroll
            for (int j = 0; j < SQ; j += 2) {
                s0 += (sq[q][j][0] + sq[q][j][1]) + (sq[q][j][2] + sq[q][j][3]);
                s1 += (sq[q][j + 1][0] + sq[q][j + 1][1]) + (sq[q][j + 1][2] + sq[q][j + 1][3]);
            }
            float ss = s0 + s1;
            const float* sp = p.ssq_in + (size_t)(ovalid[q] ? m + 16 * q : 0) * p.npart;
            for (int qi = wave * 4 + g + SQ * WAVES * 4; qi < nq; qi += WAVES * 4) {  // rows wider than 16*SQ*WAVES*... (rare)
                const f32x4 v = ld16f(sp + qi * 4);
                ss += (v[0] + v[1]) + (v[2] + v[3]);
            }
            ss += shfl_xor(ss, 16);
            ss += shfl_xor(ss, 32);
            if (g == 0) ss_part[wave][16 * q + m] = ss;
        }
    };
    if (cnt >= R) {
#pragma unroll
        for (int r = 0; r < R; ++r) issue(r, r);
        ssq_reduce();
        int slot = 0;
        for (int i = 0; i < cnt - R; ++i) {
            wait_vmcnt<(R - 1) * OPS>();
            consume(i, slot);
            issue(i + R, slot);
            slot = slot + 1 == R ? 0 : slot + 1;
        }
#pragma unroll
        for (int j = 0; j < R; ++j) {  // drain: nothing new is issued, the outstanding count shrinks by OPS per step
            wait_vmcnt_n((R - 1 - j) * OPS);
            consume(cnt - R + j, slot);
            slot = slot + 1 == R ? 0 : slot + 1;
        }
    } else {
        for (int r = 0; r < primed; ++r) issue(r, r);
        ssq_reduce();
        wait_vmcnt<0>();
        for (int r = 0; r < cnt; ++r) consume(r, r);
    }
    gemv_ring_finish<WAVES, NT, MG, EPI, FP8>(p, acc, ring, &ss_part[0][0], nt0, ks, KS, ovalid);
    stamp_end(p.stamp, blockIdx.x);
}

// ---- workgroup-shared activation form ("wg"; bf16 weights, K % 64 == 0) ------------------------------------------------------
// gemv_dma_kernel lets the waves of a workgroup split K, so every workgroup re-reads ALL of X [M, K] from L2: M / (16 NT) bytes
// of it per weight byte — 2.0 at 32 rows, which (not HBM) bounds the pooled decode step, and twice that
// again with the hi / lo rows of precision mode "split".  Here the four waves of a workgroup own different OUTPUT TILES (NTW
// each) over the same K-slice, and the activation rows of that slice come through ONE shared, double-buffered LDS chunk ring
// (CL lines of 64 k per chunk, fetched cooperatively, a quarter per wave): M / (64 NTW) activation bytes per weight byte — 8x
// (NTW = 2) less L2 -> LDS traffic.  The weight stream is unchanged: a private ring of R one-line slots per wave, non-temporal
// LDS-DMA, counted vmcnt waits — the activation DMAs sit in the same in-order queue, so a slot's wait counts the chunk fetch
// issued behind it.  One bare s_barrier per chunk publishes the next chunk (each wave first waits for its own share to land).
//
// K is split over KS workgroups per tile group (the launcher picks KS from the matrix alone); a wave owns its tiles' sums over
// the slice, so there is no cross-wave reduction, only the deterministic cross-workgroup hand-off of the split-K form above
// (sc1 partials, arrival counter, the last arriver adds the slices in k order).  The k order of every sum depends on the matrix
// only — never on M, NTW, CL or R — so a row gets the same bits from an 8-row and a 32-row pass.
//
// KH = 2 (precision mode "split"): X holds the hi rows [0, M) and the lo rows [G, G + M); both planes of a line are staged and
// every weight fragment feeds hi and lo MFMAs of the SAME accumulator (the [hi | lo] K-concatenation of the prefill GEMM) —
// one weight pass for up to 32 rows (64 operand rows), no combine step.
//
// WL (an inexact checkpoint: w = bf16 hi + bf16 lo, GemvArgs::Wp_lo): the slot also carries the lo plane's blocks of the line, and
// every lo fragment feeds one more MFMA against the activation HI plane — x.w = x_hi.w_hi + x_lo.w_hi + x_hi.w_lo, the prefill GEMM's
// third K segment (gemm.hip w_koff); twice the weight bytes per step.
template <int NTW, int XP, int KH, int CL, int R, int EPI, bool WL = false, int WAVES = 4>
__global__ __launch_bounds__(WAVES * 64) void gemv_wg_kernel(GemvArgs p) {
    constexpr int MG = (XP + 1) / 2;          // MFMA row groups of 16 token rows
    constexpr int P = XP * KH;                // 1-KiB activation pieces (8 rows x 128 B) per line
    constexpr int WO = 2 * NTW * (WL ? 2 : 1);   // weight DMA instructions per line (2 k-tiles x NTW tiles [x hi, lo planes])
    constexpr int XO = CL * P / WAVES;        // activation DMA instructions per wave and chunk
    static_assert((CL * P) % WAVES == 0, "the chunk's pieces are dealt evenly to the four waves");
    constexpr int XBUF = CL * P * 1024;       // one activation chunk
    constexpr int SLOT = WO * 1024;
    stamp_begin(p.stamp, blockIdx.x);
    VC_DYNAMIC_SMEM(char, lds);               // [2][XBUF] activation chunks | [WAVES][R][SLOT] weight rings
    __shared__ float ss_part[WAVES][16 * MG];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ntiles = p.N >> 4;
    const int KS = p.ksplit > 1 ? p.ksplit : 1;
    const int ks = (int)blockIdx.x % KS, grp = (int)blockIdx.x / KS;
    const int nlines = p.K >> 6;
    const int l0 = (int)((long)ks * nlines / KS), l1 = (int)((long)(ks + 1) * nlines / KS);
    const int nl = l1 - l0, nchunk = (nl + CL - 1) / CL;
    const int m = lane & 15, g = lane >> 4;
    const int G = p.split_rows;
    char* xb = lds;
    char* my = lds + 2 * XBUF + wave * (R * SLOT);
    const int tile0 = (grp * WAVES + wave) * NTW;
    const char* wsrc[NTW];
    long long lo_delta = 0;   // bytes from a hi block to the lo plane's block of the same (tile, k-tile)
    if constexpr (WL) lo_delta = reinterpret_cast<const char*>(p.Wp_lo) - reinterpret_cast<const char*>(p.Wp);
#pragma unroll
    for (int t = 0; t < NTW; ++t)
        wsrc[t] = reinterpret_cast<const char*>(p.Wp) + ((size_t)min(tile0 + t, ntiles - 1) * (nlines * 2) * 64 + lane) * 16;
    bool ovalid[MG];
#pragma unroll
    for (int q = 0; q < MG; ++q) ovalid[q] = m + 16 * q < p.M;
    // activation pieces: lane = (row xr of the piece, 16-byte slot xs of the line); swizzled on the source side as in gemv_dma_kernel
    const int xr = lane >> 3, xs = lane & 7;
    auto issue_x = [&](int c) {   // this wave's share of chunk c -> buffer c & 1
        char* dst = xb + (c & 1) * XBUF;
#pragma unroll
        for (int o = 0; o < XO; ++o) {
            const int op = o * WAVES + wave;              // piece `pi` of line `j` of the chunk
            const int j = op / P, pi = op % P;
            const int part = pi / XP, x = pi % XP;
            const int row = 8 * x + xr;
            const int pc = xs ^ ((row >> 1) & 7);
            const int line = min(l0 + c * CL + j, nlines - 1);
            const size_t grow = (size_t)(row < p.M ? row : 0) + (part ? (size_t)G : 0);
            glds16(reinterpret_cast<const char*>(p.X) + (grow * p.K + (size_t)line * 64) * 2 + pc * 16, dst + (j * P + pi) * 1024);
        }
    };
    auto issue_w = [&](int i, int slot) {
        char* dst = my + slot * SLOT;
        const size_t line = (size_t)(l0 + i);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int t = 0; t < NTW; ++t) {
                glds16_nt(wsrc[t] + (line * 2 + kk) * 1024, dst + (kk * NTW + t) * 1024);
                if constexpr (WL) glds16_nt(wsrc[t] + lo_delta + (line * 2 + kk) * 1024, dst + ((2 + kk) * NTW + t) * 1024);
            }
    };
    f32x4 acc[NTW][MG];
#pragma unroll
    for (int t = 0; t < NTW; ++t)
#pragma unroll
        for (int q = 0; q < MG; ++q) acc[t][q] = f32x4{0.f, 0.f, 0.f, 0.f};
    // LDS offset of logical chunk c16 (16 bytes = 8 k) of local row `row` inside piece area `part` of line j
    auto xoff = [&](int j, int part, int row, int c16) {
        return (j * P + part * XP + (row >> 3)) * 1024 + ((row & 7) * 8 + (c16 ^ ((row >> 1) & 7))) * 16;
    };
    auto consume = [&](int c, int j, int slot) {
        const char* xs_ = xb + (c & 1) * XBUF;
        const char* sl = my + slot * SLOT + lane * 16;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            u32x4 w[NTW], wl[WL ? NTW : 1];
#pragma unroll
            for (int t = 0; t < NTW; ++t) {
                w[t] = ld16(sl + (kk * NTW + t) * 1024);
                if constexpr (WL) wl[t] = ld16(sl + ((2 + kk) * NTW + t) * 1024);
            }
#pragma unroll
            for (int h = 0; h < KH; ++h) {
                u32x4 x[MG];
#pragma unroll
                for (int q = 0; q < MG; ++q) {
                    const int row = 16 * q + m;
                    x[q] = u32x4{0u, 0u, 0u, 0u};
                    if (2 * q + (m >> 3) < XP) x[q] = ld16(xs_ + xoff(j, h, row, kk * 4 + g));
                    if (row >= p.M) x[q] = u32x4{0u, 0u, 0u, 0u};
                }
#pragma unroll
                for (int t = 0; t < NTW; ++t)
#pragma unroll
                    for (int q = 0; q < MG; ++q) {
                        acc[t][q] = mfma16(w[t], x[q], acc[t][q]);
                        if constexpr (WL)
                            if (h == 0) acc[t][q] = mfma16(wl[t], x[q], acc[t][q]);   // the weight's lo plane against the activation's hi plane
                    }
            }
        }
        wait_lgkmcnt<0>();  // the slot's fragment reads have retired before it is re-armed
    };
    // 1/rms partials: loaded after the ring is primed, unconditionally (gemv_dma_kernel says why)
    constexpr int SQ = 6;
    const int nq = p.npart >> 2;
    const int primed = min(nl, R);
    if (nl > 0) issue_x(0);
    for (int r = 0; r < primed; ++r) issue_w(r, r);
    if (p.ssq_in != nullptr) {
        f32x4 sq[MG][SQ];
#pragma unroll
        for (int q = 0; q < MG; ++q) {
            const float* sp = p.ssq_in + (size_t)(ovalid[q] ? m + 16 * q : 0) * p.npart;
#pragma unroll
            for (int j = 0; j < SQ; ++j) sq[q][j] = ld16f(sp + min(wave * 4 + g + j * WAVES * 4, nq - 1) * 4);
        }
#pragma unroll
        for (int q = 0; q < MG; ++q)
#pragma unroll
            for (int j = 0; j < SQ; ++j)
                if (wave * 4 + g + j * WAVES * 4 >= nq) sq[q][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < MG; ++q) {
            float s0 = 0.f, s1 = 0.f;
#pragma unroll
            for (int j = 0; j < SQ; j += 2) {
                s0 += (sq[q][j][0] + sq[q][j][1]) + (sq[q][j][2] + sq[q][j][3]);
                s1 += (sq[q][j + 1][0] + sq[q][j + 1][1]) + (sq[q][j + 1][2] + sq[q][j + 1][3]);
            }
            float ss = s0 + s1;
            const float* sp = p.ssq_in + (size_t)(ovalid[q] ? m + 16 * q : 0) * p.npart;
            for (int qi = wave * 4 + g + SQ * WAVES * 4; qi < nq; qi += WAVES * 4) {
                const f32x4 v = ld16f(sp + qi * 4);
                ss += (v[0] + v[1]) + (v[2] + v[3]);
            }
            ss += shfl_xor(ss, 16);
            ss += shfl_xor(ss, 32);
            if (g == 0) ss_part[wave][16 * q + m] = ss;
        }
    }
    // chunk 0 of every wave has landed (everything issued behind it: the primed weight slots) and is published
    wait_vmcnt_n(primed * WO);
    wg_barrier_raw();
    int slot = 0;
    for (int c = 0; c < nchunk; ++c) {
        const bool more = c + 1 < nchunk;
        if (more) issue_x(c + 1);   // into the buffer whose last reads (chunk c - 1) every wave finished before the last barrier
        int rearmed = 0;
#pragma unroll
        for (int j = 0; j < CL; ++j) {
            const int i = c * CL + j;
            if (i < nl) {
                // DMAs issued behind slot i: the slots i+1 .. min(i+R, nl)-1, and chunk c+1's share when slot i went out before it
                const int behind = (min(i + R, nl) - i - 1) * WO + ((more && j < R) ? XO : 0);
                if (i + R <= nl) {
                    if (more && j < R) wait_vmcnt<(R - 1) * WO + XO>();
                    else wait_vmcnt<(R - 1) * WO>();
                } else {
                    wait_vmcnt_n(behind);
                }
                consume(c, j, slot);
                if (i + R < nl) {
                    issue_w(i + R, slot);
                    ++rearmed;
                }
                slot = slot + 1 == R ? 0 : slot + 1;
            }
        }
        if (more) {
            wait_vmcnt_n(rearmed * WO);   // this wave's share of chunk c+1 has landed (only the re-armed slots are younger)
            wg_barrier_raw();             // ... and so has everyone's; nobody still reads chunk c's buffer's predecessor
        }
    }
    __syncthreads();   // ss_part is complete (and, with KS > 1, nothing of the rings is live any more)
#pragma unroll
    for (int t = 0; t < NTW; ++t) {
        const int nt = tile0 + t;
        if (nt >= ntiles) continue;
#pragma unroll
        for (int q = 0; q < MG; ++q) {
            f32x4 v = acc[t][q];
            if (KS > 1) {
                // the split-K hand-off of gemv_dma_kernel: write-through partial, drained, arrival count; the last arriver adds the
                // KS partials in k order with cache-bypassing loads and re-arms the counter
                const size_t unit = (size_t)nt * 2 + q;
                float* mine = p.sk_scratch + (((size_t)ks * ntiles * 2 + unit) * 64 + lane) * 4;
#pragma unroll
                for (int e = 0; e < 4; ++e) st_agent(mine + e, v[e]);
                wait_vmcnt<0>();
                wave_lds_fence();   // (the emulator's lane fibers must all have stored before lane 0 counts the arrival)
                unsigned arrived = 0;
                if (lane == 0) arrived = atomic_inc_agent(&p.sk_counters[unit]);
                arrived = shfl(arrived, 0);
                if (arrived != (unsigned)(KS - 1)) continue;
                v = f32x4{0.f, 0.f, 0.f, 0.f};
                for (int k = 0; k < KS; ++k) {
                    const float* qp = p.sk_scratch + (((size_t)k * ntiles * 2 + unit) * 64 + lane) * 4;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] += ld_agent(qp + e);
                }
                if (lane == 0) st_agent_u32(&p.sk_counters[unit], 0u);
            }
            gemv_epilogue<WAVES, EPI, false, 16 * MG>(p, v, &ss_part[0][0], nt, m + 16 * q, g, ovalid[q]);
        }
    }
    stamp_end(p.stamp, blockIdx.x);
}

template <class K>
static void allow_big_lds(K kernel, size_t bytes) {
#ifndef VC_EMU
    if (bytes > 48 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
#endif
}

template <int WAVES, int NT, int R, bool FP8, int XP>
static void launch_gemv_dma_x(const GemvArgs& a, int epi, hipStream_t s) {
    const dim3 grid(((a.N / 16 + NT - 1) / NT) * (a.ksplit > 1 ? a.ksplit : 1)), block(WAVES * 64);
    constexpr size_t shmem = (size_t)WAVES * R * ((FP8 ? 1 : 2) * NT + XP) * 1024;
    // + the kernel's static ss_part[WAVES][16 * MG] floats
    static_assert(shmem + WAVES * 16 * ((XP + 1) / 2) * 4 <= 160 * 1024, "ring exceeds the LDS of a CU");
#define VC_GEMV_DMA(E)                                                                                                  \
    do {                                                                                                                \
        static bool once = false;                                                                                       \
        if (!once) {                                                                                                    \
            allow_big_lds(gemv_dma_kernel<WAVES, NT, R, E, FP8, XP>, shmem);                                            \
            once = true;                                                                                                \
        }                                                                                                               \
        VC_LAUNCH((gemv_dma_kernel<WAVES, NT, R, E, FP8, XP>), grid, block, shmem, s, a);                               \
    } while (0)
    switch (epi) {
        case GEMV_BF16: VC_GEMV_DMA(GEMV_BF16); break;
        case GEMV_F32: VC_GEMV_DMA(GEMV_F32); break;
        case GEMV_RESID_F32: VC_GEMV_DMA(GEMV_RESID_F32); break;
        default: VC_GEMV_DMA(GEMV_SWIGLU); break;
    }
#undef VC_GEMV_DMA
}

// rows of the activation operand X: M, or split_rows + M in precision mode "split" (hi rows, then lo rows)
static int x_rows(const GemvArgs& a) { return a.split_rows ? a.split_rows + a.M : a.M; }

// X rows <= 16: one MFMA row group, one (<= 8 rows) or two 8-row activation pieces per slot
template <int WAVES, int NT, int R, bool FP8>
static void launch_gemv_dma(const GemvArgs& a, int epi, hipStream_t s) {
    if (x_rows(a) <= 8) launch_gemv_dma_x<WAVES, NT, R, FP8, 1>(a, epi, s);
    else launch_gemv_dma_x<WAVES, NT, R, FP8, 2>(a, epi, s);
}
// X rows in 17..32: two row groups, three or four pieces
template <int WAVES, int NT, int R, bool FP8>
static void launch_gemv_dma2(const GemvArgs& a, int epi, hipStream_t s) {
    if (x_rows(a) <= 24) launch_gemv_dma_x<WAVES, NT, R, FP8, 3>(a, epi, s);
    else launch_gemv_dma_x<WAVES, NT, R, FP8, 4>(a, epi, s);
}

// K-slices of the 257..512-tile class (13b o_proj / down: 320 tiles neither fill the chip one-per-CU nor balance two-per-CU):
// with the split-K buffers, pairs of tiles x 3 K-slices = 480 workgroups, all resident (2/CU), activation fragments shared by
// the pair (measured M = 16: down 43.0 -> 30.2 us, o_proj 20.1 -> 14.9; for <= 256 tiles every split loses: 8.5 -> 10 us).
// An explicit a.ksplit is honoured.
constexpr int KS_MID = 3;

// M in 17..32 (the decode pool): two row groups per weight pass.  The K partition (waves per workgroup, K-slices) of every
// tile-count class equals the 16-row launcher's below, so a row's result does not depend on which variant served it; only
// the tiles per workgroup and the ring depth are re-balanced for the larger slots (LDS: all workgroups resident).
template <bool FP8>
static void launch_gemv_m32(const GemvArgs& a, int epilogue, hipStream_t s) {
    const int tiles = a.N / 16;
    if (a.ksplit > 1 || (a.sk_scratch && a.sk_counters && a.ksplit == 0 && tiles > 256 && tiles <= 512)) {
        GemvArgs b = a;
        if (b.ksplit <= 1) b.ksplit = KS_MID;
        if (a.ksplit > 1 && tiles <= 256) launch_gemv_dma2<4, 1, 2, FP8>(b, epilogue, s);
        else launch_gemv_dma2<4, 2, 2, FP8>(b, epilogue, s);
        return;
    }
    if (tiles <= 256) {
        // W8A16 slots are 4 KiB (3 pieces) / 5 KiB (4 pieces): a 4-slot ring of the latter would need all 160 KiB + ss_part
        if constexpr (FP8) {
            if (x_rows(a) <= 24) {
                launch_gemv_dma_x<8, 1, 4, true, 3>(a, epilogue, s);
                return;
            }
        }
        launch_gemv_dma2<8, 1, 3, FP8>(a, epilogue, s);
    } else if (tiles <= 512) launch_gemv_dma2<4, 1, 3, FP8>(a, epilogue, s);
    else {
        // W8A16, the widest matrices (13b gate/up: 1728 tiles): 4 tiles per workgroup halve the activation bytes per weight
        // byte — the bound of this kernel at 17..32 rows, DESIGN.md section 9.2 — same K partition (4 waves).  Measured at 32
        // rows: 45.2 -> 35.3 us; with 1376 tiles (7b gate/up) the 344 single-resident workgroups balance badly over 256 CUs
        // and it loses (26.2 -> 27.9 us), hence the threshold.
        if constexpr (FP8) {
            if (tiles >= 1536) {
                launch_gemv_dma2<4, 4, 2, true>(a, epilogue, s);
                return;
            }
        }
        // 513..768 tiles as pairs too (2 workgroups/CU, the pair shares the activation pieces; 28.3 vs 33.9 us at M = 24 for
        // one tile per workgroup)
        launch_gemv_dma2<4, 2, 2, FP8>(a, epilogue, s);
    }
}

template <bool FP8>
static void launch_gemv_f(const GemvArgs& a, int epilogue, hipStream_t s) {
    const int tiles = a.N / 16;
    if (x_rows(a) > 16) {  // the decode pool's 17..32 rows (split mode, ring form: 9..16 rows + their lo parts)
        launch_gemv_m32<FP8>(a, epilogue, s);
        return;
    }
    // Geometry by tile count, so that (where possible) every workgroup of the launch is resident at once — a tail of
    // late workgroups cannot keep enough bytes in flight to use the HBM (13b o_proj/down: 320 tiles at one
    // 128-KiB workgroup per CU ran a 64-workgroup second round at a third of the rate):
    //   <= 256 tiles: 8 waves, deep ring (1 workgroup/CU);  <= 512: 4 waves, 5-slot ring (2 workgroups/CU);
    //   <= 768: 4 waves x 1 tile (3/CU);  wider: 4 waves x 2 tiles (pairs share the activation fragments:
    //   gate/up 30.0 vs 33.1 us, lm_head 40.6 vs 45.5; 4 tiles per workgroup measured slower: W8A16 gate/up 23.2 vs
    //   19.7 us).  bf16 slots hold a k-tile pair, W8A16 slots one super-tile.
    if (a.ksplit > 1 || (a.sk_scratch && a.sk_counters && a.ksplit == 0 && tiles > 256 && tiles <= 512)) {
        GemvArgs b = a;
        if (b.ksplit <= 1) b.ksplit = KS_MID;
        if (a.ksplit > 1 && tiles <= 256) launch_gemv_dma<4, 1, 3, FP8>(b, epilogue, s);
        else launch_gemv_dma<4, 2, 3, FP8>(b, epilogue, s);
        return;
    }
    if (tiles <= 256) launch_gemv_dma<8, 1, FP8 ? 5 : 4, FP8>(a, epilogue, s);
    else if (tiles <= 512) launch_gemv_dma<4, 1, 5, FP8>(a, epilogue, s);
    else if (tiles <= 768 && !FP8) launch_gemv_dma<4, 1, 3, FP8>(a, epilogue, s);  // W8A16: always pairs (X is 2x W)
    else launch_gemv_dma<4, 2, 3, FP8>(a, epilogue, s);
}

// ---- launcher of the workgroup-shared form (precision mode "split" only) -----------------------------------------------------
// K-slices per tile group: a function of the MATRIX alone (so that every row count sums in the same order).  Measured on MI355X at
// the 7b shapes, 8 and 32 rows (profiles/r04_b_kbench_gemv_wg.txt): one tile per wave beats tile pairs everywhere (twice the
// workgroups), and every extra K-slice costs (the cross-workgroup hand-off) — so only as many slices as it takes to put >= 256
// workgroups on the chip: qkv (768 tiles) 2, o / down (256) 4, gate-up / lm_head 1.
static int wg_kslices(int ntiles, int K) {
    const int groups = (ntiles + 3) / 4, lines = K / 64;
    int ks = groups >= 256 ? 1 : groups >= 128 ? 2 : 4;
    while (ks > 1 && lines / ks < 8) --ks;
    return ks;
}

template <int XP, int CL, int R, bool WL, int WAVES = 4>
static void launch_gemv_wg_e(const GemvArgs& a, int epi, hipStream_t s) {
    constexpr int NTW = 1, KH = 2;
    const int groups = (a.N / 16 + WAVES * NTW - 1) / (WAVES * NTW);
    const dim3 grid((unsigned)(groups * (a.ksplit > 1 ? a.ksplit : 1))), block(WAVES * 64);
    constexpr size_t shmem = (size_t)(2 * CL * XP * KH + WAVES * R * 2 * NTW * (WL ? 2 : 1)) * 1024;
    static_assert(shmem + WAVES * 16 * ((XP + 1) / 2) * 4 <= 160 * 1024, "exceeds the LDS of a CU");
#define VC_GEMV_WG(E)                                                                                                   \
    do {                                                                                                                \
        static bool once = false;                                                                                       \
        if (!once) {                                                                                                    \
            allow_big_lds(gemv_wg_kernel<NTW, XP, KH, CL, R, E, WL, WAVES>, shmem);                                     \
            once = true;                                                                                                \
        }                                                                                                               \
        VC_LAUNCH((gemv_wg_kernel<NTW, XP, KH, CL, R, E, WL, WAVES>), grid, block, shmem, s, a);                        \
    } while (0)
    switch (epi) {
        case GEMV_BF16: VC_GEMV_WG(GEMV_BF16); break;
        case GEMV_F32: VC_GEMV_WG(GEMV_F32); break;
        case GEMV_RESID_F32: VC_GEMV_WG(GEMV_RESID_F32); break;
        default: VC_GEMV_WG(GEMV_SWIGLU); break;
    }
#undef VC_GEMV_WG
}

static int g_gemv_variant = -1;
static std::atomic<unsigned long> g_gemv_wg_launches{0};   // sessions launch from their own threads
void set_gemv_variant(int v) { g_gemv_variant = v; }
unsigned long gemv_wg_launches() { return g_gemv_wg_launches.load(std::memory_order_relaxed); }
// set_gemv_variant: 0 = the per-wave-ring kernel for everything (a split step then takes two weight passes of 16 rows); -1 / 1
// (default) = the workgroup-shared form for precision mode "split".  Measured (profiles/r04_b_kbench_gemv_wg.txt): for the bf16
// step the ring kernel is faster at every row count (3.13 vs 3.57 ms at 32 rows with the best geometry found — fewer, larger DMA
// slots in flight per wave and no barrier in its loop outweigh the activation traffic it repeats), so the bf16 form of the
// workgroup-shared kernel was removed in round 5; for the split step one weight pass over 32 rows' hi + lo planes beats the ring
// kernel's two passes of 16 (6.27 ms).
bool gemv_wg_enabled() { return g_gemv_variant != 0; }

// true when the workgroup-shared form can serve the call
bool gemv_wg_applies(int K, bool fp8_weights) { return !fp8_weights && K % 64 == 0; }
static bool gemv_wg_applies(const GemvArgs& a) {
    if (!a.split_rows || !gemv_wg_applies(a.K, a.wscale != nullptr) || a.M < 1 || a.M > 32) return false;
    return a.split_rows >= a.M && a.split_rows <= 32 && a.split_rows % 8 == 0;
}

// Waves (= tiles) per workgroup: 4, or 6 where that lowers the tile count of the BUSIEST CU — round 6: with four tiles per workgroup
// 7b gate / up is 344 workgroups (88 CUs hold two, 168 one: the launch lasts 8 tiles on a CU for 5.4 on average) and qkv 192 x 2
// K-slices = 384 (half the CUs hold two); with six they are 230 and 256 workgroups, one per CU.  A function of the matrix alone
// (the K-slices are wg_kslices's, whatever the choice: the sums keep their order; only the rstd partials are added per wave).
// set_gemv_variant(2) keeps 4 everywhere (A/B), 3 takes 6 everywhere (tests).
static int wg_waves(int ntiles, int ks) {
    if (g_gemv_variant == 2) return 4;
    if (g_gemv_variant == 3) return 6;   // (tests: the six-wave geometry on matrices too small to choose it)
    auto busiest = [&](int w) { return (((ntiles + w - 1) / w) * ks + 255) / 256 * w; };   // tiles (x 1 / ks of K) of the busiest CU
    return busiest(6) < busiest(4) ? 6 : 4;
}

static void launch_gemv_wg(const GemvArgs& a0, int epi, hipStream_t s) {
    GemvArgs a = a0;
    const int ntiles = a.N / 16;
    int ks = a.ksplit > 1 ? a.ksplit : wg_kslices(ntiles, a.K);   // an explicit request (tests) is honoured
    const size_t cap = a.sk_scratch_floats ? a.sk_scratch_floats : (size_t)4 * 512 * 2 * 256;
    const int ncnt = a.sk_counters_n ? a.sk_counters_n : 512 * 2;
    // the split-K buffers bound the slices (a geometry decision of the matrix: the buffers are sized once per model)
    while (ks > 1 && (!a.sk_scratch || !a.sk_counters || (size_t)ks * ntiles * 2 * 256 > cap || ntiles * 2 > ncnt)) --ks;
    a.ksplit = ks;
    g_gemv_wg_launches.fetch_add(1, std::memory_order_relaxed);
    // chunk length / ring depth by the activation pieces per line, so that two workgroups fit a CU (<= 72 KiB each); with the
    // weight's lo plane (an inexact checkpoint) the slots are twice as large: one workgroup per CU (80 - 96 KiB)
    const int xp = (a.M + 7) / 8;
    if (wg_waves(ntiles, ks) == 6) {
        // six tiles per workgroup where that spreads the launch evenly (7b qkv: 128 groups x 2 K-slices = 256 workgroups; gate / up:
        // 230; 13b o / down: 54 x 4 = 216): ONE workgroup per CU, so the ring can be deeper (R = 6; 4 with the lo plane's blocks)
        if (a.Wp_lo != nullptr) {
            switch (xp) {
                case 1: launch_gemv_wg_e<1, 3, 4, true, 6>(a, epi, s); break;
                case 2: launch_gemv_wg_e<2, 3, 4, true, 6>(a, epi, s); break;
                case 3: launch_gemv_wg_e<3, 2, 4, true, 6>(a, epi, s); break;
                default: launch_gemv_wg_e<4, 3, 4, true, 6>(a, epi, s); break;
            }
            return;
        }
        switch (xp) {
            case 1: launch_gemv_wg_e<1, 3, 6, false, 6>(a, epi, s); break;
            case 2: launch_gemv_wg_e<2, 3, 6, false, 6>(a, epi, s); break;
            case 3: launch_gemv_wg_e<3, 2, 6, false, 6>(a, epi, s); break;
            default: launch_gemv_wg_e<4, 3, 6, false, 6>(a, epi, s); break;
        }
        return;
    }
    if (a.Wp_lo != nullptr) {
        switch (xp) {
            case 1: launch_gemv_wg_e<1, 4, 4, true>(a, epi, s); break;
            case 2: launch_gemv_wg_e<2, 4, 4, true>(a, epi, s); break;
            case 3: launch_gemv_wg_e<3, 2, 4, true>(a, epi, s); break;
            default: launch_gemv_wg_e<4, 2, 4, true>(a, epi, s); break;
        }
        return;
    }
    switch (xp) {
        case 1: launch_gemv_wg_e<1, 4, 4, false>(a, epi, s); break;
        case 2: launch_gemv_wg_e<2, 4, 4, false>(a, epi, s); break;
        case 3: launch_gemv_wg_e<3, 2, 4, false>(a, epi, s); break;
        default: launch_gemv_wg_e<4, 2, 4, false>(a, epi, s); break;
    }
}

// ---- "wide" geometry of the ring kernel (set_gemv_wide: -1 / 1 = the measured classes below, 0 = off, 2 = every class) ----------
// ONE deep-ringed workgroup per CU, all 256 CUs holding exactly one, beats every geometry with more, unevenly spread workgroups
// (profiles/r04_q_kbench_gemv_nt3.txt: 7b qkv at 32 rows as 256 triples with a 3-slot ring, 72 KiB of weights in flight on every
// CU, 25.0 -> 20.2 us against 384 pair-workgroups sitting two on half the CUs and one on the other half).  For the matrices with
// more than 512 tiles: NT = ceil(tiles / 256) tiles per 4-wave workgroup (>= 218 workgroups, i.e. >= 85 % of the CUs), the ring as
// deep as 160 KiB allow.  The K partition stays 4 waves without K-slices, so the bits are those of the default geometry (checked on
// the device for every class and row count).  Measured on MI355X (profiles/r05_a_kbench_gemv_wide.txt, us, default -> wide at
// 8 / 16 / 24 / 32 rows), which is what the default table below encodes:
//   NT 3: 7b qkv (768 tiles -> 256 workgroups)      18.7 -> 18.2 | 20.2 -> 18.7 | 23.4 -> 20.3 | 25.0 -> 20.2   => always
//   NT 4: 13b qkv (960 -> 240)                      26.0 -> 26.5 | 27.7 -> 26.7 | 30.5 -> 28.6 | 32.6 -> 29.2   => from 9 rows on
//   NT 6: 7b gate/up (1376 -> 230)                  30.6 -> 30.9 | 31.4 -> 30.9 | 33.0 -> 33.0 | 36.7 -> 33.4   => from 9 rows on
//   NT 7: 13b gate/up (1728 -> 247)                 47.1 -> 46.9 | 50.4 -> 46.8 | 56.5 -> 48.9 | 61.4 -> 49.5   => always
// (NT 8, lm_head 2000 -> 250, lost at every row count — 41.9 -> 47.5 us at 8 rows — and was removed.)
static int g_gemv_wide = -1;
static std::atomic<unsigned long> g_gemv_wide_launches{0};   // sessions launch from their own threads
void set_gemv_wide(int v) { g_gemv_wide = v; }
unsigned long gemv_wide_launches() { return g_gemv_wide_launches.load(std::memory_order_relaxed); }
template <int NT, int R, int XP, int EPI>
static void launch_gemv_wide1(const GemvArgs& a, hipStream_t s) {
    constexpr int WAVES = 4;
    const dim3 grid((a.N / 16 + NT - 1) / NT), block(WAVES * 64);
    constexpr size_t shmem = (size_t)WAVES * R * (2 * NT + XP) * 1024;
    static_assert(shmem + WAVES * 16 * ((XP + 1) / 2) * 4 <= 160 * 1024, "ring exceeds the LDS of a CU");
    static bool once = false;
    if (!once) {
        allow_big_lds(gemv_dma_kernel<WAVES, NT, R, EPI, false, XP>, shmem);
        once = true;
    }
    VC_LAUNCH((gemv_dma_kernel<WAVES, NT, R, EPI, false, XP>), grid, block, shmem, s, a);
}
// W8A16 (round 6): the same idea for the e4m3 weights at the pool's 17..32 rows.  A slot there is NT KiB of weights + 3-4 KiB of
// activation pieces, so the pair geometry (NT = 2, a 2-slot ring, 2-3 workgroups per CU) keeps ~30 KiB of weights in flight per CU
// — qkv 3.2, gate / up 4.1 TB/s at 13b (profiles/r06_ad_kernel_stats_13b_fp8.md); ceil(tiles / 256) tiles per workgroup put one
// workgroup on every CU with a ring as deep as the LDS allows (48-84 KiB of weights in flight) and read the activation rows
// NT / 2 times less often.  Four waves, no K-slices: the pair geometry's K partition, the same bits.  From 9 rows on (measured at 12 /
// 16 / 24 / 32 rows, profiles/r06_af_kbench_gemv_rows8.txt: 13b gate / up 31.7 -> 27.0 us at 16 rows, 33.2 -> 29.8 at 32; 13b qkv 17.4
// -> 16.6 and 23.9 -> 19.9; the one loss is 13b qkv at 12 rows, 16.4 -> 16.7); 8 rows and below keep pairs (slots of NT + 1 KiB).
template <int NT, int R, int XP, int EPI>
static void launch_gemv_wide8_1(const GemvArgs& a, hipStream_t s) {
    constexpr int WAVES = 4;
    const dim3 grid((a.N / 16 + NT - 1) / NT), block(WAVES * 64);
    constexpr size_t shmem = (size_t)WAVES * R * (NT + XP) * 1024;
    static_assert(shmem + WAVES * 16 * ((XP + 1) / 2) * 4 <= 160 * 1024, "ring exceeds the LDS of a CU");
    static bool once = false;
    if (!once) {
        allow_big_lds(gemv_dma_kernel<WAVES, NT, R, EPI, true, XP>, shmem);
        once = true;
    }
    VC_LAUNCH((gemv_dma_kernel<WAVES, NT, R, EPI, true, XP>), grid, block, shmem, s, a);
}
static bool launch_gemv_wide8(const GemvArgs& a, int epi, hipStream_t s) {
    const int tiles = a.N / 16, xr = x_rows(a);
    if (tiles <= 512 || xr <= 8 || a.K % 64 != 0) return false;
    const int nt = (tiles + 255) / 256;
    if ((tiles + nt - 1) / nt < 218) return false;
    const int xp = xr <= 16 ? 2 : xr <= 24 ? 3 : 4;
#define VC_WIDE8(NT_, R_, XP_, E_)                                         \
    if (nt == NT_ && xp == XP_ && epi == E_) {                             \
        launch_gemv_wide8_1<NT_, R_, XP_, E_>(a, s);                       \
        g_gemv_wide_launches.fetch_add(1, std::memory_order_relaxed);      \
        return true;                                                       \
    }
    VC_WIDE8(3, 7, 2, GEMV_BF16) VC_WIDE8(4, 6, 2, GEMV_BF16) VC_WIDE8(6, 4, 2, GEMV_SWIGLU) VC_WIDE8(7, 4, 2, GEMV_SWIGLU)
    VC_WIDE8(3, 6, 3, GEMV_BF16) VC_WIDE8(3, 5, 4, GEMV_BF16) VC_WIDE8(4, 5, 3, GEMV_BF16) VC_WIDE8(4, 4, 4, GEMV_BF16)
    VC_WIDE8(6, 4, 3, GEMV_SWIGLU) VC_WIDE8(6, 3, 4, GEMV_SWIGLU) VC_WIDE8(7, 3, 3, GEMV_SWIGLU) VC_WIDE8(7, 3, 4, GEMV_SWIGLU)
#undef VC_WIDE8
    return false;
}
static bool launch_gemv_wide(const GemvArgs& a, int epi, hipStream_t s) {
    const int wide = g_gemv_wide < 0 ? 1 : g_gemv_wide;
    if (!wide || a.ksplit > 1 || a.split_rows) return false;
    if (a.wscale) return launch_gemv_wide8(a, epi, s);
    const int tiles = a.N / 16;
    if (tiles <= 512) return false;
    const int nt = (tiles + 255) / 256;
    if (nt > 7 || (tiles + nt - 1) / nt < 218) return false;
    const int xr = x_rows(a), xp = xr <= 8 ? 1 : xr <= 16 ? 2 : xr <= 24 ? 3 : 4;
    if (wide == 1 && xp == 1 && (nt == 4 || nt == 6)) return false;   // the two classes that lose at <= 8 rows (table above)
#define VC_WIDE(NT_, R_, XP_, E_)                                          \
    if (nt == NT_ && xp == XP_ && epi == E_) {                             \
        launch_gemv_wide1<NT_, R_, XP_, E_>(a, s);                         \
        g_gemv_wide_launches.fetch_add(1, std::memory_order_relaxed);      \
        return true;                                                       \
    }
    VC_WIDE(3, 5, 1, GEMV_BF16) VC_WIDE(3, 4, 2, GEMV_BF16) VC_WIDE(3, 3, 3, GEMV_BF16) VC_WIDE(3, 3, 4, GEMV_BF16)
    VC_WIDE(4, 4, 1, GEMV_BF16) VC_WIDE(4, 3, 2, GEMV_BF16) VC_WIDE(4, 3, 3, GEMV_BF16) VC_WIDE(4, 3, 4, GEMV_BF16)
    VC_WIDE(6, 3, 1, GEMV_SWIGLU) VC_WIDE(6, 2, 2, GEMV_SWIGLU) VC_WIDE(6, 2, 3, GEMV_SWIGLU) VC_WIDE(6, 2, 4, GEMV_SWIGLU)
    VC_WIDE(7, 2, 1, GEMV_SWIGLU) VC_WIDE(7, 2, 2, GEMV_SWIGLU) VC_WIDE(7, 2, 3, GEMV_SWIGLU) VC_WIDE(7, 2, 4, GEMV_SWIGLU)
#undef VC_WIDE
    return false;
}

void launch_gemv(const GemvArgs& a, int epilogue, hipStream_t s) {
    if (gemv_wg_enabled() && gemv_wg_applies(a)) {
        launch_gemv_wg(a, epilogue, s);
        return;
    }
    if (a.Wp_lo != nullptr)
        throw std::runtime_error("decode GEMV: the weight lo plane of an inexact checkpoint is served by the workgroup-shared split form only "
                                 "(precision mode split, bf16 weights, K % 64 == 0, set_gemv_variant != 0)");
    if (a.split_rows) {
        // hi rows [0, M) + lo rows [G, G + M) of X; the two MFMA forms that combine them: G = 8 inside one 16-slot row group
        // (M <= 8), G = 16 across the two row groups (M <= 16).  No split-K hand-off in this mode.
        if (!((a.split_rows == 8 && a.M <= 8) || (a.split_rows == 16 && a.M <= 16)))
            throw std::runtime_error("split GEMV: rows per pass must fit the group (G = 8: M <= 8, G = 16: M <= 16)");
        GemvArgs b = a;
        b.sk_scratch = nullptr;
        b.sk_counters = nullptr;
        b.ksplit = 0;
        if (b.wscale) launch_gemv_f<true>(b, epilogue, s);
        else launch_gemv_f<false>(b, epilogue, s);
        return;
    }
    if (launch_gemv_wide(a, epilogue, s)) return;
    if (a.wscale) launch_gemv_f<true>(a, epilogue, s);
    else launch_gemv_f<false>(a, epilogue, s);
}

// ---- W8A16 quantiser (load time).  One workgroup per output row n of W [N, K] bf16:
//   s_n = 2^e, e = the smallest integer with absmax_n <= 448 * 2^e     (power of two -> W/s and q*s are exact)
//   q   = e4m3(W / s_n)  (RNE)  -> packed super-tile layout for gemv_kernel<FP8>
//   W  <- bf16(q * s_n)  IN PLACE, so the prefill GEMMs (bf16 row-major) see exactly the weights decode sees.
//   Wrow (optional): the same bytes row-major [N, K] — the weight operand of the e4m3 x e4m3 prefill GEMM.
__global__ __launch_bounds__(256) void quantize_fp8_kernel(bf16_t* W, uint8_t* Wq, float* scale, int N, int K, uint8_t* Wrow) {
    __shared__ float red[4];
    const int n = blockIdx.x, tid = threadIdx.x;
    bf16_t* row = W + (size_t)n * K;
    float amax = 0.f;
    for (int c = tid; c < (K >> 3); c += 256) {
        const u32x4 v = ld16(row + c * 8);
#pragma unroll
        for (int e = 0; e < 4; ++e) amax = fmaxf(amax, fmaxf(fabsf(bf2f_lo(v[e])), fabsf(bf2f_hi(v[e]))));
    }
    amax = wave_max(amax);
    if ((tid & 63) == 0) red[tid >> 6] = amax;
    __syncthreads();
    amax = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    int e = 0;
    if (amax > 0.f) {
        const uint32_t u = __builtin_bit_cast(uint32_t, amax);
        e = (int)(u >> 23) - 127 - ((u & 0x007FFFFFu) <= 0x00600000u ? 8 : 7);  // 448 = 1.75 * 2^8
    }
    const float s = __builtin_bit_cast(float, (uint32_t)(e + 127) << 23), inv = __builtin_bit_cast(float, (uint32_t)(127 - e) << 23);
    if (tid == 0) scale[n] = s;
    const int nst = K >> 6;
    for (int c = tid; c < (K >> 4); c += 256) {  // 16 consecutive k -> one lane's 16 bytes
        const u32x4 v0 = ld16(row + c * 16), v1 = ld16(row + c * 16 + 8);
        uint32_t q[4], o[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const uint32_t pk = j < 4 ? v0[j] : v1[j - 4];
            const uint32_t a = f2fp8(bf2f_lo(pk) * inv), b = f2fp8(bf2f_hi(pk) * inv);
            if ((j & 1) == 0) q[j >> 1] = a | (b << 8);
            else q[j >> 1] |= (a << 16) | (b << 24);
            o[j] = pack_bf2(fp82f_sw(a) * s, fp82f_sw(b) * s);
        }
        st16(row + c * 16, u32x4{o[0], o[1], o[2], o[3]});
        st16(row + c * 16 + 8, u32x4{o[4], o[5], o[6], o[7]});
        const int st = c >> 2, lane = (n & 15) + 16 * (c & 3);
        st16(Wq + (((size_t)(n >> 4) * nst + st) * 64 + lane) * 16, u32x4{q[0], q[1], q[2], q[3]});
        if (Wrow) st16(Wrow + (size_t)n * K + c * 16, u32x4{q[0], q[1], q[2], q[3]});
    }
}
void launch_quantize_fp8(bf16_t* W, uint8_t* Wq, float* scale, int N, int K, hipStream_t s, uint8_t* Wrow) {
    VC_LAUNCH(quantize_fp8_kernel, dim3((unsigned)N), dim3(256), 0, s, W, Wq, scale, N, K, Wrow);
}

// Activation operand of the e4m3 x e4m3 prefill GEMM: every token row of A [M, lda] (bf16) gets its own power-of-two
// scale by the rule of the weight rows (s_m = 2^e, the smallest e with max|A[m]| <= 448 * 2^e; 1 for an all-zero row) and
// is stored as q = e4m3(A[m] / s_m) in Q [M, K].  Same software encode as the weights, so the host restatement
// (vcoder_amd/quant.py: quantize_rows) gives identical bytes.
__global__ __launch_bounds__(256) void quant_act_rows_kernel(const bf16_t* A, int lda, uint8_t* Q, float* scale, int K) {
    __shared__ float red[4];
    const int m = blockIdx.x, tid = threadIdx.x;
    const bf16_t* row = A + (size_t)m * lda;
    float amax = 0.f;
    for (int c = tid; c < (K >> 3); c += 256) {
        const u32x4 v = ld16(row + c * 8);
#pragma unroll
        for (int e = 0; e < 4; ++e) amax = fmaxf(amax, fmaxf(fabsf(bf2f_lo(v[e])), fabsf(bf2f_hi(v[e]))));
    }
    amax = wave_max(amax);
    if ((tid & 63) == 0) red[tid >> 6] = amax;
    __syncthreads();
    amax = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    int e = 0;
    if (amax > 0.f) {
        const uint32_t u = __builtin_bit_cast(uint32_t, amax);
        e = (int)(u >> 23) - 127 - ((u & 0x007FFFFFu) <= 0x00600000u ? 8 : 7);
    }
    const float inv = __builtin_bit_cast(float, (uint32_t)(127 - e) << 23);
    if (tid == 0) scale[m] = __builtin_bit_cast(float, (uint32_t)(e + 127) << 23);
    for (int c = tid; c < (K >> 4); c += 256) {
        const u32x4 v0 = ld16(row + c * 16), v1 = ld16(row + c * 16 + 8);
        uint32_t q[4];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const uint32_t pk = j < 4 ? v0[j] : v1[j - 4];
            const uint32_t a = f2fp8(bf2f_lo(pk) * inv), b = f2fp8(bf2f_hi(pk) * inv);
            if ((j & 1) == 0) q[j >> 1] = a | (b << 8);
            else q[j >> 1] |= (a << 16) | (b << 24);
        }
        st16(Q + (size_t)m * K + c * 16, u32x4{q[0], q[1], q[2], q[3]});
    }
}
// The same with the row held in REGISTERS (round 6): NC 32-byte pieces per thread, requested back to back at clamped offsets (the
// loop form above has one load in flight per thread and reads the row twice), one pass over HBM.  K % 16 == 0, K <= NC * 256 * 16.
template <int NC>
__global__ __launch_bounds__(256) void quant_act_rows_reg_kernel(const bf16_t* A, int lda, uint8_t* Q, float* scale, int K) {
    __shared__ float red[4];
    const int m = blockIdx.x, tid = threadIdx.x;
    const bf16_t* row = A + (size_t)m * lda;
    const int n16 = K >> 4;                    // 16-element pieces of the row
    u32x4 v0[NC], v1[NC];
#pragma unroll
    for (int i = 0; i < NC; ++i) {
        const int c = min(tid + i * 256, n16 - 1);
        v0[i] = ld16(row + c * 16);
        v1[i] = ld16(row + c * 16 + 8);
    }
    float amax = 0.f;
#pragma unroll
    for (int i = 0; i < NC; ++i) {
        if (tid + i * 256 >= n16) v0[i] = v1[i] = u32x4{0u, 0u, 0u, 0u};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            amax = fmaxf(amax, fmaxf(fabsf(bf2f_lo(v0[i][e])), fabsf(bf2f_hi(v0[i][e]))));
            amax = fmaxf(amax, fmaxf(fabsf(bf2f_lo(v1[i][e])), fabsf(bf2f_hi(v1[i][e]))));
        }
    }
    amax = wave_max(amax);
    if ((tid & 63) == 0) red[tid >> 6] = amax;
    __syncthreads();
    amax = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    int e = 0;
    if (amax > 0.f) {
        const uint32_t u = __builtin_bit_cast(uint32_t, amax);
        e = (int)(u >> 23) - 127 - ((u & 0x007FFFFFu) <= 0x00600000u ? 8 : 7);
    }
    const float inv = __builtin_bit_cast(float, (uint32_t)(127 - e) << 23);
    if (tid == 0) scale[m] = __builtin_bit_cast(float, (uint32_t)(e + 127) << 23);
#pragma unroll
    for (int i = 0; i < NC; ++i) {
        const int c = tid + i * 256;
        if (c >= n16) continue;
        uint32_t q[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t p0 = j < 2 ? v0[i][2 * j] : v1[i][2 * j - 4], p1 = j < 2 ? v0[i][2 * j + 1] : v1[i][2 * j - 3];
            q[j] = f32x4_to_fp8x4_inrange(bf2f_lo(p0) * inv, bf2f_hi(p0) * inv, bf2f_lo(p1) * inv, bf2f_hi(p1) * inv);
        }
        st16(Q + (size_t)m * K + c * 16, u32x4{q[0], q[1], q[2], q[3]});
    }
}
void launch_quant_act_rows(const bf16_t* A, int lda, uint8_t* Q, float* scale, int M, int K, hipStream_t s) {
    const dim3 grid((unsigned)M), block(256);
    if (K % 16 == 0 && K <= 2 * 4096) VC_LAUNCH((quant_act_rows_reg_kernel<2>), grid, block, 0, s, A, lda, Q, scale, K);
    else if (K % 16 == 0 && K <= 4 * 4096) VC_LAUNCH((quant_act_rows_reg_kernel<4>), grid, block, 0, s, A, lda, Q, scale, K);
    else VC_LAUNCH(quant_act_rows_kernel, grid, block, 0, s, A, lda, Q, scale, K);
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
