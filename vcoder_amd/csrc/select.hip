// select.hip — the tail of a decode step: token selection (greedy or sampled), EOS / keyword-stop bookkeeping, embedding
// of the selected token into the next step's residual stream and the per-row position advance — one workgroup per row.
//
//   greedy   : fp32 argmax, lowest index on ties            [HF] generation/utils.py:2894,2925
//   sampling : temperature -> top-k -> top-p -> multinomial  [HF] generation/logits_process.py (TemperatureLogitsWarper,
//              TopKLogitsWarper, TopPLogitsWarper) + utils.py:2921-2923; what vcoder_llava/serve/cli.py:122-132 asks for
//              (do_sample=True, temperature=0.2) and serve/chat.py:141-151 (temperature, top_p)
//   finished rows emit pad, EOS finishes a row                [HF] generation/utils.py:2928-2929
//   keyword stop: suffix match of the row's ids               vcoder_llava/mm_utils.py:128-151 (KeywordsStoppingCriteria)
//   embedding of the chosen token                             [HF] llama/modeling_llama.py:377
//
// Everything a row needs lives in its RowState record in device memory (kernels.h: RS_*), so ONE captured hipGraph serves
// every step of every request: rows of different requests (different prompt lengths, step counts, sampling parameters,
// stop sequences) can share a decode step.  Rows are independent; a row only ever touches its own record.
//
// Sampling runs entirely on the device: the scaled logits of the row sit in LDS (V * 4 bytes; 125 KiB for Llama's 32000),
// top-k and top-p thresholds are found by bit-wise bisection over the order-preserving integer image of the fp32 values
// (exact: 32 counting / mass passes over LDS, no sort), and the draw is a Gumbel-max over the kept set with a counter-based
// generator keyed by (seed, step, vocabulary index) — the same seed reproduces the same tokens, on any launch geometry.
#include "vc_device.h"
#include "kernels.h"

namespace vc {

VC_DEV void argmax_combine2(float& v, int& i, float ov, int oi) {
    if (ov > v || (ov == v && oi < i)) { v = ov; i = oi; }
}

// order-preserving map fp32 -> uint32 (negative values reversed, positive offset): a < b  <=>  key(a) < key(b)
VC_DEV uint32_t fkey(float f) {
    const uint32_t u = __builtin_bit_cast(uint32_t, f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

VC_DEV uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16;
    return x;
}
// uniform strictly inside (0,1) from a 32-bit hash: 23 random bits, so that (h >> 9) + 0.5 is exact in fp32 and the
// largest value is 1 - 2^-24 (with 24 bits, 16777215.5 rounds to 2^24 and u == 1 makes the Gumbel term +inf)
VC_DEV float uniform_open01(uint32_t h) { return ((float)(h >> 9) + 0.5f) * (1.0f / 8388608.0f); }
// uniform in (0,1) from (seed, step, index), never 0 or 1
VC_DEV float sample_uniform(uint32_t seed_lo, uint32_t seed_hi, uint32_t step, uint32_t idx) {
    uint32_t h = mix32(idx * 0x9E3779B1u + seed_lo);
    h = mix32(h ^ (step * 0x85EBCA6Bu + seed_hi));
    return uniform_open01(h);
}

struct BlockRed {  // cross-wave scratch of the 1024-thread workgroup
    float f[16];
    int i[16];
};

VC_DEV int block_sum_i(int v, BlockRed& r, int lane, int wave) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += shfl_xor(v, m);
    if (lane == 0) r.i[wave] = v;
    __syncthreads();
    int s = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) s += r.i[w];
    __syncthreads();
    return s;
}
VC_DEV float block_sum_f(float v, BlockRed& r, int lane, int wave) {
    v = wave_sum(v);
    if (lane == 0) r.f[wave] = v;
    __syncthreads();
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < 16; ++w) s += r.f[w];  // fixed order: bit-reproducible
    __syncthreads();
    return s;
}
VC_DEV void block_argmax(float& v, int& i, BlockRed& r, int lane, int wave) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        const float ov = shfl_xor(v, m);
        const int oi = shfl_xor(i, m);
        argmax_combine2(v, i, ov, oi);
    }
    if (lane == 0) { r.f[wave] = v; r.i[wave] = i; }
    __syncthreads();
    v = r.f[0];
    i = r.i[0];
#pragma unroll
    for (int w = 1; w < 16; ++w) argmax_combine2(v, i, r.f[w], r.i[w]);
    __syncthreads();
}

// bf16 embedding row -> fp32 residual row + sum-of-squares partials + xg = bf16(x * g) (the first GEMV's operand)
// xg_lo != nullptr (precision mode "split"): additionally the lo row bf16(x * g - xg) of the stacked hi / lo group
// sp_lo != nullptr: the lo plane of an inexact checkpoint's table row (x = hi + lo; split mode)
// chunk c (8 columns) of the row; returns the chunk's sum of squares
VC_DEV float embed_chunk_ssq(const bf16_t* sp, float* dp, const float* gw, bf16_t* xg, int c, bf16_t* xg_lo, const bf16_t* sp_lo) {
    const u32x4 v = ld16(sp + c * 8);
    f32x4 a = {bf2f_lo(v[0]), bf2f_hi(v[0]), bf2f_lo(v[1]), bf2f_hi(v[1])};
    f32x4 b = {bf2f_lo(v[2]), bf2f_hi(v[2]), bf2f_lo(v[3]), bf2f_hi(v[3])};
    if (sp_lo != nullptr) {
        const u32x4 w = ld16(sp_lo + c * 8);
        a = a + f32x4{bf2f_lo(w[0]), bf2f_hi(w[0]), bf2f_lo(w[1]), bf2f_hi(w[1])};
        b = b + f32x4{bf2f_lo(w[2]), bf2f_hi(w[2]), bf2f_lo(w[3]), bf2f_hi(w[3])};
    }
    st16f(dp + c * 8, a);
    st16f(dp + c * 8 + 4, b);
    const f32x4 g0 = ld16f(gw + c * 8), g1 = ld16f(gw + c * 8 + 4);
    const f32x4 ta = {a[0] * g0[0], a[1] * g0[1], a[2] * g0[2], a[3] * g0[3]};
    const f32x4 tb = {b[0] * g1[0], b[1] * g1[1], b[2] * g1[2], b[3] * g1[3]};
    const u32x4 hi = {pack_bf2(ta[0], ta[1]), pack_bf2(ta[2], ta[3]), pack_bf2(tb[0], tb[1]), pack_bf2(tb[2], tb[3])};
    st16(xg + c * 8, hi);
    if (xg_lo != nullptr)
        st16(xg_lo + c * 8, u32x4{pack_bf2(ta[0] - bf2f_lo(hi[0]), ta[1] - bf2f_hi(hi[0])),
                                  pack_bf2(ta[2] - bf2f_lo(hi[1]), ta[3] - bf2f_hi(hi[1])),
                                  pack_bf2(tb[0] - bf2f_lo(hi[2]), tb[1] - bf2f_hi(hi[2])),
                                  pack_bf2(tb[2] - bf2f_lo(hi[3]), tb[3] - bf2f_hi(hi[3]))});
    return ((a[0] * a[0] + a[1] * a[1]) + (a[2] * a[2] + a[3] * a[3])) +
           ((b[0] * b[0] + b[1] * b[1]) + (b[2] * b[2] + b[3] * b[3]));
}
VC_DEV void embed_row_ssq(const bf16_t* sp, float* dp, float* ssq_row, const float* gw, bf16_t* xg, int D, int npart,
                          int lane, bf16_t* xg_lo = nullptr, const bf16_t* sp_lo = nullptr) {
    float ss = 0.f;
    for (int c = lane; c < D / 8; c += 64) ss += embed_chunk_ssq(sp, dp, gw, xg, c, xg_lo, sp_lo);
    ss = wave_sum(ss);
    for (int q = lane; q < npart; q += 64) ssq_row[q] = q == 0 ? ss : 0.f;
}

__global__ __launch_bounds__(1024) void select_embed_kernel(SelectArgs p) {
    VC_DYNAMIC_SMEM(float, zs);  // [V] scaled logits of the row (sampling only)
    __shared__ BlockRed red;
    __shared__ int tok_s;
    const int r = p.row0 + (int)blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int* rs = p.rows + (size_t)r * RS_STRIDE;
    if (!rs[RS_ACTIVE]) return;
    const int step = rs[RS_STEP];
    const float* lg = p.logits + (size_t)blockIdx.x * p.ldl;
    const int V = p.V;
    float best = -INFINITY;
    int bi = 0x7FFFFFFF;
    if (!rs[RS_SAMPLE]) {
        for (int i = tid * 4; i < V; i += 4096) {
            const f32x4 v = ld16f(lg + i);
#pragma unroll
            for (int e = 0; e < 4; ++e) argmax_combine2(best, bi, v[e], i + e);
        }
        block_argmax(best, bi, red, lane, wave);
    } else {
        const float inv_t = __builtin_bit_cast(float, rs[RS_INVTEMP]);
        const int top_k = rs[RS_TOPK];
        const float top_p = __builtin_bit_cast(float, rs[RS_TOPP]);
        const uint32_t seed_lo = (uint32_t)rs[RS_SEED_LO], seed_hi = (uint32_t)rs[RS_SEED_HI];
        const bool staged = p.lds_floats >= V;
        auto z = [&](int i) { return staged ? zs[i] : lg[i] * inv_t; };
        if (staged) {
            for (int i = tid; i < V; i += 1024) zs[i] = lg[i] * inv_t;
            __syncthreads();
        }
        // ---- top-k: keep z >= (k-th largest value); ties at the threshold are all kept, as TopKLogitsWarper does
        uint32_t kmin = 0;  // keep keys >= kmin
        if (top_k > 0 && top_k < V) {
            uint32_t t = 0;
            for (int bit = 31; bit >= 0; --bit) {
                const uint32_t cand = t | (1u << bit);
                int c = 0;
                for (int i = tid; i < V; i += 1024) c += fkey(z(i)) >= cand;
                if (block_sum_i(c, red, lane, wave) >= top_k) t = cand;
            }
            kmin = t;
        }
        // ---- top-p over the survivors: ascending cumulative probability <= 1 - top_p is removed (TopPLogitsWarper)
        uint32_t kgt = 0;  // keep keys > kgt (0 = none removed: no finite value has key 0)
        bool only_max = false;
        if (top_p < 1.0f) {
            float mx = -INFINITY;
            for (int i = tid; i < V; i += 1024) mx = fmaxf(mx, z(i));
            int dummy = 0;
            block_argmax(mx, dummy, red, lane, wave);
            float zsum = 0.f;
            for (int i = tid; i < V; i += 1024) {
                const float v = z(i);
                if (fkey(v) >= kmin) zsum += __expf(v - mx);
            }
            zsum = block_sum_f(zsum, red, lane, wave);
            const float target = (1.0f - top_p) * zsum;
            if (!(top_p > 0.f)) {
                only_max = true;
            } else {
                uint32_t t = 0;
                for (int bit = 31; bit >= 0; --bit) {
                    const uint32_t cand = t | (1u << bit);
                    float mass = 0.f;
                    for (int i = tid; i < V; i += 1024) {
                        const float v = z(i);
                        const uint32_t k = fkey(v);
                        if (k >= kmin && k <= cand) mass += __expf(v - mx);
                    }
                    if (block_sum_f(mass, red, lane, wave) <= target) t = cand;
                }
                kgt = t;
            }
        }
        // ---- multinomial over the kept set as a Gumbel-max: argmax_i (z_i - log(-log u_i)) ~ softmax(z) restricted to it
        for (int i = tid; i < V; i += 1024) {
            const float v = z(i);
            const uint32_t k = fkey(v);
            if (k < kmin || k <= kgt || !(v > -INFINITY)) continue;
            const float g = only_max ? 0.f : -__logf(-__logf(sample_uniform(seed_lo, seed_hi, (uint32_t)step, (uint32_t)i)));
            argmax_combine2(best, bi, v + g, i);
        }
        block_argmax(best, bi, red, lane, wave);
    }
    // ---- bookkeeping (one lane) ----------------------------------------------------------------------------------
    if (tid == 0) {
        const int eos = rs[RS_EOS], pad = rs[RS_PAD], nstop = rs[RS_NSTOP], max_new = rs[RS_MAXNEW];
        const bool can_finish = eos >= 0 || nstop > 0;
        const int was_finished = can_finish ? rs[RS_FINISHED] : 0;
        int tok = bi;
        if (can_finish) {
            if (was_finished) tok = pad;
            if (eos >= 0 && tok == eos) rs[RS_FINISHED] = 1;
        }
        p.next_tok[r] = tok;
        int* out = p.out_ids + rs[RS_OUT_OFF];
        if (step < max_new) out[step] = tok;
        if (nstop > 0 && !was_finished) {  // suffix match of the row's ids (prompt tail | generated so far | tok)
            bool hit = false;
            for (int sq = 0; sq < nstop; ++sq) {
                const int* e = rs + RS_STOP + sq * (1 + VC_MAX_STOP_LEN);
                const int L = e[0];
                bool ok = L > 0;
                for (int i = 0; i < L && ok; ++i) {
                    const int back = L - 1 - i;  // 0 = the token just selected
                    int v;
                    if (back == 0) v = tok;
                    else if (step - back >= 0) v = out[step - back];
                    else v = rs[RS_TAIL + (VC_MAX_STOP_LEN - 1) + (step - back)];
                    ok = v == e[1 + i];
                }
                hit = hit || ok;
            }
            if (hit) rs[RS_FINISHED] = 1;
        }
        tok_s = tok;
        if (p.advance & 1) rs[RS_STEP] = step + 1;
        if (p.advance & 2) rs[RS_POS] += 1;
    }
    __syncthreads();
    if (p.embed == nullptr) return;
    const int tok = tok_s;
    const int G = p.xg_G;  // split mode: row r -> hi at row (r / G) * 2G + r % G of xg, lo G rows further
    const size_t xrow = G ? (size_t)(r / G) * 2 * G + r % G : (size_t)r;
    const bf16_t* sp = p.embed + (size_t)tok * p.D;
    const bf16_t* sp_lo = p.embed_lo ? p.embed_lo + (size_t)tok * p.D : nullptr;
    bf16_t* xg = p.xg + xrow * p.D;
    bf16_t* xg_lo = G ? p.xg + (xrow + G) * p.D : nullptr;
    const int nch = p.D / 8;
    if (nch > 1024) {   // rows wider than 8192: one wave walks the row
        if (wave == 0) embed_row_ssq(sp, p.x + (size_t)r * p.D, p.ssq + (size_t)r * p.npart, p.xg_w, xg, p.D, p.npart, lane, xg_lo, sp_lo);
        return;
    }
    // The row's 16-byte chunks over the WHOLE block, one per thread (round 6: one wave used to walk them, D / 512 dependent round
    // trips to a table row nobody has touched, on the critical path of every decode step); the sum of squares keeps its order — a
    // lane's chunks lane, lane + 64, ... added in sequence, then the wave reduction — through an LDS array of per-chunk partials
    __shared__ float part[1024];
    if (tid < nch) part[tid] = embed_chunk_ssq(sp, p.x + (size_t)r * p.D, p.xg_w, xg, tid, xg_lo, sp_lo);
    __syncthreads();
    if (wave == 0) {
        float ss = 0.f;
        for (int c = lane; c < nch; c += 64) ss += part[c];
        ss = wave_sum(ss);
        float* ssq_row = p.ssq + (size_t)r * p.npart;
        for (int q = lane; q < p.npart; q += 64) ssq_row[q] = q == 0 ? ss : 0.f;
    }
}

// ---- in-situ timing slots (vc_device.h stamp_begin / stamp_end; engine.hip vc_pool_profile) ------------------------------------
// One workgroup per launch slot: earliest start / latest end over the slot's workgroup entries (32-bit ticks: compared relative to
// one valid entry, so a wrap of the low word inside a launch does not matter), the slot re-zeroed.  The workgroup that finishes
// last (agent-scope hand-off as in decode.hip's split-K: sc1 stores, arrival counter, sc1 loads) walks the slots in launch order
// and adds, for the slot's kind — 5 slots per layer (qkv 0, attention 1, o 2, gate/up 3, down 4), the last slot lm_head (5) —
//   exec   = latest end - earliest start of the launch's workgroups, and
//   period = latest end - latest end of the PREVIOUS stamped launch of the step (the first launch: its exec): the launch as the
//            step's dependency chain pays for it, dispatch / drain / inter-kernel gap included — what bench.py's `roofline` uses
// to acc[kind] = {exec ticks, period ticks, launches}.  A slot nobody stamped (a decode attention over free rows only) is skipped.
static_assert(STAMP_SLOT_WORDS == 2 * (size_t)STAMP_WGS, "slot size");
__global__ __launch_bounds__(256) void stamp_accumulate_kernel(unsigned* stamps, int n, int layers, unsigned long long* acc,
                                                               unsigned* scratch /*[1 + 2 n]: arrival counter, {start, end} per slot*/) {
    __shared__ unsigned ref;
    __shared__ int lo, hi;
    __shared__ unsigned last;
    const int tid = threadIdx.x, j = blockIdx.x;
    unsigned* slot = stamps + (size_t)j * STAMP_SLOT_WORDS;
    if (tid == 0) {
        ref = 0;
        lo = 0x7fffffff;
        hi = -0x7fffffff;
    }
    __syncthreads();
    // all loads first (one 8-byte load per entry, issued back to back), then the re-zeroing stores: a store that depends on its own
    // load serialises the eight round trips of a thread
    u32x2 tt[STAMP_WGS / 256];
#pragma unroll
    for (int i = 0; i < STAMP_WGS / 256; ++i) tt[i] = ld8(slot + 2 * (i * 256 + tid));
    unsigned t0[STAMP_WGS / 256], t1[STAMP_WGS / 256];
#pragma unroll
    for (int i = 0; i < STAMP_WGS / 256; ++i) {
        t0[i] = tt[i][0];
        t1[i] = tt[i][1];
        if (t0[i] | t1[i]) st8(slot + 2 * (i * 256 + tid), u32x2{0u, 0u});
        if (t0[i] != 0 && t1[i] != 0) ref = t0[i];   // any valid entry serves as the reference (benign race: all are valid)
    }
    __syncthreads();
    const unsigned r = ref;
    int mylo = 0x7fffffff, myhi = -0x7fffffff;
#pragma unroll
    for (int i = 0; i < STAMP_WGS / 256; ++i)
        if (r != 0 && t0[i] != 0 && t1[i] != 0) {
            mylo = min(mylo, (int)(t0[i] - r));
            myhi = max(myhi, (int)(t1[i] - r));
        }
    if (myhi >= mylo) {
#ifdef VC_EMU
        __atomic_fetch_min(&lo, mylo, __ATOMIC_RELAXED);
        __atomic_fetch_max(&hi, myhi, __ATOMIC_RELAXED);
#else
        atomicMin(&lo, mylo);
        atomicMax(&hi, myhi);
#endif
    }
    __syncthreads();
    if (tid == 0) {
        const bool ok = r != 0 && hi >= lo;
        // {0, 0} = not stamped (a stamped end is odd: | 1 survives r + hi only by luck, so validity travels in the start word's bit 0)
        st_agent_u32(scratch + 1 + 2 * j, ok ? ((r + (unsigned)lo) | 1u) : 0u);
        st_agent_u32(scratch + 2 + 2 * j, ok ? r + (unsigned)hi : 0u);
        wait_vmcnt<0>();
        last = atomic_inc_agent(scratch) == (unsigned)(n - 1) ? 1u : 0u;
    }
    __syncthreads();
    if (!last) return;
    // the last arriver: all its threads fetch the per-slot results side by side (sc1 loads), one thread walks them in launch order
    constexpr int MAXS = 512;
    __shared__ unsigned sst[MAXS], sen[MAXS];
    for (int s_ = tid; s_ < n && s_ < MAXS; s_ += 256) {
        sst[s_] = ld_agent_u32(scratch + 1 + 2 * s_);
        sen[s_] = ld_agent_u32(scratch + 2 + 2 * s_);
    }
    __syncthreads();
    if (tid != 0) return;
    unsigned prev_end = 0;
    bool have_prev = false;
    unsigned long long ex[PROF_KINDS] = {}, pe[PROF_KINDS] = {}, cn[PROF_KINDS] = {};
    for (int s_ = 0; s_ < n && s_ < MAXS; ++s_) {
        const unsigned st = sst[s_], en = sen[s_];
        if (!(st & 1u)) continue;
        const int kind = s_ < 5 * layers ? s_ % 5 : 5;
        const unsigned e_ = en - (st & ~1u);
        ex[kind] += e_;
        pe[kind] += have_prev ? (unsigned)(en - prev_end) : e_;
        cn[kind] += 1;
        prev_end = en;
        have_prev = true;
    }
    for (int k = 0; k < PROF_KINDS; ++k) {   // only this thread of this launch touches acc; launches are stream-ordered
        acc[3 * k] += ex[k];
        acc[3 * k + 1] += pe[k];
        acc[3 * k + 2] += cn[k];
    }
    st_agent_u32(scratch, 0u);   // re-armed for the next step
}
void launch_stamp_accumulate(unsigned* stamps, int n, int layers, unsigned long long* acc, unsigned* scratch, hipStream_t s) {
    VC_LAUNCH(stamp_accumulate_kernel, dim3((unsigned)n), dim3(256), 0, s, stamps, n, layers, acc, scratch);
}

void launch_select_embed(const SelectArgs& a0, hipStream_t s) {
    SelectArgs a = a0;
    // the row's scaled logits are staged in LDS when they fit (sampling only reads them ~70 times)
    size_t lds = (size_t)a.V * 4;
    if (lds > 150 * 1024) lds = 0;
    a.lds_floats = (int)(lds / 4);
#ifndef VC_EMU
    static size_t allowed = 0;
    if (lds > 48 * 1024 && lds > allowed) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(select_embed_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)lds);
        allowed = lds;
    }
#endif
    VC_LAUNCH(select_embed_kernel, dim3(a.nrows), dim3(1024), lds, s, a);
}

// test hook: the sampler's hash -> uniform map on explicit hash values (the extreme h = 0xFFFFFFFF must stay below 1)
__global__ __launch_bounds__(64) void uniform_probe_kernel(const uint32_t* h, float* u, float* gumbel, int n) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    u[i] = uniform_open01(h[i]);
    gumbel[i] = -__logf(-__logf(u[i]));
}
void launch_uniform_probe(const uint32_t* h, float* u, float* gumbel, int n, hipStream_t s) {
    VC_LAUNCH(uniform_probe_kernel, dim3((n + 63) / 64), dim3(64), 0, s, h, u, gumbel, n);
}

// embedding + sum-of-squares partials for tokens supplied by the host (vc_decode_step with explicit tokens)
__global__ __launch_bounds__(256) void embed_tokens_ssq_kernel(const int* tok, const bf16_t* embed, float* x, float* ssq,
                                                               const float* xg_w, bf16_t* xg, int B, int D, int npart, int G,
                                                               const bf16_t* embed_lo) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= B) return;
    const size_t xrow = G ? (size_t)(row / G) * 2 * G + row % G : (size_t)row;
    embed_row_ssq(embed + (size_t)tok[row] * D, x + (size_t)row * D, ssq + (size_t)row * npart, xg_w, xg + xrow * D, D,
                  npart, threadIdx.x & 63, G ? xg + (xrow + G) * D : nullptr, embed_lo ? embed_lo + (size_t)tok[row] * D : nullptr);
}
void launch_embed_tokens_ssq(const int* tok, const bf16_t* embed, float* x, float* ssq, const float* xg_w, bf16_t* xg, int B,
                             int D, int npart, hipStream_t s, int xg_G, const bf16_t* embed_lo) {
    VC_LAUNCH(embed_tokens_ssq_kernel, dim3((B + 3) / 4), dim3(256), 0, s, tok, embed, x, ssq, xg_w, xg, B, D, npart, xg_G, embed_lo);
}

}  // namespace vc
