// vc_device.h — device-side primitives for the VCoder gfx950 (CDNA4) kernels.
//
// The kernels are written once and built two ways:
//   * hipcc --offload-arch=gfx950           -> the product library (libvcoder_hip.so)
//   * host clang++ -DVC_EMU (tests/emu/)    -> a thread-per-lane functional emulator used ONLY by the
//     CPU test-suite to check index math / LDS layouts / reductions before a GPU is available.  The
//     product never links or loads the emulator.
//
// wave = 64 lanes; MFMA = v_mfma_f32_16x16x32_bf16 (A[i][k]: lane i+16*(k/8), elem k%8;
// B[k][j]: lane j+16*(k/8), elem k%8; D[i][j]: lane j+16*(i/4), reg i%4).
#pragma once
#include <stdint.h>

#ifdef VC_EMU
#include "hip_emu.h"
#else
#include <hip/hip_runtime.h>
#define VC_DYNAMIC_SMEM(type, name) extern __shared__ __attribute__((aligned(16))) type name[]
#include <stdexcept>
#include <string>
// a refused launch (too much LDS, bad grid ...) must not pass silently: the thread's stale last-error is dropped, the
// kernel launched, and a fresh error raised as an exception the C entry points turn into a vc_status
namespace vc {
inline void check_launch(const char* what) {
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) throw std::runtime_error(std::string("launch of ") + what + " failed: " + hipGetErrorString(e));
}
}
#define VC_LAUNCH(kernel, grid, block, shmem, stream, ...)                       \
    do {                                                                         \
        (void)hipGetLastError();                                                 \
        hipLaunchKernelGGL(kernel, grid, block, shmem, stream, __VA_ARGS__);     \
        vc::check_launch(#kernel);                                               \
    } while (0)
#endif

#define VC_DEV __device__ __forceinline__
#define VC_WAVE 64

namespace vc {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef uint16_t bf16_t;  // raw bf16 bit pattern

// ---- the 16-bit operand format <-> fp32 ----------------------------------------------------------
// `bf16_t` is the 16-bit container of every MFMA operand and every stored activation.  The product library holds bf16 in it (8
// significant bits, fp32's range: the benchmarked path).  Built with -DVC_F16 (libvcoder_hip_f16.so, round 6) the SAME kernels hold
// IEEE fp16 in it — 11 significant bits, the precision of the reference's own GPU path (vcoder_llava/model/builder.py:39
// torch_dtype=float16, :142 the tower cast to fp16) — and contract with v_mfma_f32_16x16x32_f16, which runs at the bf16 rate on the
// same byte layout.  Conversions saturate at +-65504 (an overflowing activation must not become inf; the fp32 residual stream,
// norms, softmax and RoPE are unaffected).  Everything else in the kernels moves 16-bit elements without looking inside them.
// true_bf2f / true_f2bf are ALWAYS bfloat16: checkpoint data that arrives as bf16 bits, the synthetic generator's value grid.
VC_DEV float true_bf2f(uint16_t b) { return __builtin_bit_cast(float, (uint32_t)b << 16); }
VC_DEV uint16_t true_f2bf(float f) {
    uint32_t u = __builtin_bit_cast(uint32_t, f);
    if ((u & 0x7FFFFFFFu) > 0x7F800000u) return (uint16_t)((u >> 16) | 0x40);  // NaN
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
typedef float f32x2 __attribute__((ext_vector_type(2)));
#ifdef VC_F16
VC_DEV float sat_f16(float f) { return __builtin_fminf(__builtin_fmaxf(f, -65504.f), 65504.f); }   // (NaN passes through)
#ifdef VC_EMU
// software IEEE fp16 <-> fp32 (RNE, subnormals) for the host build: the host compiler's _Float16 needs runtime-library calls whose
// ABI differs between toolchains
VC_DEV float h2f_sw(uint16_t h) {
    const uint32_t sign = (uint32_t)(h & 0x8000u) << 16, e = (h >> 10) & 31u, m = h & 0x3FFu;
    if (e == 0) {
        const float v = (float)m * 5.9604644775390625e-08f;   // m * 2^-24
        return __builtin_bit_cast(float, sign | __builtin_bit_cast(uint32_t, v));
    }
    if (e == 31) return __builtin_bit_cast(float, sign | 0x7F800000u | (m << 13));
    return __builtin_bit_cast(float, sign | ((e + 112u) << 23) | (m << 13));
}
VC_DEV uint16_t f2h_sw(float f) {   // |f| <= 65504 (callers saturate) or NaN
    const uint32_t u = __builtin_bit_cast(uint32_t, f), sign = (u >> 16) & 0x8000u, a = u & 0x7FFFFFFFu;
    if (a > 0x7F800000u) return (uint16_t)(sign | 0x7E00u);
    const float af = __builtin_bit_cast(float, a);
    if (af < 6.103515625e-05f) return (uint16_t)(sign | (uint32_t)(int)rintf(af * 16777216.f));   // subnormals: units of 2^-24 (RNE; 1024 -> 2^-14)
    uint32_t r = a + 0x00000FFFu + ((a >> 13) & 1u);   // RNE on bit 13
    return (uint16_t)(sign | (((r >> 23) - 112u) << 10) | ((r >> 13) & 0x3FFu));
}
VC_DEV float bf2f(bf16_t b) { return h2f_sw(b); }
VC_DEV float bf2f_lo(uint32_t packed) { return h2f_sw((uint16_t)(packed & 0xFFFFu)); }
VC_DEV float bf2f_hi(uint32_t packed) { return h2f_sw((uint16_t)(packed >> 16)); }
VC_DEV bf16_t f2bf(float f) { return f2h_sw(sat_f16(f)); }
VC_DEV uint32_t pack_bf2(float lo, float hi) { return (uint32_t)f2bf(lo) | ((uint32_t)f2bf(hi) << 16); }
#else
VC_DEV float bf2f(bf16_t b) { return (float)__builtin_bit_cast(_Float16, b); }
VC_DEV float bf2f_lo(uint32_t packed) { return (float)__builtin_bit_cast(_Float16, (uint16_t)(packed & 0xFFFFu)); }
VC_DEV float bf2f_hi(uint32_t packed) { return (float)__builtin_bit_cast(_Float16, (uint16_t)(packed >> 16)); }
VC_DEV bf16_t f2bf(float f) { return __builtin_bit_cast(bf16_t, (_Float16)sat_f16(f)); }   // RNE
typedef _Float16 f16x2_hw __attribute__((ext_vector_type(2)));
VC_DEV uint32_t pack_bf2(float lo, float hi) {   // one v_cvt_pk_f16_f32 (gfx950) behind the two clamps
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{sat_f16(lo), sat_f16(hi)}, f16x2_hw));
}
#endif
#else
VC_DEV float bf2f(bf16_t b) { return __builtin_bit_cast(float, (uint32_t)b << 16); }
VC_DEV float bf2f_lo(uint32_t packed) { return __builtin_bit_cast(float, packed << 16); }
VC_DEV float bf2f_hi(uint32_t packed) { return __builtin_bit_cast(float, packed & 0xFFFF0000u); }

#ifdef VC_EMU
VC_DEV bf16_t f2bf(float f) { return true_f2bf(f); }
#else
VC_DEV bf16_t f2bf(float f) { return __builtin_bit_cast(bf16_t, (__bf16)f); }  // v_cvt_pk_bf16_f32 (RNE)
#endif
#ifdef VC_EMU
VC_DEV uint32_t pack_bf2(float lo, float hi) { return (uint32_t)f2bf(lo) | ((uint32_t)f2bf(hi) << 16); }
#else
// ONE v_cvt_pk_bf16_f32 (two scalar conversions are not merged by hipcc 7.2: cvt, cvt, shift, or — 4 VALU per pair)
typedef __bf16 bf16x2_hw __attribute__((ext_vector_type(2)));
VC_DEV uint32_t pack_bf2(float lo, float hi) {
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{lo, hi}, bf16x2_hw));
}
#endif
#endif

// ---- fp24: the top 24 bits of an fp32 (sign, 8 exponent, 15 mantissa bits; round to nearest even) -----------------------------
// The KV-cache element of precision mode "split": |x - fp24(x)| <= 2^-17 |x| (bf16: 2^-9; the bf16 hi + lo pair the MFMA operands
// carry: ~2^-18) at 3 bytes instead of fp32's 4.  A cache row of hd elements is stored as two planes — hd x u16 (bits 31..16: what a
// truncated bf16 would hold) followed by hd x u8 (bits 15..8) — so both planes are read with aligned vector loads.
VC_DEV uint32_t f32_to_f24(float f) {   // -> the 24-bit code in bits 23..0
    uint32_t u = __builtin_bit_cast(uint32_t, f);
    if ((u & 0x7F800000u) != 0x7F800000u) u += 0x7Fu + ((u >> 8) & 1u);   // RNE on bit 8 (inf / nan pass through)
    return u >> 8;
}
VC_DEV float f24_to_f32(uint32_t hi16, uint32_t lo8) { return __builtin_bit_cast(float, (hi16 << 16) | (lo8 << 8)); }
// 8 values -> their hi plane (8 x u16) and lo plane (8 x u8)
VC_DEV void pack_f24x8(const float* v, u32x4& hi, u32x2& lo) {
    uint32_t c[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) c[e] = f32_to_f24(v[e]);
#pragma unroll
    for (int i = 0; i < 4; ++i) hi[i] = (c[2 * i] >> 8) | ((c[2 * i + 1] >> 8) << 16);
    lo[0] = (c[0] & 0xFFu) | ((c[1] & 0xFFu) << 8) | ((c[2] & 0xFFu) << 16) | ((c[3] & 0xFFu) << 24);
    lo[1] = (c[4] & 0xFFu) | ((c[5] & 0xFFu) << 8) | ((c[6] & 0xFFu) << 16) | ((c[7] & 0xFFu) << 24);
}
// element e (0..7) of a lane's 8-element group from its hi words (4 x 2 u16) and lo bytes (2 x 4 u8)
VC_DEV float f24_elem(const u32x4& hi, const u32x2& lo, int e) {
    const uint32_t w = hi[e >> 1], l = lo[e >> 2];
    const uint32_t h = (e & 1) ? (w & 0xFFFF0000u) : (w << 16);
    const int sh = 8 * (e & 3);
    const uint32_t b = sh >= 8 ? ((l >> (sh - 8)) & 0xFF00u) : ((l << 8) & 0xFF00u);
    return __builtin_bit_cast(float, h | b);
}

// ---- fp8 (OCP e4m3fn: 1-4-3, bias 7, max 448, no inf) ----------------------------------------
// Encode is software on both builds (load-time only; round-to-nearest-even, saturating) so the device, the emulator
// and vcoder_amd/quant.py produce identical bytes.  Every e4m3 value is exactly representable in bf16, so the
// decode direction is exact whichever unit does it.
VC_DEV uint8_t f2fp8(float x) {
    const uint32_t u = __builtin_bit_cast(uint32_t, x);
    const uint32_t sign = (u >> 24) & 0x80u;
    const float a = __builtin_fabsf(x);
    if (!(a < 448.f)) return (uint8_t)(sign | 0x7Eu);
    if (a < 0.015625f) return (uint8_t)(sign | (uint32_t)(int)rintf(a * 512.f));  // subnormals: units of 2^-9 (8 -> 2^-6)
    const int e = (int)((u >> 23) & 0xFFu) - 127;                                   // -6 .. 8
    const float m = __builtin_bit_cast(float, (u & 0x007FFFFFu) | 0x3F800000u);     // [1, 2)
    const uint32_t code = ((uint32_t)(e + 7) << 3) + (uint32_t)(int)rintf((m - 1.f) * 8.f);  // mantissa carry rolls over
    return (uint8_t)(sign | (code > 0x7Eu ? 0x7Eu : code));
}
// Four values that are KNOWN to be finite and within +-448 (an activation row already divided by its own power-of-two scale: the
// row's maximum maps to <= 448 by construction) -> four e4m3 bytes.  On the device one v_cvt_pk_fp8_f32 per pair (round to nearest
// even, OCP e4m3 on gfx950): for such inputs the same bytes as f2fp8 — checked on the device for EVERY bf16 value at three scales
// against the host restatement (tests/kernel_cases.py check_quant_act_rows_exhaustive) — at a twentieth of the instructions (round 6:
// the software encode was most of rmsnorm_q8_kernel's and quant_act_rows_kernel's time).  Anything that may overflow, be infinite or
// NaN keeps f2fp8 (the KV-cache writers, the load-time weight quantiser).
#ifdef VC_EMU
VC_DEV uint32_t f32x4_to_fp8x4_inrange(float a, float b, float c, float d) {
    return (uint32_t)f2fp8(a) | ((uint32_t)f2fp8(b) << 8) | ((uint32_t)f2fp8(c) << 16) | ((uint32_t)f2fp8(d) << 24);
}
#else
VC_DEV uint32_t f32x4_to_fp8x4_inrange(float a, float b, float c, float d) {
    int w = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
    w = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, w, true);
    return (uint32_t)w;
}
#endif
VC_DEV float fp82f_sw(uint32_t b) {
    const int e = (int)((b >> 3) & 15u), m = (int)(b & 7u);
    const float mag = e == 0 ? (float)m * 0.001953125f
                             : __builtin_bit_cast(float, (uint32_t)(e + 120) << 23 | (uint32_t)m << 20);
    return (b & 0x80u) ? -mag : mag;
}
// 4 packed fp8 -> 4 bf16 (two packed words), exact
#ifdef VC_EMU
VC_DEV u32x2 fp8x4_to_bf16x4(uint32_t v) {   // (every e4m3 value is exact in bf16 and in fp16)
    return u32x2{pack_bf2(fp82f_sw(v & 0xFF), fp82f_sw((v >> 8) & 0xFF)), pack_bf2(fp82f_sw((v >> 16) & 0xFF), fp82f_sw(v >> 24))};
}
#else
typedef float f32x2_hw __attribute__((ext_vector_type(2)));
VC_DEV u32x2 fp8x4_to_bf16x4(uint32_t v) {
    const f32x2_hw lo = __builtin_amdgcn_cvt_pk_f32_fp8((int)v, false), hi = __builtin_amdgcn_cvt_pk_f32_fp8((int)v, true);
    // exact values, so the RNE pack (one v_cvt_pk_bf16_f32 per pair) is lossless.  (A v_perm_b32 of the fp32 top halves
    // would do as well, but hipcc 7.2 folds __builtin_amdgcn_perm over the two halves of this builtin's result into a
    // perm of lo[0] with itself.)
    return u32x2{pack_bf2(lo[0], lo[1]), pack_bf2(hi[0], hi[1])};
}
#endif

// 4 packed e4m3 -> 4 floats (exact)
#ifdef VC_EMU
VC_DEV f32x4 fp8x4_to_f32x4(uint32_t v) {
    return f32x4{fp82f_sw(v & 0xFF), fp82f_sw((v >> 8) & 0xFF), fp82f_sw((v >> 16) & 0xFF), fp82f_sw(v >> 24)};
}
#else
VC_DEV f32x4 fp8x4_to_f32x4(uint32_t v) {
    const f32x2_hw lo = __builtin_amdgcn_cvt_pk_f32_fp8((int)v, false), hi = __builtin_amdgcn_cvt_pk_f32_fp8((int)v, true);
    return f32x4{lo[0], lo[1], hi[0], hi[1]};
}
#endif
// 4 floats -> 4 packed e4m3 (software encode: the same bytes on the device, in the emulator and in vcoder_amd/quant.py)
VC_DEV uint32_t f32x4_to_fp8x4(float a, float b, float c, float d) {
    return (uint32_t)f2fp8(a) | ((uint32_t)f2fp8(b) << 8) | ((uint32_t)f2fp8(c) << 16) | ((uint32_t)f2fp8(d) << 24);
}

// ---- MFMA ----------------------------------------------------------------------------------
#ifndef VC_EMU
#ifdef VC_F16
typedef _Float16 bf16x8_hw __attribute__((ext_vector_type(8)));   // (the operand vector type of this build: fp16)
VC_DEV f32x4 mfma16(u32x4 a, u32x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(bf16x8_hw, a), __builtin_bit_cast(bf16x8_hw, b), c, 0, 0, 0);
}
#else
typedef __bf16 bf16x8_hw __attribute__((ext_vector_type(8)));
VC_DEV f32x4 mfma16(u32x4 a, u32x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_hw, a),
                                                   __builtin_bit_cast(bf16x8_hw, b), c, 0, 0, 0);
}
#endif
// v_mfma_f32_32x32x16_bf16: A[i][k]: lane i + 32 (k / 8), elem k % 8 (i < 32, k < 16); B[k][j]: lane j + 32 (k / 8), elem k % 8;
// D[i][j]: lane j + 32 ((i / 4) % 2), reg i % 4 + 4 (i / 8)  (cdna_hip_programming.md section 3)
VC_DEV f32x16 mfma32(u32x4 a, u32x4 b, f32x16 c) {
#ifdef VC_F16
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(bf16x8_hw, a), __builtin_bit_cast(bf16x8_hw, b), c, 0, 0, 0);
#else
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_hw, a), __builtin_bit_cast(bf16x8_hw, b), c, 0, 0, 0);
#endif
}
// v_mfma_f32_16x16x4_f32 (exact fp32 FMA chain): A[i][k]: lane i+16k; B[k][j]: lane j+16k; D as for bf16
VC_DEV f32x4 mfma16_f32(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
// v_mfma_scale_f32_16x16x128_f8f6f4 with both operands OCP e4m3 and every block scale 2^0 (E8M0 127): the K = 128 fp8
// form that runs at twice the bf16 MFMA rate (the unscaled 16x16x32 fp8 MFMA runs AT the bf16 rate).  A lane holds 32
// consecutive k of its row: A[i][k]: lane i + 16*(k/32), byte k%32 (a0 = bytes 0..15, a1 = 16..31); B likewise by
// column; D as for bf16.  Both operands use the same k map, so the result does not depend on it.
typedef int i32x8_hw __attribute__((ext_vector_type(8)));
VC_DEV f32x4 mfma16_f8(u32x4 a0, u32x4 a1, u32x4 b0, u32x4 b1, f32x4 c) {
    const i32x8_hw a = {(int)a0[0], (int)a0[1], (int)a0[2], (int)a0[3], (int)a1[0], (int)a1[1], (int)a1[2], (int)a1[3]};
    const i32x8_hw b = {(int)b0[0], (int)b0[1], (int)b0[2], (int)b0[3], (int)b1[0], (int)b1[1], (int)b1[2], (int)b1[3]};
    return __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
}
VC_DEV int lane_id() { return (int)(threadIdx.x & 63); }
template <class T> VC_DEV T shfl_xor(T v, int mask) { return __shfl_xor(v, mask, 64); }
template <class T> VC_DEV T shfl(T v, int src) { return __shfl(v, src, 64); }
#endif

// ---- wave / block reductions ---------------------------------------------------------------
VC_DEV float wave_sum(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += shfl_xor(v, m);
    return v;
}
VC_DEV float wave_max(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = fmaxf(v, shfl_xor(v, m));
    return v;
}

// All-reduce (max) over the four 16-lane rows of a wave — lanes l, l^16, l^32, l^48 — which is how the 16x16 MFMA accumulator
// layouts spread one matrix column.  gfx950's v_permlane16_swap / v_permlane32_swap exchange rows inside the VALU: with both
// operands holding v, the pair afterwards holds (row 2k, row 2k+1) resp. (half 0, half 1) of v in every lane, so the maximum of
// the pair is the reduction — no ds_bpermute round trip through the LDS pipe (two of them sat on the flash kernel's per-tile
// critical path).  wave_any: a wave-uniform "does any lane ...".
// Sum over the N (4, 8, 16, 32) adjacent lanes of an aligned lane group, result in every lane.  DPP operand modifiers inside
// the VALU adds — quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror, row_mirror — instead of ds_bpermute round trips
// through the LDS pipe (what hipcc makes of __shfl_xor).  After each step all lanes of the sub-group hold the same value, so
// taking lane 7 - i / 15 - i instead of i ^ 4 / i ^ 8 adds the same two numbers: bit-identical to the xor butterfly.
#ifdef VC_EMU
template <int N> VC_DEV float lanes_sum(float s) {
#pragma unroll
    for (int mk = 1; mk < N; mk <<= 1) s += shfl_xor(s, mk);
    return s;
}
#else
template <int CTRL> VC_DEV float dpp_f32(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false));
}
template <int N> VC_DEV float lanes_sum(float s) {
    static_assert(N == 4 || N == 8 || N == 16 || N == 32, "lane group");
    s += dpp_f32<0xB1>(s);
    s += dpp_f32<0x4E>(s);
    if constexpr (N >= 8) s += dpp_f32<0x141>(s);
    if constexpr (N >= 16) s += dpp_f32<0x140>(s);
    if constexpr (N >= 32) s += shfl_xor(s, 16);
    return s;
}
#endif

// rotate-half RoPE of one pair (x, y) = (element d, element d + hd / 2): out[d] = x cos - y sin, out[d + hd/2] = y cos + x sin
// ([HF] llama/modeling_llama.py:130-160).  ONE definition with explicit fused multiply-adds and no further contraction, so that the
// prefill's split kernel and the QKV GEMM's fused epilogue produce the same bits in every build (left to -ffp-contract=fast the two
// call sites were contracted differently in the fp16-operand build: 4e-5 of the elements an ulp apart).
VC_DEV void rope_pair(float x, float y, float c, float s, float& ox, float& oy) {
#pragma clang fp contract(off)
    ox = __builtin_fmaf(x, c, -(y * s));
    oy = __builtin_fmaf(y, c, x * s);
}

// exchange inside aligned quads of lanes: the value of lane l ^ 1 / l ^ 2 (DPP quad_perm [1,0,3,2] / [2,3,0,1] on the device)
#ifdef VC_EMU
VC_DEV uint32_t quad_xor1(uint32_t v) { return shfl_xor(v, 1); }
VC_DEV uint32_t quad_xor2(uint32_t v) { return shfl_xor(v, 2); }
#else
VC_DEV uint32_t quad_xor1(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, false); }
VC_DEV uint32_t quad_xor2(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, false); }
#endif
// 4 x 4 transpose of 32-bit values across an aligned quad of lanes: afterwards P[s] of lane r (= lane & 3) is what lane s held in P[r]
VC_DEV void quad_transpose4(uint32_t (&P)[4], int r) {
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        const uint32_t got = quad_xor1((r & 1) ? P[2 * a] : P[2 * a + 1]);
        if (r & 1) P[2 * a] = got;
        else P[2 * a + 1] = got;
    }
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const uint32_t got = quad_xor2((r & 2) ? P[b] : P[b + 2]);
        if (r & 2) P[b] = got;
        else P[b + 2] = got;
    }
}

// max of three without the canonicalising v_max x, x hipcc puts in front of every fmaxf operand it cannot prove quiet
#ifdef VC_EMU
VC_DEV float vmax3(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }
VC_DEV float vmax2(float a, float b) { return fmaxf(a, b); }
#else
VC_DEV float vmax2(float a, float b) {
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
VC_DEV float vmax3(float a, float b, float c) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
#endif
#ifdef VC_EMU
VC_DEV float rows_max(float v) {
    v = fmaxf(v, shfl_xor(v, 16));
    return fmaxf(v, shfl_xor(v, 32));
}
VC_DEV bool wave_any(bool pred) {
    int x = pred ? 1 : 0;
    for (int m = 32; m >= 1; m >>= 1) x |= shfl_xor(x, m);
    return x != 0;
}
#else
VC_DEV float rows_max(float v) {
    float a = v, b = v;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    asm("v_max_f32 %0, %1, %2" : "=v"(a) : "v"(a), "v"(b));
    b = a;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    asm("v_max_f32 %0, %1, %2" : "=v"(a) : "v"(a), "v"(b));
    return a;
}
VC_DEV bool wave_any(bool pred) { return __builtin_amdgcn_ballot_w64(pred) != 0; }
#endif
// sum over the four 16-lane rows (lanes l, l ^ 16, l ^ 32, l ^ 48), result in every lane: the order ((r0 + r1) + (r2 + r3)) of the
// shfl_xor(16), shfl_xor(32) butterfly, in the VALU (permlane swaps) on the device
#ifdef VC_EMU
VC_DEV float rows_sum(float v) {
    v += shfl_xor(v, 16);
    return v + shfl_xor(v, 32);
}
#else
VC_DEV float rows_sum(float v) {
    float a = v, b = v;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    a = a + b;
    b = a;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    return a + b;
}
#endif

// 16-byte global/LDS accessors on raw pointers (the emulator's LDS race check hooks in here: tests/emu/hip_emu.h)
#ifdef VC_EMU
#define VC_LDS_R(p) vc_emu::lds_read(p)
#define VC_LDS_W(p) vc_emu::lds_write(p, false)
#else
#define VC_LDS_R(p) ((void)0)
#define VC_LDS_W(p) ((void)0)
#endif
VC_DEV u32x4 ld16(const void* p) { VC_LDS_R(p); return *reinterpret_cast<const u32x4*>(p); }
VC_DEV void st16(void* p, u32x4 v) { VC_LDS_W(p); *reinterpret_cast<u32x4*>(p) = v; }
VC_DEV u32x2 ld8(const void* p) { VC_LDS_R(p); return *reinterpret_cast<const u32x2*>(p); }
VC_DEV void st8(void* p, u32x2 v) { VC_LDS_W(p); *reinterpret_cast<u32x2*>(p) = v; }
VC_DEV f32x4 ld16f(const void* p) { VC_LDS_R(p); return *reinterpret_cast<const f32x4*>(p); }
VC_DEV void st16f(void* p, f32x4 v) { VC_LDS_W(p); *reinterpret_cast<f32x4*>(p) = v; }

// non-temporal 16-byte load for streams that are read once per launch (decode weights, KV cache rows)
#ifdef VC_EMU
VC_DEV u32x4 ld16_stream(const void* p) { return ld16(p); }
VC_DEV u32x2 ld8_stream(const void* p) { return ld8(p); }
#else
VC_DEV u32x4 ld16_stream(const void* p) { return __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p)); }
VC_DEV u32x2 ld8_stream(const void* p) { return __builtin_nontemporal_load(reinterpret_cast<const u32x2*>(p)); }
#endif

// ---- LDS-DMA: 16 bytes per lane, global -> LDS without passing through VGPRs (global_load_lds_dwordx4).
// The LDS destination is WAVE-UNIFORM base + lane*16; the global source is per lane (so LDS swizzles are applied to
// the source address).  Completion is tracked by vmcnt; hipcc waits vmcnt(0) before the next __syncthreads().
#ifdef VC_EMU
VC_DEV void glds16(const void* gsrc_lane, void* lds_wave_base) {   // lands when the lane's vmcnt wait says so (hip_emu.h)
    vc_emu::dma_issue(gsrc_lane, reinterpret_cast<char*>(lds_wave_base) + lane_id() * 16, lds_wave_base);
}
#else
VC_DEV void glds16(const void* gsrc_lane, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc_lane,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}
#endif

// ---- hand-placed synchronisation for the counted-vmcnt GEMM schedule (gemm.hip, 8-phase kernel) ---------------
#ifdef VC_EMU
VC_DEV void wg_barrier_raw() { vc_emu::block_barrier(); }   // bare s_barrier: no implied vmcnt drain
template <int N> VC_DEV void wait_vmcnt() { vc_emu::dma_wait(N); }
template <int N> VC_DEV void wait_lgkmcnt() { vc_emu::lgkm_wait(N); }
template <int P> VC_DEV void set_prio() {}
VC_DEV void sched_fence() {}
VC_DEV void pin_vgprs(f32x4&) {}
#else
// an empty volatile asm that "modifies" v: the instructions producing v cannot be moved past it
VC_DEV void pin_vgprs(f32x4& v) { asm volatile("" : "+v"(v)); }
VC_DEV void wg_barrier_raw() { __builtin_amdgcn_s_barrier(); }  // bare s_barrier: no implied vmcnt/lgkmcnt drain
template <int N> VC_DEV void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
template <int N> VC_DEV void wait_lgkmcnt() { asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory"); }
template <int P> VC_DEV void set_prio() { __builtin_amdgcn_s_setprio(P); }
VC_DEV void sched_fence() { __builtin_amdgcn_sched_barrier(0); }
#endif

// non-temporal LDS-DMA for streams read exactly once (decode weights): measured 6.8 vs 6.0 TB/s on a pure stream
#ifdef VC_EMU
VC_DEV void glds16_nt(const void* gsrc_lane, void* lds_wave_base) { glds16(gsrc_lane, lds_wave_base); }
VC_DEV void wait_vmcnt_n(int n) { vc_emu::dma_wait(n); }
#else
VC_DEV void glds16_nt(const void* gsrc_lane, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc_lane,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 2);
}
// s_waitcnt vmcnt(n) for an n that folds to a constant after unrolling (anything else drains)
VC_DEV void wait_vmcnt_n(int n) {
    switch (n) {
#define VC_VMCASE(N) case N: wait_vmcnt<N>(); break;
        VC_VMCASE(1) VC_VMCASE(2) VC_VMCASE(3) VC_VMCASE(4) VC_VMCASE(5) VC_VMCASE(6) VC_VMCASE(7) VC_VMCASE(8) VC_VMCASE(9)
        VC_VMCASE(10) VC_VMCASE(11) VC_VMCASE(12) VC_VMCASE(13) VC_VMCASE(14) VC_VMCASE(15) VC_VMCASE(16) VC_VMCASE(17)
        VC_VMCASE(18) VC_VMCASE(19) VC_VMCASE(20) VC_VMCASE(21) VC_VMCASE(22) VC_VMCASE(23) VC_VMCASE(24) VC_VMCASE(25)
        VC_VMCASE(26) VC_VMCASE(27) VC_VMCASE(28) VC_VMCASE(29) VC_VMCASE(30) VC_VMCASE(31) VC_VMCASE(32)
        VC_VMCASE(33) VC_VMCASE(34) VC_VMCASE(35) VC_VMCASE(36) VC_VMCASE(37) VC_VMCASE(38) VC_VMCASE(39) VC_VMCASE(40) VC_VMCASE(41)
        VC_VMCASE(42) VC_VMCASE(43) VC_VMCASE(44) VC_VMCASE(45) VC_VMCASE(46) VC_VMCASE(47) VC_VMCASE(48) VC_VMCASE(49) VC_VMCASE(50)
        VC_VMCASE(51) VC_VMCASE(52) VC_VMCASE(53) VC_VMCASE(54) VC_VMCASE(55) VC_VMCASE(56) VC_VMCASE(57) VC_VMCASE(58) VC_VMCASE(59)
        VC_VMCASE(60) VC_VMCASE(61) VC_VMCASE(62) VC_VMCASE(63)
#undef VC_VMCASE
        default: wait_vmcnt<0>(); break;
    }
}
#endif

// LDS hand-off between lanes of ONE wave (write by some lanes, read by others, no other wave involved): the LDS queue of
// a wave is in order, so the hardware needs nothing; the emulator's lane fibers need a rendezvous
#ifdef VC_EMU
VC_DEV void wave_lds_fence() { vc_emu::wave_sync(); }
#else
VC_DEV void wave_lds_fence() { __builtin_amdgcn_wave_barrier(); }
#endif

// ---- device-scope hand-off between workgroups (split-K finisher of the decode GEMV) ------------------------------
// Payload moves with agent-scope relaxed atomics = write-through `sc1` stores / cache-bypassing `sc1` loads; the producer
// drains them (vmcnt(0)) before it counts its arrival.  No agent-scope FENCE: on this multi-XCD part a release/acquire
// fence writes back / invalidates the XCD's whole L2 (measured: 4-8x slower launches).
#ifdef VC_EMU
// the emulator runs workgroups on several OS threads at once: real atomics.  The arrival count carries acquire / release here —
// what the drained write-through stores and cache-bypassing loads provide on the device (a plain `(*p)++` lost an arrival about
// once in 40 runs of the split-K test: two workgroups read the same count and nobody finished the tile)
VC_DEV void st_agent(float* p, float v) { __atomic_store_n(reinterpret_cast<unsigned*>(p), __builtin_bit_cast(unsigned, v), __ATOMIC_RELAXED); }
VC_DEV float ld_agent(const float* p) {
    return __builtin_bit_cast(float, __atomic_load_n(reinterpret_cast<const unsigned*>(p), __ATOMIC_RELAXED));
}
VC_DEV void st_agent_u32(unsigned* p, unsigned v) { __atomic_store_n(p, v, __ATOMIC_RELEASE); }
VC_DEV unsigned atomic_inc_agent(unsigned* p) { return __atomic_fetch_add(p, 1u, __ATOMIC_ACQ_REL); }
VC_DEV unsigned ld_agent_u32(const unsigned* p) { return __atomic_load_n(p, __ATOMIC_ACQUIRE); }
#else
VC_DEV void st_agent(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
VC_DEV float ld_agent(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
VC_DEV void st_agent_u32(unsigned* p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
VC_DEV unsigned atomic_inc_agent(unsigned* p) { return __hip_atomic_fetch_add(p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
VC_DEV unsigned ld_agent_u32(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
#endif

// ---- in-situ launch timing (vc_pool_profile; bench.py `roofline`): the first thread of every workgroup writes the low 32 bits of
// the constant-rate wall clock (| 1: zero means "not stamped") as its first and as its last instruction into the workgroup's own
// entry of the launch's slot — plain 4-byte stores, no atomics, nothing shared between workgroups (a same-address agent-scope
// atomic per workgroup measured +3 us per launch; this form is not measurable).  stamp_accumulate_kernel (select.hip) takes the
// minimum start / maximum end of a slot after the step and re-zeroes it.
constexpr int STAMP_WGS = 2048;   // workgroup entries per launch slot ({start, end} x u32 each); workgroups beyond do not stamp
#ifdef VC_EMU
VC_DEV unsigned vc_wall_clock32() { return (unsigned)vc_emu_wall_clock(); }
#else
VC_DEV unsigned vc_wall_clock32() { return (unsigned)wall_clock64(); }
#endif
// `stamp` = the launch's slot (nullptr: off, the default); wg = the workgroup's linear index in the grid
VC_DEV void stamp_begin(unsigned* stamp, unsigned wg) {
    if (stamp != nullptr && threadIdx.x == 0 && wg < (unsigned)STAMP_WGS) stamp[2 * wg] = vc_wall_clock32() | 1u;
}
VC_DEV void stamp_end(unsigned* stamp, unsigned wg) {
    if (stamp != nullptr && threadIdx.x == 0 && wg < (unsigned)STAMP_WGS) stamp[2 * wg + 1] = vc_wall_clock32() | 1u;
}

// ---- activations (fp32) --------------------------------------------------------------------
// v_exp_f32 (2^x, no range reduction or denormal handling: the softmax arguments are <= 0)
#ifdef VC_EMU
VC_DEV float fast_exp2(float x) { return exp2f(x); }
#else
VC_DEV float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
#endif
VC_DEV float quick_gelu(float x) { return x / (1.0f + __expf(-1.702f * x)); }   // x*sigmoid(1.702x)
VC_DEV float erf_gelu(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }
VC_DEV float silu(float x) { return x / (1.0f + __expf(-x)); }

}  // namespace vc
