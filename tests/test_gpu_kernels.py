"""`-m gpu`: every HIP kernel at the TRUE shapes of VCoder-DS LLaVA-1.5-7b / CLIP ViT-L/14@336, called through
the C ABI of libvcoder_hip.so (include/vcoder_kernels.h) and compared with the oracle."""
import ctypes
import os

import pytest

import kernel_cases as kc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def be():
    return kc.HipBackend()


@pytest.fixture(scope="module")
def be16():
    """libvcoder_hip_f16.so: the same kernels built with -DVC_F16 (IEEE fp16 MFMA operands, round 6)"""
    return kc.HipBackend("fp16")


def test_fp16_operand_library_kernels(be16):
    """The -DVC_F16 build of the kernels at true shapes — v_mfma_f32_16x16x32_f16 fragments, the saturating fp16 conversions, the
    fp16 forms of every epilogue: GEMM (8-phase 256 x 256 incl. a split-K round, the 128 x 128 form), the decode GEMV ring kernel,
    flash attention, the fused QKV epilogue (bit for bit against GEMM + split), against the same float64 references and tolerances
    as the bf16 build (whose operands hold 3 bits less)."""
    assert be16.lib.vc_operand_format() == 1
    kc.check_gemm(be16, 1216, 12288, 4096, 0, False)
    kc.check_gemm(be16, 1154, 4096, 1024, 1, True)
    kc.check_gemm(be16, 1216, 22016, 4096, 5, False)
    kc.check_gemm(be16, 9728, 4096, 4096, 4, False, ws_mb=64)
    kc.check_gemm(be16, 70, 264, 192, 3, True)
    kc.check_gemv(be16, 8, 12288, 4096, 0)
    kc.check_gemv(be16, 8, 4096, 11008, 2)
    kc.check_attention(be16, 2, 32, 1216, 128, True)
    kc.check_attention(be16, 3, 16, 577, 64, False)
    kc.check_gemm_qkv_fused(be16, 2, 1216, 32, 4096)


# (M, N, K, epilogue): ViT QKV/out/fc1/fc2, patchify, adapters, Llama qkv/o/gate-up/down, ragged edges
@pytest.mark.parametrize("M,N,K,epi,bias", [
    (1154, 3072, 1024, 0, True), (1154, 1024, 1024, 4, True), (1154, 4096, 1024, 1, True), (1154, 1024, 4096, 4, True),
    (1152, 1024, 640, 3, False), (1152, 4096, 1024, 2, True), (1152, 4096, 4096, 0, True),
    (1216, 12288, 4096, 0, False), (1216, 4096, 4096, 4, False), (1216, 22016, 4096, 5, False),
    (1216, 4096, 11008, 4, False), (70, 264, 192, 0, True), (300, 320, 256, 3, False)])
def test_gemm(be, M, N, K, epi, bias):
    kc.check_gemm(be, M, N, K, epi, bias)


@pytest.mark.parametrize("B,T,H,K,ws,f8,kv8", [(8, 1216, 32, 4096, 64, False, False), (2, 1217, 40, 5120, 0, False, False),
                                               (2, 1217, 40, 5120, 64, False, False), (3, 1190, 32, 4096, 0, True, True),
                                               (3, 1190, 32, 4096, 64, True, True), (1, 96, 2, 256, 0, False, False)])
def test_gemm_qkv_fused_epilogue(be, B, T, H, K, ws, f8, kv8):
    """Round 6: the QKV GEMM with RoPE + head split + KV write in its epilogue at the 7b (B = 8: split-K remainder round +
    the QKV fix-up launch) and 13b geometries, sample lengths off the 32-token grid, the e4m3 operand / e4m3 cache form: every output
    bit for bit what vck_gemm + vck_qkv_split_kv / _kv8 write (with a split-K workspace AND a length off the grid the two forms slice
    different tiles: single bf16 roundings may then differ, nothing else)."""
    kc.check_gemm_qkv_fused(be, B, T, H, K, ws_mb=ws, f8=f8, kv8=kv8)


@pytest.mark.parametrize("M,N,K,epi,bias,ws", [(1154, 3072, 1024, 0, True, 0), (1154, 4096, 1024, 1, True, 0), (1152, 4096, 4096, 2, True, 0),
                                               (1216, 12288, 4096, 3, False, 0), (1216, 22016, 4096, 5, False, 0),
                                               (9728, 4096, 4096, 4, False, 64), (9728, 12288, 4096, 0, False, 64),
                                               (70, 264, 192, 0, True, 0)])
def test_gemm_mfma_32x32x16(be, M, N, K, epi, bias, ws):
    """The 8-phase schedule on v_mfma_f32_32x32x16_bf16 (vck_set_gemm_variant 7: every size, incl. ragged tiles and split-K
    remainder rounds) against float64 — fragment maps, the (row >> 1) & 7 tile swizzle and the 32 x 32 C/D layout on hardware."""
    be.lib.vck_set_gemm_variant(7)
    try:
        kc.check_gemm(be, M, N, K, epi, bias, ws_mb=ws)
    finally:
        be.lib.vck_set_gemm_variant(-1)


@pytest.mark.parametrize("M,N,K,epi", [(8, 12288, 4096, 0), (8, 4096, 4096, 2), (8, 22016, 4096, 3),
                                       (8, 4096, 11008, 2), (8, 32000, 4096, 1), (16, 15360, 5120, 0), (1, 320, 256, 1),
                                       (3, 48, 288, 1)])
def test_gemv(be, M, N, K, epi):
    kc.check_gemv(be, M, N, K, epi)


@pytest.mark.parametrize("M,N,K,epi,norm", [(8, 12288, 4096, 0, True), (8, 4096, 4096, 2, False), (8, 22016, 4096, 3, True),
                                            (8, 4096, 11008, 2, False), (16, 15360, 5120, 0, True),
                                            (16, 5120, 13824, 2, False), (3, 48, 320, 1, False), (5, 32, 320, 1, True)])
def test_gemv_fp8(be, M, N, K, epi, norm):
    kc.check_gemv_fp8(be, M, N, K, epi, norm)


@pytest.mark.parametrize("M,N,K,epi,ws", [(1216, 12288, 4096, 0, 0), (1216, 4096, 4096, 4, 0), (1216, 22016, 4096, 5, 0),
                                          (1216, 4096, 11008, 4, 0), (9728, 4096, 4096, 4, 64), (9728, 12288, 4096, 0, 64),
                                          (2432, 5120, 13824, 4, 64), (70, 272, 384, 0, 0), (16, 16, 128, 0, 0)])
def test_gemm_f8(be, M, N, K, epi, ws):
    """W8A8 prefill GEMM on v_mfma_scale_f32_16x16x128_f8f6f4 at the 7b / 13b decoder shapes (incl. split-K remainder
    rounds): quantisers bit-identical to vcoder_amd/quant.py, products exact, fp32 accumulation."""
    kc.check_gemm_f8(be, M, N, K, epi, ws_mb=ws)


@pytest.mark.parametrize("M,N,K,ks", [(8, 4096, 4096, 2), (8, 4096, 11008, 3), (16, 5120, 13824, 4), (16, 5120, 5120, 4),
                                      (3, 48, 320, 3)])
def test_gemv_splitk(be, M, N, K, ks):
    kc.check_gemv_splitk(be, M, N, K, ks)


@pytest.mark.parametrize("M,N,K,epi,bias", [(9728, 4096, 4096, 4, False), (9728, 12288, 4096, 0, False),
                                            (13848, 4096, 1024, 1, True), (9728, 4096, 11008, 4, False),
                                            (13824, 4096, 4096, 2, True)])
def test_gemm_splitk_remainder_round(be, M, N, K, epi, bias):
    """The shapes whose last round of 256x256 tiles is short (608 = 2x256+96, 1824 = 7x256+32, 880, 864 tiles): K-slices of
    the remainder tiles + fix-up launch against the oracle."""
    kc.check_gemm(be, M, N, K, epi, bias, ws_mb=64)


def test_small_ops(be):
    kc.check_interleave(be, 11008, 256)
    kc.check_layernorm(be, 4616, 1024)
    kc.check_layernorm(be, 5, 128)
    kc.check_rmsnorm(be, 9728, 4096)
    kc.check_rmsnorm(be, 8, 5120, gather=True)
    kc.check_rmsnorm(be, 3, 256)
    kc.check_rmsnorm_q8(be, 9728, 4096)
    kc.check_rmsnorm_q8(be, 37, 5120)
    kc.check_quant_act_rows_exhaustive(be)   # every bf16 value at three row scales: the hardware fp8 conversion == the host's software encode
    kc.check_im2col(be, 3, 336, 14, 640)
    kc.check_im2col(be, 2, 56, 14, 640)
    kc.check_vit_embed_ln(be, 3, 577, 1024)
    kc.check_select_rows(be, 3, 577, 1024)
    kc.check_splice(be, 4096)
    kc.check_greedy(be, 8, 32000)
    kc.check_synth(be, n=1 << 20)


@pytest.mark.parametrize("B,T,H,hd,rope", [(2, 1216, 4, 128, True), (3, 577, 16, 64, False), (1, 70, 2, 128, True)])
def test_qkv_split(be, B, T, H, hd, rope):
    kc.check_qkv_split(be, B, T, H, hd, rope)


@pytest.mark.parametrize("B,H,T,hd,causal,spike", [(2, 16, 577, 64, False, False), (1, 4, 577, 64, False, True),
                                                   (2, 8, 1216, 128, True, False), (1, 2, 1216, 128, True, True),
                                                   (1, 1, 17, 64, False, False), (1, 2, 200, 128, True, True)])
def test_attention(be, B, H, T, hd, causal, spike):
    kc.check_attention(be, B, H, T, hd, causal, spike=spike)


def test_fused_decode_kernels(be):
    kc.check_gemv_norm_chain(be, 8, 4096, 12288)
    kc.check_gemv_norm_chain(be, 16, 5120, 1024, seed=1)
    kc.check_gemv_norm_chain(be, 24, 4096, 12288, seed=2)     # the decode pool: 17..32 rows per weight pass
    kc.check_gemv_norm_chain(be, 32, 4096, 4096, seed=3)
    kc.check_gemv_norm_chain(be, 3, 256, 64, seed=2)
    kc.check_attention_decode_fused(be, 8, 32, 128, 1216)
    kc.check_attention_decode_fused(be, 2, 4, 128, 1343)
    kc.check_attention_decode_fused(be, 1, 2, 64, 5)
    kc.check_attention_decode_fused(be, 2, 4, 128, 4000)   # near the 4096-key limit of the LDS score buffer
    kc.check_attention_decode_fused(be, 24, 32, 128, 1300, per_row=True)   # the decode pool: a position per row
    kc.check_select_embed(be, 8, 32000, 4096)


@pytest.mark.parametrize("N,K,epi,norm,fp8", [(12288, 4096, 0, True, False), (22016, 4096, 3, True, False),
                                              (4096, 11008, 2, False, False), (22016, 4096, 3, True, True),
                                              (27648, 5120, 3, True, True), (15360, 5120, 0, True, True),
                                              (5120, 13824, 2, False, True)])
def test_gemv_rows_agree_between_pool_and_session_passes(be, N, K, epi, norm, fp8):
    """a row gets bit-for-bit the same result from a 29-row pass (the decode pool) as from a <= 16-row pass, bf16 and e4m3
    weights, at the 7b / 13b decoder shapes (W8A16 gate/up: the 4-tiles-per-workgroup geometry)"""
    kc.check_gemv_rows_agree_across_variants(be, N, K, epi, norm, fp8=fp8)


def test_strict_fp32_kernels(be):
    kc.check_gemm_f32(be, 1216, 12288, 4096, 3, bias=False)
    kc.check_gemm_f32(be, 577, 4096, 1024, 1)
    kc.check_gemm_f32(be, 300, 1024, 4096, 4)
    kc.check_gemm_f32(be, 70, 22016, 4096, 5, bias=False)
    kc.check_gemm_f32(be, 576, 1024, 588, 3, bias=False)
    kc.check_attention_f32(be, 2, 16, 577, 64, False)
    kc.check_attention_f32(be, 1, 4, 1216, 128, True)
    kc.check_attention_f32(be, 8, 32, 1, 128, True, decode_pos=1300)
    kc.check_qkv_rope_f32(be, 2, 64, 32, 128, 0)
    kc.check_qkv_rope_f32(be, 8, 1, 32, 128, 1216)


# ---- precision mode "split": the fast kernels with every MFMA operand as bf16 hi + lo, at the true shapes ---------------
@pytest.mark.parametrize("M,N,K,epi,bias,ws", [
    (1216, 12288, 4096, 3, False, 0), (1216, 4096, 4096, 4, False, 0), (1216, 22016, 4096, 5, False, 0),
    (1216, 4096, 11008, 4, False, 0), (9728, 12288, 4096, 3, False, 64), (9728, 4096, 11008, 4, False, 64),
    (1154, 3072, 1024, 3, True, 0), (1154, 4096, 1024, 1, True, 0), (1154, 1024, 4096, 4, True, 0),
    (1152, 1024, 640, 3, False, 0), (1152, 4096, 4096, 2, True, 0), (1216, 32000, 4096, 3, False, 0),
    (70, 264, 192, 0, True, 0), (300, 320, 256, 5, False, 0)])
def test_gemm_split(be, M, N, K, epi, bias, ws):
    """[hi | lo] activation rows against a wrapped bf16 weight (2 MFMAs per weight k-step) vs float64: 3e-5 of max|out|"""
    kc.check_gemm_split(be, M, N, K, epi, bias, ws_mb=ws)


@pytest.mark.parametrize("M,N,K,epi,norm", [(8, 12288, 4096, 1, True), (8, 4096, 4096, 2, False), (8, 22016, 4096, 3, True),
                                            (8, 4096, 11008, 2, False), (8, 32000, 4096, 1, True), (16, 15360, 5120, 1, True),
                                            (16, 5120, 13824, 2, False), (13, 27648, 5120, 3, True), (3, 48, 288, 1, False)])
def test_gemv_split(be, M, N, K, epi, norm):
    kc.check_gemv_split(be, M, N, K, epi, norm)


def test_split_small_and_attention_kernels(be):
    for (N, K, epi) in [(12288, 4096, 1), (22016, 4096, 3), (32000, 4096, 1)]:
        kc.check_gemv_split_groups_agree(be, N, K, epi)
    kc.check_gemv_split(be, 8, 4096, 4096, 1, True, fp8=True)
    kc.check_norm_split(be, 9728, 4096, rms=True)
    kc.check_norm_split(be, 37, 5120, rms=True)
    kc.check_norm_split(be, 1154, 1024, rms=False)
    kc.check_qkv_split32_and_attention_split(be, 2, 8, 1216, 128, True)
    kc.check_qkv_split32_and_attention_split(be, 1, 2, 1216, 128, True, seed=1, spike=True)
    kc.check_qkv_split32_and_attention_split(be, 2, 16, 577, 64, False, rope=False, seed=2)
    kc.check_qkv_split32_and_attention_split(be, 1, 1, 17, 64, False, rope=False, seed=3)
    kc.check_attention_decode_kv32(be, 8, 32, 128, 1216)
    kc.check_attention_decode_kv32(be, 16, 4, 128, 1343, seed=1)
    kc.check_attention_decode_kv32(be, 24, 8, 128, 1300, seed=2)     # the pool's span: two groups of 16
    kc.check_attention_decode_kv32(be, 2, 4, 128, 4000, seed=3)
    kc.check_attention_decode_kv32(be, 1, 2, 64, 5, seed=4)
    # the fp24 caches (the split mode's default KV format): writer + decode step at the true head shapes
    kc.check_kv24(be, 8, 32, 128, 1216, T_prefill=130)
    kc.check_kv24(be, 32, 4, 128, 1343, T_prefill=64, seed=1)
    kc.check_kv24(be, 2, 4, 128, 4000, T_prefill=70, seed=3)
    kc.check_kv24(be, 1, 2, 64, 5, T_prefill=3, seed=4)
    # the e4m3 caches of the fp8 weight format
    kc.check_kv8(be, 8, 40, 128, 1216, T_prefill=130)
    kc.check_kv8(be, 32, 4, 128, 1343, T_prefill=64, seed=1)
    kc.check_kv8(be, 1, 2, 64, 5, T_prefill=3, seed=4)


def test_dma_kernels_are_race_free_and_bit_reproducible(be):
    """The counted-vmcnt schedules (8-phase GEMM, LDS-DMA ring GEMV) order LDS-DMA writes against ds_reads by hand; a
    misplaced wait shows up as rare wrong tiles that depend on timing.  Screen: many back-to-back launches of the true
    shapes (odd k-tile counts, ragged M, both ring geometries) must all produce the bit pattern of the first launch, and
    that pattern must match the oracle (checked by test_gemm / test_gemv on the same shapes)."""
    import numpy as np
    import torch

    rng = np.random.RandomState(3)
    for (M, N, K, epi) in [(1216, 4096, 11008, 0), (9728, 1024, 4096, 3), (2000, 768, 4160, 0)]:
        A, W = be.bf16(kc.bf16_round(rng.randn(M, K))), be.bf16(kc.bf16_round(rng.randn(N, K) * 0.05))
        outs = []
        for it in range(12):
            out = be.zeros((M, N), "f32" if epi == 3 else "bf16")
            be.lib.vck_gemm(be.ptr(A), be.ptr(W), None, be.ptr(out), M, N, K, K, K, N, epi, None)   # no sync in between
            outs.append(out)
        be.sync()
        for o in outs[1:]:
            assert torch.equal(o, outs[0]), f"gemm M{M} N{N} K{K}: launches differ"
    for (M, N, K, epi) in [(8, 12288, 4096, 0), (8, 22016, 4096, 3), (8, 4096, 11008, 1), (16, 5120, 13824, 1), (3, 48, 288, 1)]:
        X = be.bf16(kc.bf16_round(rng.randn(M, K)))
        Wb = be.bf16(kc.bf16_round(rng.randn(N, K) * 0.05))
        Wp = be.zeros((N * K,), "bf16")
        kc._call(be, "vck_pack_weight", Wb, Wp, N, K)
        outs = []
        for it in range(24):
            out = be.zeros((M, N // 2 if epi == 3 else N), "f32" if epi == 1 else "bf16")
            be.lib.vck_gemv(be.ptr(X), be.ptr(Wp), be.ptr(out), M, N, K, N // 2 if epi == 3 else N, epi, None)
            outs.append(out)
        be.sync()
        for o in outs[1:]:
            assert torch.equal(o, outs[0]), f"gemv M{M} N{N} K{K}: launches differ"


@pytest.mark.parametrize("M,N,K,epi,norm,G,ks", [
    (8, 12288, 4096, 1, True, 8, 0), (16, 22016, 4096, 3, True, 16, 0), (32, 12288, 4096, 1, True, 32, 0), (32, 4096, 11008, 2, False, 32, 0),
    (32, 22016, 4096, 3, True, 32, 0), (24, 4096, 4096, 2, False, 24, 0), (29, 32000, 4096, 1, True, 32, 0), (16, 15360, 5120, 1, True, 16, 0),
    (32, 5120, 13824, 2, False, 32, 0), (3, 48, 320, 1, True, 8, 0)])
def test_gemv_wg(be, M, N, K, epi, norm, G, ks):
    """gemv_wg_kernel (workgroup-shared activation chunks; the GEMV of precision mode "split") at the true 7b / 13b decode shapes,
    8..32 rows, every epilogue, both planes in one weight pass (G = 32: 64 operand rows), the default K-slice geometry of each matrix"""
    kc.check_gemv_wg(be, M, N, K, epi, norm, G, ks)


@pytest.mark.parametrize("N,K,epi,G", [(12288, 4096, 1, True), (22016, 4096, 3, True), (4096, 4096, 2, True), (4096, 11008, 2, True),
                                       (32000, 4096, 1, True)])
def test_gemv_wg_rows_agree(be, N, K, epi, G):
    """the pool's promise for the wg form at true shapes: identical bits for a row from 5 / 8 / 13 / 16 / 29-row passes"""
    kc.check_gemv_wg_rows_agree(be, N, K, epi, True, G)


@pytest.mark.parametrize("N,K,epi,rows", [(12288, 4096, 0, (8, 16, 19, 29, 32)), (16 * 767, 4096, 0, (13, 32)), (15360, 5120, 0, (8, 16, 24, 32)),
                                          (22016, 4096, 3, (8, 16, 24, 32)), (27648, 5120, 3, (8, 16, 32))])
def test_gemv_wide_geometry(be, N, K, epi, rows):
    """the one-workgroup-per-CU geometries at the true 7b / 13b shapes (+ a ragged 767-tile matrix): ceil(tiles / 256) tiles per
    workgroup give every row the bits of the pair geometry (measured: profiles/r05_a_kbench_gemv_wide.txt)"""
    kc.check_gemv_wide(be, N, K, epi, rows)


def test_weight_lo_plane_kernels(be):
    """inexact checkpoints (w = bf16 hi + bf16 lo): the loader's plane kernel, the split prefill GEMM's third K segment on the
    8-phase and the 128^2 kernels at true shapes, the strict GEMM, the lo-plane form of the workgroup-shared decode GEMV"""
    kc.check_weight_planes(be, n=100003)
    kc.check_gemm_split_wlo(be, 1216, 12288, 4096, 3, seed=1)       # 7b qkv, one sample's prefill rows: 8-phase kernel, 192 k-tiles
    kc.check_gemm_split_wlo(be, 2432, 4096, 11008, 4, seed=2)       # down: RESID, odd tile count
    kc.check_gemm_split_wlo(be, 1216, 22016, 4096, 5, seed=3)       # gate/up: SwiGLU with stacked hi / lo output
    kc.check_gemm_split_wlo(be, 577, 1024, 1024, 3, seed=4)         # ViT shape (128^2 DMA kernel)
    kc.check_gemm_split_wlo(be, 9728, 12288, 4096, 3, seed=5, ws_mb=64)   # B = 8 prefill rows: split-K remainder round over the three segments
    kc.check_gemm_f32_wlo(be, 70, 512, 1024, 3)
    for (M, N, K, epi, G, ks) in [(8, 12288, 4096, 1, 8, 0), (16, 22016, 4096, 3, 16, 0), (32, 4096, 11008, 2, 32, 0),
                                  (29, 32000, 4096, 1, 32, 0), (24, 4096, 4096, 2, 24, 0)]:
        kc.check_gemv_split_wlo(be, M, N, K, epi, G, ks)


@pytest.mark.parametrize("N,K,epi,norm,rows", [(12288, 4096, 1, False, (8, 29)), (22016, 4096, 3, False, (8, 32)), (5120, 13824, 2, False, (16, 32)),
                                               (22016, 4096, 3, True, (13, 32))])
def test_gemv_wg_six_waves_per_workgroup(be, N, K, epi, norm, rows):
    """the balanced geometry of the split step's GEMV (six tiles per workgroup: 7b qkv 256 workgroups, gate / up 230, 13b down 216)
    against the four-wave geometry at the true shapes: equal bits without the folded norm, the float64 tolerance with it"""
    kc.check_gemv_wg_six_waves(be, N, K, epi, rows, norm)


def test_gemv_wg_is_race_free_and_bit_reproducible(be):
    """hand-placed counted vmcnt waits + one bare barrier per chunk: back-to-back launches must all give the first launch's bits"""
    import numpy as np
    import torch

    be.lib.vck_set_gemv_variant(1)
    try:
        rng = np.random.RandomState(4)
        for (M, N, K, epi) in [(8, 12288, 4096, 0), (32, 22016, 4096, 3), (32, 4096, 11008, 1), (24, 5120, 13824, 1), (29, 32000, 4096, 1)]:
            X = be.bf16(kc.bf16_round(rng.randn(64, K)))   # hi rows [0, M), lo rows [32, 32 + M)
            Wb = be.bf16(kc.bf16_round(rng.randn(N, K) * 0.05))
            Wp = be.zeros((N * K,), "bf16")
            kc._call(be, "vck_pack_weight", Wb, Wp, N, K)
            scratch, counters = be.zeros((8 * (N // 16) * 2 * 256,), "f32"), be.zeros((N // 16 * 2,), "i32")
            outs = []
            for it in range(24):
                out = be.zeros((64, N // 2 if epi == 3 else N), "f32" if epi == 1 else "bf16")
                be.lib.vck_gemv_full(be.ptr(X), be.ptr(Wp), None, be.ptr(out), None, None, None, None, 16, ctypes.c_float(1e-5),
                                     be.ptr(scratch), ctypes.c_ulonglong(8 * (N // 16) * 2 * 256), be.ptr(counters), N // 16 * 2, 0,
                                     M, N, K, N // 2 if epi == 3 else N, epi, 32, None)
                outs.append(out)
            be.sync()
            for o in outs[1:]:
                assert torch.equal(o, outs[0]), f"gemv_wg M{M} N{N} K{K}: launches differ"
    finally:
        be.lib.vck_set_gemv_variant(-1)
