"""CPU tests of the HIP kernels' logic through the thread-per-lane emulator (tests/emu): index math,
LDS layouts/swizzles, MFMA fragment plumbing, reductions, masks — at small shapes, against the oracle.
These do NOT prove hardware semantics (that is what `-m gpu` is for); they keep the GPU budget for
measurement instead of debugging."""
import pytest

import kernel_cases as kc


@pytest.fixture(scope="module")
def be():
    return kc.EmuBackend()


@pytest.mark.parametrize("M,N,K,epi,bias", [(128, 128, 64, 3, True), (200, 136, 128, 0, True), (70, 264, 192, 1, True),
                                            (64, 128, 64, 2, True), (100, 128, 128, 4, False), (100, 256, 64, 5, False)])
def test_gemm(be, M, N, K, epi, bias):
    kc.check_gemm(be, M, N, K, epi, bias)


def test_gemm_splitk_remainder_round(be):
    """260 tiles of 256x256 = one full round + 4 remainder tiles, cut into 2 K-slices each and finished by the fix-up
    launch (bias + quick-GELU epilogue; ragged M)."""
    kc.check_gemm(be, 1000 + 24, 16640 - 8, 128, 1, True, seed=2, ws_mb=4)


@pytest.mark.parametrize("M,N,K,epi", [(8, 64, 256, 0), (3, 32, 1024, 1), (16, 48, 288, 2), (8, 64, 256, 3),
                                       (1, 16, 32, 1)])
def test_gemv(be, M, N, K, epi):
    kc.check_gemv(be, M, N, K, epi)


@pytest.mark.parametrize("M,N,K,epi,norm", [(8, 64, 256, 0, False), (3, 32, 1024, 1, False), (16, 48, 320, 2, False),
                                            (8, 64, 256, 3, False), (1, 16, 64, 1, False), (8, 64, 512, 0, True),
                                            (16, 32, 5120, 3, True), (5, 32, 320, 1, True)])
def test_gemv_fp8(be, M, N, K, epi, norm):
    kc.check_gemv_fp8(be, M, N, K, epi, norm)


@pytest.mark.parametrize("M,N,K,epi", [(100, 144, 128, 0), (300, 272, 256, 4), (70, 528, 384, 5), (16, 16, 128, 0)])
def test_gemm_f8(be, M, N, K, epi):
    kc.check_gemm_f8(be, M, N, K, epi)


def test_gemm_f8_splitk_remainder_round(be):
    """one full round of 256 tiles + 4 remainder tiles cut into K-slices; scales applied by the fix-up launch"""
    kc.check_gemm_f8(be, 1000 + 24, 16640 - 16, 256, 0, seed=3, ws_mb=4)


@pytest.mark.parametrize("M,N,K,ks", [(8, 64, 512, 2), (16, 48, 320, 3), (3, 32, 1024, 4), (8, 32, 64, 4)])
def test_gemv_splitk(be, M, N, K, ks):
    kc.check_gemv_splitk(be, M, N, K, ks)


@pytest.mark.parametrize("N,K,epi,norm,ks", [(32, 256, 0, True, 0), (48, 320, 1, True, 0), (64, 256, 3, True, 0),
                                              (32, 512, 2, False, 0), (48, 320, 1, False, 3), (32, 1024, 0, True, 2)])
def test_gemv_32_rows(be, N, K, epi, norm, ks):
    """M in 17..32 (the decode pool): two row groups per weight pass, bit-identical per row to the 16-row pass."""
    kc.check_gemv_rows_agree_across_variants(be, N, K, epi, norm, ks)


@pytest.mark.parametrize("N,K,epi,norm", [(24576, 128, 3, True), (64, 256, 0, True), (4096, 192, 2, False),
                                          (16 * 767, 128, 0, True), (16 * 900, 192, 0, False), (16 * 1376, 128, 3, True), (16 * 1727, 64, 3, True)])
def test_gemv_32_rows_w8a16(be, N, K, epi, norm):
    """the same promise with e4m3 weights; the matrices of more than 512 tiles take the one-workgroup-per-CU geometry at 17..32 rows
    since round 6 (3 / 4 / 6 / 7 tiles per workgroup, ragged last workgroup included), whose finishing stage gives every wave
    several (tile, row group) units — against the pair geometry that serves the same rows in 8- and 16-row passes"""
    kc.check_gemv_rows_agree_across_variants(be, N, K, epi, norm, fp8=True)


def test_gemv_chain_24_rows(be):
    kc.check_gemv_norm_chain(be, 24, 256, 64, seed=7)
    kc.check_gemv_norm_chain(be, 32, 512, 96, seed=8)


def test_interleave(be):
    kc.check_interleave(be, 24, 64)


@pytest.mark.parametrize("rows,D", [(5, 128), (9, 1024)])
def test_layernorm(be, rows, D):
    kc.check_layernorm(be, rows, D)


@pytest.mark.parametrize("rows,D,gather", [(3, 256, False), (6, 4096, False), (2, 5120, True)])
def test_rmsnorm(be, rows, D, gather):
    kc.check_rmsnorm(be, rows, D, gather)


@pytest.mark.parametrize("rows,D", [(9, 256), (5, 1280), (3, 5120)])
def test_rmsnorm_q8(be, rows, D):
    kc.check_rmsnorm_q8(be, rows, D)
    kc.check_quant_act_rows_exhaustive(be)


def test_vit_front(be):
    kc.check_im2col(be, 2, 56, 14, 640)
    kc.check_vit_embed_ln(be, 2, 17, 128)
    kc.check_select_rows(be, 2, 17, 128)


@pytest.mark.parametrize("B,T,H,hd,rope", [(1, 70, 2, 128, True), (2, 17, 2, 64, False)])
def test_qkv_split(be, B, T, H, hd, rope):
    kc.check_qkv_split(be, B, T, H, hd, rope)



@pytest.mark.parametrize("B,H,T,hd,causal,spike", [(1, 1, 17, 64, False, False), (1, 2, 150, 64, False, True),
                                                   (1, 1, 70, 128, True, False), (1, 1, 200, 128, True, True),
                                                   (2, 4, 300, 128, True, True), (4, 2, 150, 64, False, False)])  # B H % 8 == 0: XCD-grouped order
def test_attention(be, B, H, T, hd, causal, spike):
    kc.check_attention(be, B, H, T, hd, causal, spike=spike)


def test_splice_greedy_synth(be):
    kc.check_splice(be, 256)
    kc.check_greedy(be, 3, 320)
    kc.check_synth(be)


def test_e4m3_kv_cache_kernels(be):
    """the fp8 weight format's KV cache (1 byte per element): the prefill's writer and the bf16 step's decode attention"""
    kc.check_kv8(be, 3, 2, 128, 140)
    kc.check_kv8(be, 10, 1, 64, 70, T_prefill=33, seed=1)
    kc.check_kv8(be, 1, 2, 128, 400, T_prefill=33, seed=2)   # past one round of the e4m3 rows (384 keys): the early first batch


def test_fused_decode_kernels(be):
    kc.check_gemv_norm_chain(be, 8, 256, 64)
    kc.check_gemv_norm_chain(be, 3, 512, 96, seed=1)
    kc.check_gemv_norm_chain(be, 16, 5120, 32, seed=3)   # 13b row width, all 16 token slots
    kc.check_gemv_norm_chain(be, 5, 288, 64, seed=5)     # odd k-tile count: the last ring slot holds half a pair
    kc.check_gemv_norm_chain(be, 12, 1024, 64, seed=4)
    kc.check_attention_decode_fused(be, 2, 2, 128, 70)
    kc.check_attention_decode_fused(be, 1, 2, 128, 128)
    kc.check_attention_decode_fused(be, 1, 1, 64, 5)
    kc.check_attention_decode_fused(be, 3, 2, 128, 300, per_row=True)   # a position per row, one row inactive (the pool)
    kc.check_attention_decode_fused(be, 2, 1, 64, 40, per_row=True)
    # the first K batch requested behind phase 0's operand loads (pos >= one round of the 8 waves: 256 keys at hd 128, 512 at 64) and
    # the old order on either side of that boundary
    kc.check_attention_decode_fused(be, 2, 1, 128, 256, seed=2)
    kc.check_attention_decode_fused(be, 1, 1, 128, 255, seed=3)
    kc.check_attention_decode_fused(be, 1, 2, 128, 600, seed=4)
    kc.check_attention_decode_fused(be, 1, 1, 64, 520, seed=5)
    kc.check_attention_decode_kv32(be, 2, 1, 128, 300, seed=6)   # the split step's fp32 / fp24 caches past one round (192 keys)
    kc.check_attention_decode_kv32(be, 1, 1, 128, 100, seed=7)
    kc.check_select_embed(be, 3, 320, 256)


def test_device_sampling(be):
    """temperature / top-k / top-p sampling in the select kernel against the HF warper semantics (small vocabulary)."""
    kc.check_sampling(be, 64, 0.7, 0, 1.0, draws=512)
    kc.check_sampling(be, 64, 0.2, 5, 1.0, draws=256)        # serve/cli.py: temperature 0.2 (+ HF's default top_k)
    kc.check_sampling(be, 96, 1.0, 0, 0.6, draws=512)        # serve/chat.py: top_p
    kc.check_sampling(be, 64, 1.3, 12, 0.8, draws=512)
    kc.check_uniform_extremes(be)


def test_strict_fp32_kernels(be):
    kc.check_gemm_f32(be, 70, 72, 40, 3)
    kc.check_gemm_f32(be, 64, 64, 64, 1)
    kc.check_gemm_f32(be, 33, 128, 100, 4, bias=False)
    kc.check_gemm_f32(be, 20, 64, 32, 5, bias=False)
    kc.check_gemm_f32(be, 5, 16, 588, 2)
    kc.check_attention_f32(be, 1, 2, 17, 64, False)
    kc.check_attention_f32(be, 1, 2, 40, 128, True)
    kc.check_attention_f32(be, 2, 2, 1, 128, True, decode_pos=37)
    kc.check_qkv_rope_f32(be, 2, 5, 2, 128, 0)
    kc.check_qkv_rope_f32(be, 1, 1, 2, 64, 11)


@pytest.mark.parametrize("B,T,H,K,bias,ws,f8,kv8", [(2, 96, 2, 128, False, 0, False, False), (2, 70, 2, 192, True, 0, False, False),
                                                     (3, 133, 4, 128, False, 0, False, True), (1, 64, 2, 256, False, 0, True, True)])
def test_gemm_qkv_fused_epilogue(be, B, T, H, K, bias, ws, f8, kv8):
    kc.check_gemm_qkv_fused(be, B, T, H, K, bias=bias, ws_mb=ws, f8=f8, kv8=kv8)


def test_gemm_qkv_fused_epilogue_splitk_round(be):
    """more than one round of 256 tiles with a short last round: K-slices + the QKV fix-up launch (2 x 640 tokens, H = 22:
    5 x 66 = 330 tiles -> 74 remainder tiles in 3 slices)"""
    kc.check_gemm_qkv_fused(be, 2, 630, 22, 192, ws_mb=64)


def test_alternate_kernel_variants():
    """The non-default template variants stay correct: the same cases in subprocesses with the tuning knobs flipped (the
    library reads them once per process); the sweeps run side by side."""
    import os
    import subprocess
    import sys

    plain_gemm = "test_gemm and not f8 and not splitk"
    sweeps = [
        # register-staged GEMM, 4 x 32 attention
        (dict(VC_GEMM_VARIANT="0", VC_ATTN_VARIANT="1"), plain_gemm + " or test_attention"),
        # 8 x 32 attention
        (dict(VC_ATTN_VARIANT="2"), "test_attention"),
        # tile-order knobs of the GEMM (8-phase kernel forced onto every small, ragged, 1-3 k-tile case; split-K rounds
        # of the bf16 and the e4m3 form)
        (dict(VC_GEMM_VARIANT="5"), "test_gemm"),
        # 2: the one-barrier 256x256 kernel; 4: 256x256 as 4 waves x (128 x 128)
        (dict(VC_GEMM_VARIANT="2"), plain_gemm),
        (dict(VC_GEMM_VARIANT="4"), plain_gemm),
        # 7: the 8-phase kernel on v_mfma_f32_32x32x16_bf16 for every size (round 6): ragged tiles, all epilogues, split-K rounds,
        # and the split mode's K-wrapped / lo-plane forms
        # (the fused-QKV check compares two launches bit for bit, and its reference GEMM would run on the other MFMA shape, whose
        # 16-wide k-steps round differently: excluded here)
        (dict(VC_GEMM_VARIANT="7"), "(" + plain_gemm + " or (test_gemm and splitk and not f8) or test_gemm_split or lo_plane) and not qkv_fused"),
    ]
    procs = []
    for knobs, sel in sweeps:
        env = dict(os.environ, **knobs)
        procs.append((knobs, subprocess.Popen([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-x", "-k",
                                               f"({sel}) and not alternate"], env=env, stdout=subprocess.PIPE,
                                              stderr=subprocess.STDOUT, text=True)))
    for (knobs, pr), (_, sel) in zip(procs, sweeps):
        out, _ = pr.communicate()
        if pr.returncode != 0:
            # seven pytest processes side by side oversubscribe a small host; a sweep that failed there is repeated alone
            # and must pass on its own — the kernels are deterministic, so a real defect fails both times
            r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-x", "-k",
                                f"({sel}) and not alternate"], env=dict(os.environ, **knobs), capture_output=True, text=True)
            assert r.returncode == 0, f"{knobs}: first run:\n{out[-1500:]}\nrepeated alone:\n{r.stdout[-1500:]}"


@pytest.mark.parametrize("M,N,K,epi,norm,G,ks", [
    (5, 64, 256, 0, True, 8, 0), (13, 48, 320, 2, True, 16, 0), (29, 96, 576, 3, True, 32, 0), (32, 64, 1024, 1, True, 32, 4),
    (21, 80, 384, 2, True, 24, 2), (8, 64, 256, 3, True, 8, 0), (1, 16, 64, 1, False, 8, 0), (16, 48, 320, 2, False, 16, 0),
    (24, 64, 512, 0, False, 24, 3)])
def test_gemv_wg(be, M, N, K, epi, norm, G, ks):
    """the workgroup-shared-activation decode GEMV of precision mode "split": all row counts (1..4 activation pieces), epilogues,
    both planes in one weight pass, K-slices over workgroups"""
    kc.check_gemv_wg(be, M, N, K, epi, norm, G, ks)


@pytest.mark.parametrize("N,K,epi,norm,rows", [(16 * 700, 192, 0, True, (5, 13, 19, 32)), (16 * 767, 128, 0, True, (8, 29)),
                                               (16 * 900, 128, 0, False, (8, 16, 19, 32)), (16 * 1376, 128, 3, True, (3, 16, 24, 29)),
                                               (16 * 1727, 64, 3, True, (8, 9, 17, 32))])
def test_gemv_wide_geometry(be, N, K, epi, norm, rows):
    """the one-workgroup-per-CU geometries (3 / 4 / 6 / 7 tiles per workgroup) give the pair geometry's bits"""
    kc.check_gemv_wide(be, N, K, epi, rows, norm)


@pytest.mark.parametrize("N,K,epi,norm,rows,ks", [(16 * 13, 576, 0, False, (5, 16, 19, 32), 0), (16 * 13, 576, 2, False, (8, 29), 2),
                                                  (16 * 7, 1024, 1, False, (13, 32), 4), (16 * 12, 320, 3, False, (8, 24), 0),
                                                  (16 * 13, 576, 2, True, (8, 32), 3), (16 * 6, 256, 0, True, (16, 17), 0)])
def test_gemv_wg_six_waves_per_workgroup(be, N, K, epi, norm, rows, ks):
    """six tiles per workgroup (the balanced geometry of 7b qkv / gate-up and 13b o / down in precision mode split): the four-wave
    geometry's bits without the folded norm, its values within the float64 tolerance with it; ragged last workgroup, K-slices, every epilogue"""
    kc.check_gemv_wg_six_waves(be, N, K, epi, rows, norm, ks)


def test_gemv_wg_chooses_six_waves_for_unbalanced_launches(be):
    """1376 tiles (7b gate / up): 344 four-wave workgroups would leave 88 CUs with two — the launcher takes six waves (230
    workgroups); the result against float64 as for every other geometry"""
    kc.check_gemv_wg(be, 29, 16 * 1376, 128, 3, True, 32, 0)
    kc.check_gemv_wg(be, 8, 16 * 1376, 128, 3, False, 8, 0, seed=1)


@pytest.mark.parametrize("N,K,epi,G,ks", [(64, 512, 0, True, 0), (48, 1024, 1, True, 4), (64, 256, 3, True, 0), (32, 512, 2, True, 2)])
def test_gemv_wg_rows_agree(be, N, K, epi, G, ks):
    kc.check_gemv_wg_rows_agree(be, N, K, epi, True, G, ks)
