"""CPU tests (emulator) of precision mode "split" (vc_model_set_precision(m, 2)): the FAST kernels with
every MFMA operand carried as bf16 hi + lo.  The bar is BASELINE.json's: logits within 1e-3 of the reference's fp32 CPU path,
greedy ids bit-exact — against the committed fixtures of the live reference (tests/golden)."""
import threading

import numpy as np
import pytest

import e2e_cases
import kernel_cases as kc


@pytest.fixture(scope="module")
def be():
    return kc.EmuBackend()


def test_split_kernels(be):
    # GEMM: K-concatenated [hi | lo] rows, wrapped weight, every epilogue; 128^2 DMA kernel and (M >= 1024) the 8-phase kernel
    for epi in (0, 1, 2, 3, 4, 5):
        kc.check_gemm_split(be, 70, 136, 128, epi, bias=epi != 5, seed=epi)
    kc.check_gemm_split(be, 1030, 520, 192, 3, seed=7)                  # 8-phase 256x256 path, ragged M / N, 6 k-tiles of A
    kc.check_gemm_split(be, 1024, 512, 64, 5, bias=False, seed=8)       # one weight k-tile: wraps after every tile
    # GEMV: stacked hi / lo rows, both group forms, norm folding, residual producer
    for M in (1, 5, 8):
        kc.check_gemv_split(be, M, 64, 256, 1, seed=M)
    kc.check_gemv_split(be, 13, 96, 192, 1, seed=3)
    kc.check_gemv_split(be, 16, 48, 320, 3, seed=4)                     # SwiGLU, stacked hi / lo output
    kc.check_gemv_split(be, 7, 64, 288, 2, norm=False, seed=5)          # odd k-tile count, residual + xg_out + ssq_out
    kc.check_gemv_split(be, 12, 64, 256, 2, norm=False, seed=6)
    kc.check_gemv_split(be, 6, 64, 256, 0, seed=9)
    kc.check_gemv_split(be, 6, 64, 256, 1, seed=10, fp8=True)           # W8A16 weight bytes under split activations
    for epi in (1, 3):
        kc.check_gemv_split_groups_agree(be, 64, 384, epi)
    kc.check_norm_split(be, 9, 256, rms=True)
    kc.check_norm_split(be, 5, 192, rms=False)


def test_split_attention_kernels(be):
    kc.check_qkv_split32_and_attention_split(be, 1, 2, 70, 128, True)
    kc.check_qkv_split32_and_attention_split(be, 2, 1, 130, 64, False, rope=False, seed=1)     # ViT form
    kc.check_qkv_split32_and_attention_split(be, 1, 1, 200, 128, True, seed=2, spike=True)     # late running-max jump
    kc.check_attention_decode_kv32(be, 3, 2, 128, 140)
    kc.check_attention_decode_kv32(be, 10, 1, 64, 70, seed=1)                                  # two groups of G = 16? no: one group, G = 16


def test_fp24_kv_cache_kernels(be):
    """the split mode's KV format (3 bytes per element): writer of the prefill, append + attention of the decode step"""
    kc.check_kv24(be, 3, 2, 128, 140)
    kc.check_kv24(be, 10, 1, 64, 70, T_prefill=33, seed=1)
    kc.check_kv24(be, 19, 1, 128, 90, T_prefill=5, seed=2)    # 17..32 rows: one stacked group of G = 32


@pytest.mark.parametrize("name", ["ds_img_depth_seg", "vc_img_text_seg", "llava_img"])
def test_split_mode_meets_north_star_bar(be, name):
    r = e2e_cases.check_fixture_strict(name, lib=be.lib, mode="split")
    assert r["logits_err"] < 1e-4 and r["decode_logits_err"] < 1e-4, r


@pytest.mark.parametrize("name,mode", [("ds_img_depth_seg", "split"), ("ds_img_depth_seg", "strict"), ("llava_img", "split")])
def test_inexact_checkpoint(be, name, mode):
    """fp16-valued LLM tensors and an fp32-valued CLIP tower (the reference's own checkpoint dtypes; bf16 cannot hold them): the
    weights' lo planes keep strict and split within 1e-3 of the fp32 oracle on the ORIGINAL values, ids bit-exact"""
    r = e2e_cases.check_inexact_checkpoint(name, lib=be.lib, mode=mode)
    assert r["logits_err"] < 1e-4 and r["decode_logits_err"] < 1e-4, r
    # (the record of what the bf16 fast path does to such weights: it rounds them, an error of the order of its activation rounding)
    assert r["bf16_path_logits_err"] < 4e-2 * r["scale"], r
    print(name, mode, r)


def test_lo_plane_bookkeeping_of_the_loader(be):
    """(i) a tensor loaded twice — first from an fp16-valued source (a lo plane is made), then as bf16 bits (a projector /
    override checkpoint saved in bf16) — must not keep the first load's lo values: the strict logits equal those of an engine that
    only ever saw the bf16 tensor; vc_model_inexact_tensors counts tensors, not loads.  (ii) precision mode split on a checkpoint
    with lo planes is REFUSED by vc_model_set_precision when the workgroup-shared GEMV is switched off (it used to throw out of a
    launch)."""
    import torch
    from vcoder_amd import synth
    from vcoder_amd.engine import HipEngine

    g, cfg, ids, imgs, segs, deps = e2e_cases.fixture_inputs("ds_img_depth_seg")
    sd = synth.synth_state_dict(cfg, int(g["seed"]), dtypes="reference")
    key = "model.mm_projector.0.weight"
    w16 = torch.from_numpy(np.ascontiguousarray(sd[key])).to(torch.bfloat16)      # what a bf16 override checkpoint holds
    a = HipEngine(cfg, lib=be.lib)
    a.load_state_dict(sd)
    n0 = a.inexact_tensors()
    a.load_tensor(key, torch.from_numpy(np.ascontiguousarray(sd[key])))            # the same fp32 source again: still one tensor
    assert a.inexact_tensors() == n0
    a.load_tensor(key, w16)                                                        # the bf16 override
    assert a.inexact_tensors() == n0 - 1
    a.finalize()
    b = HipEngine(cfg, lib=be.lib)
    b.load_state_dict({**sd, key: w16})
    b.finalize()
    assert b.inexact_tensors() == n0 - 1
    for e_ in (a, b):
        e_.set_precision("strict")
    la, _, _ = a.prefill(ids, imgs, segs, deps)
    lb, _, _ = b.prefill(ids, imgs, segs, deps)
    assert np.array_equal(la, lb), "a stale lo plane survived the bf16 reload"
    b.close()
    a.set_precision("bf16")
    be.lib.vck_set_gemv_variant(0)
    try:
        with pytest.raises(Exception) as ei:
            a.set_precision("split")
        assert "lo planes" in str(ei.value)
    finally:
        be.lib.vck_set_gemv_variant(-1)
    a.set_precision("split")       # with the default GEMV form the mode is available again
    a.close()


def test_weight_lo_plane_kernels(be):
    kc.check_weight_planes(be)
    for epi in (0, 3, 4, 5):
        kc.check_gemm_split_wlo(be, 70, 136, 128, epi, seed=epi)                 # 128^2 DMA kernel
    kc.check_gemm_split_wlo(be, 1030, 520, 192, 3, seed=7)                       # 8-phase kernel, ragged M / N
    kc.check_gemm_split_wlo(be, 1024, 512, 64, 4, seed=8)                        # one weight k-tile per segment
    kc.check_gemm_split_wlo(be, 1024, 16640 - 8, 64, 3, seed=9, ws_mb=4)         # split-K remainder round over the three segments
    kc.check_gemm_f32_wlo(be, 33, 96, 100, 3)
    for (M, N, K, epi, G, ks) in [(5, 64, 256, 1, 8, 0), (13, 48, 320, 2, 16, 0), (29, 96, 576, 3, 32, 0), (32, 64, 1024, 1, 32, 4),
                                  (21, 80, 384, 0, 24, 2)]:
        kc.check_gemv_split_wlo(be, M, N, K, epi, G, ks)


def _loop_ids(eng, ids, imgs, segs, deps, n_new):
    last, _, _ = eng.prefill(ids, imgs, segs, deps, reserve=n_new)
    toks = [np.argmax(last, -1).astype(np.int32)]
    for _ in range(n_new - 1):
        _, nxt = eng.decode_step(toks[-1], want_logits=False)
        toks.append(nxt)
    return np.stack(toks, 1)


def test_split_pool_equals_session_loop(be):
    """concurrent split-mode generate() calls share the pool's steps (stacked groups of 16) and get the ids of their own
    session loop (groups of 8): the two GEMV forms agree bit for bit"""
    names = ["ds_img_depth_seg", "ds_img_only", "ds_img_seg"]
    root = e2e_cases.engine_for("vcoder_ds", be.lib)
    sessions = [root, root.fork(), root.fork()]
    for s in sessions:
        s.set_precision("split")
    try:
        cases, refs = [], []
        for n in names:
            g, cfg, ids, imgs, segs, deps = e2e_cases.fixture_inputs(n)
            cases.append((ids, imgs, segs, deps))
            refs.append(_loop_ids(root, ids, imgs, segs, deps, 5))
            assert np.array_equal(refs[-1], g["greedy_ids"][:, :5]), "split session loop differs from the reference fixture"
        outs, errs = [[None] * 3 for _ in sessions], []

        def work(si):
            try:
                for j in range(3):
                    ci = (si + j) % 3
                    outs[si][j] = (ci, sessions[si].generate_greedy(*cases[ci], max_new_tokens=5))
            except BaseException as e:
                errs.append(e)

        ths = [threading.Thread(target=work, args=(i,)) for i in range(3)]
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        assert not errs, errs
        for si in range(3):
            for ci, got in outs[si]:
                assert np.array_equal(got, refs[ci]), f"session {si} case {names[ci]}: pooled split ids differ"
    finally:
        for s in sessions:
            s.set_precision("bf16")
        for s in sessions[1:]:
            s.close()
    # back on the bf16 path the pool is rebuilt for it and still serves
    g, cfg, ids, imgs, segs, deps = e2e_cases.fixture_inputs("ds_img_only")
    assert root.generate_greedy(ids, imgs, segs, deps, max_new_tokens=3).shape == (ids.shape[0], 3)


def test_split_batch_above_eight_rows(be):
    """B = 10: the session's split step uses groups of 16 (two MFMA row groups); rows equal the B = 1 results"""
    from vcoder_amd import synth

    g, cfg, ids, imgs, segs, deps = e2e_cases.fixture_inputs("ds_img_only")
    eng = e2e_cases.engine_for(cfg.variant, be.lib)
    eng.set_precision("split")
    try:
        B = 10
        ids_b = np.concatenate([ids[:1]] * B, axis=0)
        pix, _, _ = synth.synth_batch(B, cfg.vit_image_size)
        big = _loop_ids(eng, ids_b, pix, None, None, 4)
        for b in (0, 3, 9):
            one = _loop_ids(eng, ids_b[b:b + 1], pix[b:b + 1], None, None, 4)
            assert np.array_equal(big[b:b + 1], one), f"row {b} of the 10-row split batch differs from its lone run"
    finally:
        eng.set_precision("bf16")
