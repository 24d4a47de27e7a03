"""CPU end-to-end test of the host engine + kernels through the emulator (tests/emu) on ONE small fixture.
Covers weight placement, the splice planner, KV-cache handling, graph replay bookkeeping and the greedy
loop before any GPU time is spent; all other fixtures run under `-m gpu`."""
import ctypes
import os

import pytest

import e2e_cases
import kernel_cases as kc


@pytest.fixture(scope="module")
def emu_lib():
    return kc.EmuBackend().lib


def test_ds_reference_order_fixture(emu_lib):
    r = e2e_cases.check_fixture("ds_img_only", lib=emu_lib, check_generate=False, check_emu_oracle=False)
    assert r["logits_err_vs_ref"] < e2e_cases.TOL_VS_FP32_REF


@pytest.mark.parametrize("name", ["ds_img_depth_seg", "vc_img_text_seg", "llava_img"])
def test_strict_mode_meets_north_star_bar(emu_lib, name):
    """Strict (fp32) mode through the same engine: logits within 1e-3 of the real reference, ids bit-exact."""
    r = e2e_cases.check_fixture_strict(name, lib=emu_lib)
    assert r["logits_err"] < 1e-4 and r["decode_logits_err"] < 1e-4


@pytest.mark.parametrize("fmt", ["bf16", "fp8"])
def test_qkv_fused_epilogue_through_the_engine(emu_lib, fmt):
    """Round 6: the prefill with RoPE + head split + KV write inside the QKV GEMM's epilogue gives the bits of the
    separate GEMM + qkv_split launches — all-position logits, the KV cache the decode steps then read (teacher-forced steps),
    generate() — for a spliced length that is not a multiple of 32 (padded token rows), bf16 and the fp8 format (e4m3 cache rows)."""
    import numpy as np
    from vcoder_amd.engine import HipEngine

    g, cfg, ids, imgs, segs, deps = e2e_cases.fixture_inputs("ds_img_depth_seg")
    eng = HipEngine(cfg, lib=emu_lib)
    eng.load_synthetic(int(g["seed"]))
    if fmt != "bf16":
        eng.set_weight_format(fmt)
    eng.finalize()
    outs = []
    for on in (0, 2):
        eng.set_qkv_fused(on)
        last, full, S = eng.prefill(ids, imgs, segs, deps, all_logits=True)
        assert S % 32 != 0
        tok = np.argmax(last, -1).astype(np.int32)
        steps = [eng.decode_step(tok)[0] for _ in range(3)]
        outs.append((full, np.stack(steps), eng.generate_greedy(ids, imgs, segs, deps, max_new_tokens=6)))
    for a, b in zip(outs[0], outs[1]):
        assert np.array_equal(a, b)
    eng.close()


def test_fp16_operand_library_fixture():
    """Round 6: the kernels built with -DVC_F16 (IEEE fp16 MFMA operands — the precision of the reference's own GPU path — on the
    emulator's -DVC_F16 build): logits of a live-reference fixture at all positions and every cached step within 1.2e-3 of
    |logit|max (measured 4.7e-4; the bf16 build: 4e-3), greedy ids equal, generate() == the step loop."""
    be16 = kc.EmuBackend("fp16")
    assert be16.lib.vc_operand_format() == 1
    r = e2e_cases.check_fixture_fp16("ds_img_depth_seg", lib=be16.lib)
    assert r["ids_equal"] and r["logits_rel_err_vs_ref"] < 1e-3


def test_other_projector_types_fixture(emu_lib):
    """'linear' <image> adapter + 'mlp3x_gelu' <seg> / <depth> adapter, fixture of the live reference (round 3): fast path
    within the bf16 tolerance, strict and split modes within 1e-3 with bit-exact ids ('identity': under -m gpu)."""
    r = e2e_cases.check_fixture("ds_proj_linear_mlp3x", lib=emu_lib, check_generate=True, check_emu_oracle=False)
    assert r["ids_equal"]
    for mode in ("strict", "split"):
        r = e2e_cases.check_fixture_strict("ds_proj_linear_mlp3x", lib=emu_lib, mode=mode)
        assert r["logits_err"] < 1e-4


def test_device_preprocessing_matches_pil(emu_lib, tmp_path):
    import preprocess_cases as pc

    eng = e2e_cases.engine_for("vcoder_ds", emu_lib)   # 56 x 56 tower
    pc.check_preprocess(eng, [(56, 56), (40, 100), (120, 75), (200, 200), (30, 31)])
    pc.check_against_hf_processor(eng, tmp_path)


@pytest.mark.parametrize("fmt", ["w8a16", "fp8"])
def test_fp8_weight_format(emu_lib, fmt):
    """W8A16 decoder weights through the emulator: quantise-at-finalize, byte-streaming GEMV, strict + fast paths;
    'fp8' adds the W8A8 prefill (activation rows quantised per token, e4m3 x e4m3 GEMM)."""
    r = e2e_cases.check_fp8_weights("ds_img_only", lib=emu_lib, n_new=4, fmt=fmt)
    print(fmt, r)
    assert r["strict_err"] < 1e-4


@pytest.mark.parametrize("fmt", ["bf16", "w8a16", "fp8"])
def test_layers_teacher_forced(emu_lib, fmt):
    """vc_debug_prefill_layers: each decoder layer on the oracle's own input (tiny model; the 13b geometry runs under -m gpu)"""
    from vcoder_amd import config as vcfg

    r = e2e_cases.check_layers_teacher_forced(vcfg.tiny("vcoder_ds"), 21, fmt, layers=(0, 1), B=2, S=70, lib=emu_lib)
    print(fmt, r)


@pytest.mark.parametrize("fmt", ["w8a16", "fp8"])
def test_layers_teacher_forced_with_massive_activation_channels(emu_lib, fmt):
    """the same with three hidden channels of the residual stream 300 x the rest (what trained LLaMA-family checkpoints carry):
    a token row's e4m3 scale is then set by channels without information; device vs the oracle quantising the same rows, relative
    to the layer's update"""
    from vcoder_amd import config as vcfg

    r = e2e_cases.check_layers_teacher_forced(vcfg.tiny("vcoder_ds"), 23, fmt, layers=(0, 1), B=2, S=70, lib=emu_lib, outlier_gain=300.0)
    print(fmt, r)


@pytest.mark.parametrize("mode", ["bf16", "strict", "split"])
def test_padded_batch_attention_mask(emu_lib, mode):
    """a 2-D attention_mask that hides positions (right- and left-padded rows) against the live reference's fixture: masked
    prefill, both cached-step forms, generate()"""
    r = e2e_cases.check_masked_fixture(lib=emu_lib, mode=mode)
    print(mode, r)


@pytest.mark.parametrize("mode", ["bf16", "strict", "split"])
def test_output_hidden_states(emu_lib, mode):
    print(mode, e2e_cases.check_hidden_states(lib=emu_lib, mode=mode))


def test_device_side_stop_sequences(emu_lib):
    e2e_cases.check_stop_sequences("ds_img_only", lib=emu_lib)


def test_generate_splits_batches_larger_than_a_replica(emu_lib):
    """B = 18 > 16: generate runs two pieces; rows equal the rows of smaller calls, and unequal spliced lengths across the
    pieces raise like they do inside one batch (quirk 6)."""
    import numpy as np
    import pytest as _pt
    from vcoder_amd import synth

    g, cfg, ids, imgs, segs, deps = e2e_cases.fixture_inputs("ds_img_only")
    eng = e2e_cases.engine_for(cfg.variant, emu_lib)
    B = 18
    big_ids = np.concatenate([ids] * (B // ids.shape[0]), axis=0)
    pix, _, _ = synth.synth_batch(B, cfg.vit_image_size)
    out = eng.generate_greedy(big_ids, pix, None, None, max_new_tokens=3)
    assert out.shape == (B, 3)
    tail = eng.generate_greedy(big_ids[16:], pix[16:], None, None, max_new_tokens=3)
    head = eng.generate_greedy(big_ids[:4], pix[:4], None, None, max_new_tokens=3)
    assert np.array_equal(out[16:], tail) and np.array_equal(out[:4], head)
    # second piece with a different spliced length: its rows carry no <seg> placeholder (a text token instead)
    g2, cfg2, ids2, imgs2, segs2, _ = e2e_cases.fixture_inputs("ds_img_seg")
    big2 = np.concatenate([ids2] * (B // ids2.shape[0]), axis=0)
    pix2, seg2, _ = synth.synth_batch(B, cfg2.vit_image_size)
    mixed = big2.copy()
    mixed[16:][mixed[16:] == -300] = 5
    same = eng.generate_greedy(big2, pix2, seg2, None, max_new_tokens=2)
    assert same.shape == (B, 2)
    with _pt.raises(UnboundLocalError):
        eng.generate_greedy(mixed, pix2, seg2, None, max_new_tokens=2)


def test_list_and_5d_image_inputs(emu_lib):
    """Several images per sample (list / 5-D form) against the live reference's fixture, through the engine."""
    r = e2e_cases.check_list_fixture("ds_list_uneven", lib=emu_lib)
    assert r["logits_err_vs_ref"] < e2e_cases.TOL_VS_FP32_REF


def test_vision_tower_boundary(emu_lib):
    """vc_vision_tower_forward (CLIPVisionTower.forward + feature_select) vs the live reference's tower output."""
    assert e2e_cases.check_tower_fixture(lib=emu_lib, strict=True) < 1e-4
    e2e_cases.check_tower_fixture(lib=emu_lib)


def test_kv_cache_grows_under_a_long_decode_loop(emu_lib):
    """A host-driven decode_step loop far past the prefill's reserve (what the reference CLI's max_new_tokens=512 host loop
    does): the cache grows and the tokens equal those of a run that reserved the space up front."""
    import numpy as np

    g, cfg, ids, imgs, segs, deps = e2e_cases.fixture_inputs("ds_img_only")
    eng = e2e_cases.engine_for(cfg.variant, emu_lib)
    n = 112                                                                     # S + n crosses the 64- and the 128-key capacities
    ref = eng.generate_greedy(ids, imgs, segs, deps, max_new_tokens=n + 1)      # sized for n up front
    last, _, S = eng.prefill(ids, imgs, segs, deps, reserve=8)                   # tiny reserve: forces two growths
    tok = np.argmax(last, -1).astype(np.int32)
    toks = [tok]
    for _ in range(n):
        _, tok = eng.decode_step(tok, want_logits=False)
        toks.append(tok)
    assert np.array_equal(np.stack(toks, 1), ref)
    eng.prefill(ids, imgs, segs, deps, reserve=64)


def test_load_tensor_checks_the_full_shape(emu_lib):
    """A transposed weight has the right element count and the wrong meaning: refused."""
    import numpy as np
    import pytest as _pt
    from vcoder_amd import config as vcfg
    from vcoder_amd.engine import HipEngine

    cfg = vcfg.tiny("vcoder_ds")
    eng = HipEngine(cfg, lib=emu_lib)
    F, D = cfg.intermediate_size, cfg.hidden_size
    assert eng.load_tensor("model.layers.0.mlp.down_proj.weight", np.zeros((D, F), np.float32))
    with _pt.raises(ValueError, match="expected shape"):
        eng.load_tensor("model.layers.0.mlp.down_proj.weight", np.zeros((F, D), np.float32))
    # the conv patch embedding may come as [Dv,3,P,P] or flattened
    Dv, P = cfg.mm_hidden_size, cfg.vit_patch_size
    assert eng.load_tensor("vision_model.embeddings.patch_embedding.weight", np.zeros((Dv, 3, P, P), np.float32))
    assert eng.load_tensor("vision_model.embeddings.patch_embedding.weight", np.zeros((Dv, 3 * P * P), np.float32))
    eng.close()


def test_generate_rejects_bad_pad_and_depth_index(emu_lib):
    import numpy as np
    import pytest as _pt
    from vcoder_amd import synth

    g, cfg, ids, imgs, segs, deps = e2e_cases.fixture_inputs("ds_img_seg_depth")
    eng = e2e_cases.engine_for(cfg.variant, emu_lib)
    with _pt.raises(IndexError, match="pad_token_id"):
        eng.generate_greedy(ids, imgs, segs, deps, max_new_tokens=2, eos_token_id=2, pad_token_id=cfg.vocab_size + 5)
    # two <depth> placeholders in row 0 advance the depth index twice: row 1 then indexes past the list (reference: IndexError)
    bad = np.concatenate([ids, np.full((ids.shape[0], 1), 9)], axis=1)
    bad[0, -1] = synth.DEPTH_TOKEN_INDEX
    with _pt.raises(IndexError):
        eng.prefill(bad, imgs, segs, deps)


def test_full_depth_parity_logic_on_the_tiny_model(emu_lib):
    """The teacher-forced full-depth parity check of tests/test_gpu_fulldepth.py, run here on the tiny architecture through
    the emulator so that its logic is validated before it meets the 7b / 13b models on the GPU box."""
    import test_gpu_fulldepth as fd
    from vcoder_amd import config as vcfg

    r = fd.run_case(vcfg.tiny("vcoder_ds"), B=3, n_new=6, seed=42, oracle_rows=(0, 2), checkpoints=(1, 2), strict_tokens=3,
                    pooled_calls=2, lib=emu_lib)
    assert r["e_strict"] < 1e-4 and r["e_split"] < 1e-4 and r["err32"].max() < e2e_cases.TOL_VS_FP32_REF
    r = fd.run_case(vcfg.tiny("vcoder_ds"), B=2, n_new=4, seed=42, oracle_rows=(0, 1), checkpoints=(2,), strict_tokens=2,
                    split=False, pooled_calls=1, lib=emu_lib)
    assert r["e_strict"] < 1e-4 and r["err32"].max() < e2e_cases.TOL_VS_FP32_REF
    # the form the GPU cases use: the bf16 path measured against the split path forced with its ids
    r = fd.run_case(vcfg.tiny("vcoder_ds"), B=3, n_new=5, seed=42, oracle_rows=(0, 2), checkpoints=(2,), strict_tokens=2,
                    pooled_calls=2, lib=emu_lib, fast_vs="split")
    assert r["e_strict"] < 1e-4 and r["e_split"] < 1e-4 and r["err32"].max() < e2e_cases.TOL_VS_FP32_REF
    # a checkpoint with the reference's value classes, generated on the "device" (vc_model_synth_tensor_rounded): the lo planes
    r = fd.run_case(vcfg.tiny("vcoder_ds"), B=2, n_new=4, seed=43, oracle_rows=(1,), checkpoints=(), strict_tokens=2,
                    pooled_calls=1, lib=emu_lib, fast_vs="split", dtypes="reference")
    assert r["e_strict"] < 1e-4 and r["e_split"] < 1e-4
    # ... and with the values the reference COMPUTES with: model/builder.py:142 casts the loaded tower to fp16 ("reference_loaded",
    # the class the full-depth GPU case runs)
    r = fd.run_case(vcfg.tiny("vcoder_ds"), B=2, n_new=4, seed=43, oracle_rows=(1,), checkpoints=(), strict_tokens=2,
                    pooled_calls=1, lib=emu_lib, fast_vs="split", dtypes="reference_loaded")
    assert r["e_strict"] < 1e-4 and r["e_split"] < 1e-4


def test_context_limit_is_a_clean_error(emu_lib):
    """Maximum sizes: a sequence may fill the context (max_position_embeddings) to the last slot and then generation stops with
    an error that names the limit — not a write past the KV cache; a prompt whose spliced length exceeds it is refused up front.
    (HF's LlamaRotaryEmbedding re-sizes its table instead; the engine's cap is documented in include/vcoder_hip.h.)"""
    import numpy as np
    import pytest as _pt

    g, cfg, ids, imgs, segs, deps = e2e_cases.fixture_inputs("ds_img_only")
    ref_eng = e2e_cases.engine_for(cfg.variant, emu_lib)
    eng = e2e_cases.engine_for(cfg.variant, emu_lib, overrides={"max_position_embeddings": 64})
    _, _, S = eng.prefill(ids, imgs, segs, deps)
    assert S < 64
    n_fit = 64 - S              # prompt + new tokens <= max_position_embeddings
    want = ref_eng.generate_greedy(ids, imgs, segs, deps, max_new_tokens=n_fit)
    got = eng.generate_greedy(ids, imgs, segs, deps, max_new_tokens=n_fit)
    assert np.array_equal(got, want)
    with _pt.raises((RuntimeError, ValueError), match="max_position_embeddings"):
        eng.generate_greedy(ids, imgs, segs, deps, max_new_tokens=n_fit + 8)
    # a host-driven decode_step loop runs into the same wall
    last, _, S = eng.prefill(ids, imgs, segs, deps, reserve=4)
    tok = np.argmax(last, -1).astype(np.int32)
    for _ in range(64 - S):     # the step that caches position 63 is the last one that fits
        _, tok = eng.decode_step(tok, want_logits=False)
    with _pt.raises((RuntimeError, ValueError), match="max_position_embeddings"):
        eng.decode_step(tok, want_logits=False)
    # the engine is usable afterwards
    assert np.array_equal(eng.generate_greedy(ids, imgs, segs, deps, max_new_tokens=4), want[:, :4])
    long_ids = np.concatenate([ids, np.full((ids.shape[0], 64), 5, dtype=ids.dtype)], axis=1)
    with _pt.raises((RuntimeError, ValueError), match="max_position_embeddings"):
        eng.prefill(long_ids, imgs, segs, deps)


def test_empty_inputs_are_refused(emu_lib):
    """Empty batches, empty prompts and max_new_tokens = 0 are errors (HF's generate validates max_new_tokens > 0 the same way),
    not launches over zero rows."""
    import pytest as _pt

    g, cfg, ids, imgs, segs, deps = e2e_cases.fixture_inputs("ds_img_only")
    eng = e2e_cases.engine_for(cfg.variant, emu_lib)
    with _pt.raises(ValueError):
        eng.generate_greedy(ids, imgs, segs, deps, max_new_tokens=0)
    with _pt.raises(ValueError):
        eng.generate_greedy(ids[:0], imgs[:0], None, None, max_new_tokens=2)
    with _pt.raises(ValueError):
        eng.prefill(ids[:, :0], imgs, segs, deps)
    assert eng.generate_greedy(ids, imgs, segs, deps, max_new_tokens=1).shape == (ids.shape[0], 1)


@pytest.mark.parametrize("mode", ["bf16", "split"])
def test_prefill_with_folded_rmsnorm(emu_lib, mode, monkeypatch):
    """VC_PREFILL_FOLD=1 (opt-in): the residual-writing GEMMs of a prefill hand xg = bf16(x * g) and
    sum-of-squares partials to the next GEMM, which scales its accumulator by 1/rms — no RMSNorm pass behind layer 0.  The logits
    stay within the mode's tolerance of the reference fixture (split: 1e-3 absolute, ids exact) and of the unfolded prefill."""
    import numpy as np

    g, cfg, ids, imgs, segs, deps = e2e_cases.fixture_inputs("ds_img_depth_seg")
    eng = e2e_cases.engine_for("vcoder_ds", emu_lib)
    if mode != "bf16":
        eng.set_precision(mode)
    base, base_all, _ = eng.prefill(ids, imgs, segs, deps, all_logits=True)
    monkeypatch.setenv("VC_PREFILL_FOLD", "1")
    fold, fold_all, _ = eng.prefill(ids, imgs, segs, deps, all_logits=True)
    monkeypatch.delenv("VC_PREFILL_FOLD")
    assert not np.array_equal(fold_all, base_all)   # the other rounding order really ran
    ref = g["prefill_logits"]
    tol = 1e-3 if mode == "split" else e2e_cases.TOL_VS_FP32_REF * float(np.abs(ref).max())
    assert np.abs(fold_all - ref).max() < tol and np.abs(base_all - ref).max() < tol
    assert np.array_equal(fold.argmax(-1), g["greedy_ids"][:, 0])
    eng.set_precision("bf16")   # (engine_for caches its engines)
