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


def test_device_preprocessing_matches_pil(emu_lib, tmp_path):
    import preprocess_cases as pc

    eng = e2e_cases.engine_for("vcoder_ds", emu_lib)   # 56 x 56 tower
    pc.check_preprocess(eng, [(56, 56), (40, 100), (120, 75), (200, 200), (30, 31)])
    pc.check_against_hf_processor(eng, tmp_path)


def test_fp8_weight_format(emu_lib):
    """W8A16 decoder weights through the emulator: quantise-at-finalize, byte-streaming GEMV, strict + fast paths."""
    r = e2e_cases.check_fp8_weights("ds_img_only", lib=emu_lib, n_new=4)
    assert r["strict_err"] < 1e-4


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
