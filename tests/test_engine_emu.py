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
