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
