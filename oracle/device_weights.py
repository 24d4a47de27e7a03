"""TEST / CHECKER INFRASTRUCTURE (only tests/ and bench.py's CPU legs import this): the seeded synthetic checkpoint regenerated ON
THE DEVICE (bit-identical to vcoder_amd/synth.py, tests/test_gpu_e2e.py::test_host_weight_load_equals_device_synth) and copied back
for the oracle — the host generator would need ~15 minutes for 6.7 G parameters."""
import ctypes

import numpy as np
import torch

from vcoder_amd import synth


class LazyState(dict):
    """bf16 tensors on the host, widened to fp32 per access (13.5 GB instead of 27 GB for 7b; exact)."""

    def __getitem__(self, k):
        return dict.__getitem__(self, k).float()


DECODER_LINEARS = ("q_proj", "k_proj", "v_proj", "o_proj", "gate_proj", "up_proj", "down_proj")


def effective_fp8_rows(W_bf16: torch.Tensor) -> torch.Tensor:
    """W [N, K] bf16 (any device) -> the EFFECTIVE weights of the fp8 formats, bf16: per output row the power-of-two scale s = 2^e
    with the smallest e such that max|W[n]| <= 448 * 2^e, q = e4m3fn(W[n] / s) (RNE), W_eff = q * s — vcoder_amd/quant.py's rule
    (row_scales' bit arithmetic) with torch's float8_e4m3fn cast, which that module equals byte for byte."""
    Wf = W_bf16.float()
    amax = Wf.abs().amax(dim=1)
    u = amax.view(torch.int32)
    e = (u >> 23) - 127 - torch.where((u & 0x007FFFFF) <= 0x00600000, 8, 7)
    e = torch.where(amax > 0, e, torch.zeros_like(e))
    s = torch.ldexp(torch.ones_like(amax), e)[:, None]
    q = (Wf / s).to(torch.float8_e4m3fn).float()
    eff = (q * s).to(torch.bfloat16)
    assert torch.equal(eff.float(), q * s), "q * 2^e must be exact in bf16"
    return eff


def device_state_dict(eng, cfg, seed, prefixes=None, dtypes="bf16", effective_fp8=False):
    """the seeded checkpoint regenerated on the device and copied back in its own value class: bf16, or (dtypes="reference") fp16 for
    the LLM / projector tensors and fp32 for the CLIP tower — exactly the values vc_model_synth_tensor_rounded loaded.
    effective_fp8: the seven decoder linears per layer replaced by their e4m3-quantised effective values (effective_fp8_rows)."""
    dev = torch.device("cuda:0")
    sd = LazyState()
    for key, shape, off, hw in synth.tensor_specs(cfg):
        if prefixes is not None and not key.startswith(prefixes):
            continue
        if "depth_mm_projector" in key or "mm2_projector" in key or "vcoder_lm_emb" in key:
            continue                      # dead at inference: the oracle never reads them
        n = int(np.prod(shape))
        if dtypes in synth.REFERENCE_CLASSES:
            rounding = synth.reference_rounding(key, dtypes)
            buf = torch.empty(n, dtype=torch.float32, device=dev)
            eng.lib.vck_synth_f32_rounded(ctypes.c_void_p(buf.data_ptr()), ctypes.c_uint64(n), ctypes.c_uint32(synth.tensor_seed(key, seed)),
                                          ctypes.c_float(off), ctypes.c_float(hw), synth.ROUNDING_CODE[rounding], None)
            torch.cuda.synchronize()
            host = buf.cpu()
            if rounding == "fp16":
                h16 = host.to(torch.float16)
                assert torch.equal(h16.float(), host), f"{key}: not fp16-representable"
                host = h16                                  # exact, half the host memory
            dict.__setitem__(sd, key, host.reshape(shape))
            continue
        buf = torch.empty(n, dtype=torch.int16, device=dev)
        eng.lib.vck_synth_bf16(ctypes.c_void_p(buf.data_ptr()), ctypes.c_uint64(n), ctypes.c_uint32(synth.tensor_seed(key, seed)),
                               ctypes.c_float(off), ctypes.c_float(hw), None)
        torch.cuda.synchronize()
        w = buf.view(torch.bfloat16).reshape(shape)
        if effective_fp8 and key.startswith("model.layers.") and any(f".{n}.weight" in key for n in DECODER_LINEARS):
            w = effective_fp8_rows(w)      # what vc_model_set_weight_format(w8a16 / fp8) computes with (quantised at finalize)
        dict.__setitem__(sd, key, w.cpu())
    return sd
