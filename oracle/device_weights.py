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


def device_state_dict(eng, cfg, seed, prefixes=None, dtypes="bf16"):
    """the seeded checkpoint regenerated on the device and copied back in its own value class: bf16, or (dtypes="reference") fp16 for
    the LLM / projector tensors and fp32 for the CLIP tower — exactly the values vc_model_synth_tensor_rounded loaded"""
    dev = torch.device("cuda:0")
    sd = LazyState()
    for key, shape, off, hw in synth.tensor_specs(cfg):
        if prefixes is not None and not key.startswith(prefixes):
            continue
        if "depth_mm_projector" in key or "mm2_projector" in key or "vcoder_lm_emb" in key:
            continue                      # dead at inference: the oracle never reads them
        n = int(np.prod(shape))
        if dtypes == "reference":
            rounding = synth.reference_rounding(key)
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
        dict.__setitem__(sd, key, buf.cpu().view(torch.bfloat16).reshape(shape))
    return sd
