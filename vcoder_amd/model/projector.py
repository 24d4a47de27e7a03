"""The adapter plugin surface: build_vision_projector / build_seg_projector / build_depth_projector.

Same factory names, config keys and type strings as the reference
(vcoder_llava/model/multimodal_projector/builder.py:33-51, multimodal_adapter/builder.py:31-49,
multimodal_depth_adapter/builder.py:32-50):  'linear' | 'mlp{N}x_gelu' | 'identity'.
The returned module is a weight container + a device forward made of libvcoder_hip GEMMs with fused
bias / erf-GELU epilogues (K9); inside the model the same weights are consumed by vc_encode."""
from __future__ import annotations

import ctypes as C
import re
from typing import Dict, List

import numpy as np


def _depth(projector_type: str) -> int:
    if projector_type == "linear":
        return 1
    if projector_type == "identity":
        return 0
    m = re.match(r"^mlp(\d+)x_gelu$", projector_type)
    if m:
        return int(m.group(1))
    raise ValueError(f"Unknown projector type: {projector_type}")


class HipProjector:
    """Linear | Linear,(GELU,Linear)x(N-1) | identity, state-dict compatible with the reference nn.Module
    ('weight','bias' for linear; '0.weight','0.bias','2.weight',... for mlpNx_gelu)."""

    def __init__(self, projector_type: str, in_features: int, out_features: int, role: str):
        self.projector_type = projector_type
        self.depth = _depth(projector_type)
        self.in_features, self.out_features, self.role = in_features, out_features, role
        self._state: Dict[str, np.ndarray] = {}

    @property
    def config(self):
        return {f"{self.role}_projector_type": self.projector_type}

    def keys(self) -> List[str]:
        if self.depth == 0:
            return []
        if self.depth == 1:
            return ["weight", "bias"]
        out = []
        for j in range(self.depth):
            out += [f"{2 * j}.weight", f"{2 * j}.bias"]
        return out

    def shape_of(self, key: str):
        idx = 0 if self.depth == 1 else int(key.split(".")[0]) // 2
        k = self.in_features if idx == 0 else self.out_features
        return (self.out_features, k) if key.endswith("weight") else (self.out_features,)

    def load_state_dict(self, sd, strict: bool = True):
        for k in self.keys():
            if k in sd:
                v = sd[k]
                v = v.detach().float().cpu().numpy() if hasattr(v, "detach") else np.asarray(v, dtype=np.float32)
                if tuple(v.shape) != self.shape_of(k):
                    raise ValueError(f"size mismatch for {k}: {tuple(v.shape)} vs {self.shape_of(k)}")
                self._state[k] = np.ascontiguousarray(v, dtype=np.float32)
            elif strict:
                raise KeyError(f"Missing key {k} in state_dict")
        return self

    def state_dict(self):
        return dict(self._state)

    def is_loaded(self) -> bool:
        return all(k in self._state for k in self.keys())

    def _device_weights(self, device):
        """bf16 weights / fp32 biases on `device`, uploaded once per loaded state"""
        import torch

        stamp = (str(device), tuple(id(self._state[k]) for k in self.keys()))
        if getattr(self, "_dev_stamp", None) != stamp:
            self._dev = {k: torch.from_numpy(v).to(device, torch.bfloat16 if k.endswith("weight") else torch.float32).contiguous()
                         for k, v in self._state.items()}
            self._dev_stamp = stamp
        return self._dev

    def forward(self, x, lib=None):
        """x: torch CUDA tensor [..., in_features] -> [..., out_features]: the module's own forward on the library's MFMA
        GEMM (bf16 operands, fp32 accumulate, bias / erf-GELU epilogues) — the arithmetic vc_encode applies to the tower
        features inside the model.  The K dimension is zero-padded to the GEMM's 64-element k-tile."""
        import torch
        from .. import _lib

        if self.depth == 0:
            return x
        missing = [k for k in self.keys() if k not in self._state]
        if missing:
            raise RuntimeError(f"{self.role}_projector ({self.projector_type}) has no weights for {missing}: load a state dict "
                               f"(load_pretrained_model / load_state_dict fill the plugin modules)")
        lib = lib if lib is not None else _lib.load()
        dev = self._device_weights(x.device)
        lead = x.shape[:-1]
        cur = x.reshape(-1, x.shape[-1]).to(torch.bfloat16).contiguous()
        M = cur.shape[0]
        for j in range(self.depth):
            wk, bk = ("weight", "bias") if self.depth == 1 else (f"{2 * j}.weight", f"{2 * j}.bias")
            w, b = dev[wk], dev[bk]
            N, K = w.shape
            Kp = (K + 63) // 64 * 64
            if Kp != K:
                w = torch.nn.functional.pad(w, (0, Kp - K)).contiguous()
                cur = torch.nn.functional.pad(cur, (0, Kp - K)).contiguous()
            out = torch.empty((M, N), dtype=torch.bfloat16, device=x.device)
            epi = 0 if j == self.depth - 1 else 2  # EPI_BF16 | EPI_BF16_GELU
            lib.vck_gemm(C.c_void_p(cur.data_ptr()), C.c_void_p(w.data_ptr()), C.c_void_p(b.data_ptr()),
                         C.c_void_p(out.data_ptr()), M, N, Kp, Kp, Kp, N, epi, None)
            if x.is_cuda:
                torch.cuda.synchronize()
            cur = out
        return cur.reshape(*lead, self.out_features).to(x.dtype)

    __call__ = forward


def build_vision_projector(config, delay_load=False, **kwargs) -> HipProjector:
    return HipProjector(getattr(config, "mm_projector_type", "linear"), config.mm_hidden_size, config.hidden_size, "mm")


def build_seg_projector(config, delay_load=False, **kwargs) -> HipProjector:
    return HipProjector(getattr(config, "seg_mm_projector_type", "linear"), config.seg_mm_hidden_size,
                        config.hidden_size, "seg_mm")


def build_depth_projector(config, delay_load=False, **kwargs) -> HipProjector:
    return HipProjector(getattr(config, "depth_mm_projector_type", "linear"), config.depth_mm_hidden_size,
                        config.hidden_size, "depth_mm")
