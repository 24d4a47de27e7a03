"""W8A16 weight format of the decode GEMV (BASELINE.json configs[4], "fp8 weights") — host restatement.

The device quantiser (`quantize_fp8_kernel`, csrc/decode.hip) runs at `vc_model_finalize`; this module states the same
arithmetic in numpy so checkpoints can be inspected / pre-quantised on the host and so the tests can hand the CPU
reference restatement exactly the effective weights the GPU computes with:

  per output row n of W [N, K]:   s_n = 2^e, e = the smallest integer with max|W[n]| <= 448 * 2^e
                                  q   = e4m3fn( W[n] / s_n )      round-to-nearest-even, saturating at +-448
                                  W_eff[n] = q * s_n               (exactly representable in bf16)

The reference's own reduced-precision switch is `load_8bit` (`vcoder_llava/model/builder.py:31-33`, bitsandbytes
LLM.int8); fp8-e4m3 with a power-of-two row scale is the MI355X-native counterpart: the scale is exact in every
format on the path, so prefill (bf16 MFMA over W_eff) and decode (bytes widened in registers, scale applied to the
fp32 accumulator) see identical weights.
"""
from __future__ import annotations

import numpy as np

E4M3_MAX = 448.0


def e4m3_encode(x: np.ndarray) -> np.ndarray:
    """float32 -> OCP e4m3fn bytes (RNE, saturating).  Mirrors `f2fp8` in csrc/vc_device.h."""
    x = np.asarray(x, dtype=np.float32)
    u = x.view(np.uint32)
    sign = ((u >> 24) & 0x80).astype(np.uint32)
    a = np.abs(x)
    sub = np.rint(np.minimum(a, np.float32(1.0)) * np.float32(512.0)).astype(np.uint32)
    e = ((u >> 23) & 0xFF).astype(np.int32) - 127
    m = ((u & 0x007FFFFF) | 0x3F800000).astype(np.uint32).view(np.float32)
    code = ((np.clip(e, -6, 8) + 7).astype(np.uint32) << 3) + np.rint((m - np.float32(1.0)) * np.float32(8.0)).astype(np.uint32)
    code = np.minimum(code, 0x7E)
    out = np.where(a < np.float32(0.015625), sub, code)
    out = np.where(a < np.float32(448.0), out, 0x7E)
    return (sign | out).astype(np.uint8)


def e4m3_decode(b: np.ndarray) -> np.ndarray:
    b = np.asarray(b, dtype=np.uint8).astype(np.uint32)
    e = (b >> 3) & 15
    m = b & 7
    normal = (((e + 120) << 23) | (m << 20)).astype(np.uint32).view(np.float32)
    mag = np.where(e == 0, m.astype(np.float32) * np.float32(2.0 ** -9), normal)
    return np.where(b & 0x80, -mag, mag).astype(np.float32)


def row_scales(W: np.ndarray) -> np.ndarray:
    """Per-row power-of-two scale: the smallest 2^e with max|row| <= 448 * 2^e (1.0 for an all-zero row)."""
    amax = np.abs(np.asarray(W, dtype=np.float32)).max(axis=1)
    u = amax.view(np.uint32)
    e = (u >> 23).astype(np.int32) - 127 - np.where((u & 0x007FFFFF) <= 0x00600000, 8, 7)
    e = np.where(amax > 0, e, 0)
    return np.ldexp(np.float32(1.0), e).astype(np.float32)


def quantize_rows(W: np.ndarray):
    """W [N, K] (bf16-valued float32) -> (q uint8 [N, K], scale float32 [N], W_eff float32 [N, K])."""
    W = np.asarray(W, dtype=np.float32)
    s = row_scales(W)
    q = e4m3_encode(W / s[:, None])
    return q, s, e4m3_decode(q) * s[:, None]


def pack_supertiles(q: np.ndarray) -> np.ndarray:
    """q [N, K] bytes -> the decode GEMV's order [N/16][K/64][64 lanes][16 B], lane = n%16 + 16*((k%64)/16)."""
    N, K = q.shape
    t = q.reshape(N // 16, 16, K // 64, 4, 16)  # [nt][n%16][st][g][byte]
    return np.ascontiguousarray(t.transpose(0, 2, 3, 1, 4)).reshape(-1)


# decoder state-dict keys that the W8A16 format quantises (HF names; q/k/v are quantised row-wise, so the fused
# [3D, D] device matrix equals the three matrices quantised separately — likewise gate/up)
QUANTIZED_SUFFIXES = ("self_attn.q_proj.weight", "self_attn.k_proj.weight", "self_attn.v_proj.weight",
                      "self_attn.o_proj.weight", "mlp.gate_proj.weight", "mlp.up_proj.weight", "mlp.down_proj.weight")


def effective_state_dict(sd: dict) -> dict:
    """The state dict the device computes with under W8A16: decoder linears replaced by their dequantised values."""
    out = {}
    for k, v in sd.items():
        if k.startswith("model.layers.") and k.endswith(QUANTIZED_SUFFIXES):
            arr = v.detach().float().numpy() if hasattr(v, "detach") else np.asarray(v, dtype=np.float32)
            w_eff = quantize_rows(arr)[2]
            if hasattr(v, "detach"):
                import torch

                out[k] = torch.from_numpy(w_eff).to(v.dtype)
            else:
                out[k] = w_eff
        else:
            out[k] = v
    return out
