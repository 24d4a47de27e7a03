#!/usr/bin/env python
"""How exact is v_mfma_scale_f32_16x16x128_f8f6f4?  e4m3 x e4m3 products are exact in fp32; this prints the deviation of the
W8A8 GEMM (fp32 output through the residual epilogue onto zeros) from the float64 product of the SAME quantised operands."""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from vcoder_amd import _lib, quant, synth  # noqa: E402

lib = _lib.load()
dev = torch.device("cuda:0")
P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None


def run(M, N, K, kind, seed=0):
    rng = np.random.RandomState(seed)
    if kind == "uniform":
        A = rng.randn(M, K)
        W = rng.randn(N, K) * 0.05
    elif kind == "positive":       # no cancellation: every partial sum grows
        A = np.abs(rng.randn(M, K)) + 0.5
        W = np.abs(rng.randn(N, K)) * 0.05 + 0.02
    else:                          # one large product per row next to many small ones
        A = rng.randn(M, K) * 0.01
        A[:, 5] = 400.0
        W = rng.randn(N, K) * 0.05
    A = synth.round_to_bf16(A.astype(np.float32))
    W = synth.round_to_bf16(W.astype(np.float32))
    qa, sa, a_eff = quant.quantize_rows(A)
    qw, sw, w_eff = quant.quantize_rows(W)
    ref = a_eff.astype(np.float64) @ w_eff.T.astype(np.float64)
    Q = torch.from_numpy(qa).to(dev)
    Wr = torch.from_numpy(qw).to(dev)
    sad, swd = torch.from_numpy(sa).to(dev), torch.from_numpy(sw).to(dev)
    out = torch.zeros((M, N), dtype=torch.float32, device=dev)
    lib.vck_gemm_f8(P(Q), P(sad), P(Wr), P(swd), P(out), M, N, K, N, 4, None, C.c_size_t(0), None)
    torch.cuda.synchronize()
    got = out.cpu().numpy().astype(np.float64)
    err = np.abs(got - ref)
    rowmax = np.abs(ref).max(axis=1, keepdims=True) + 1e-30
    # fp32 accumulation in float order for comparison
    ref32 = (a_eff @ w_eff.T).astype(np.float64)
    print(f"{kind:9s} M{M} N{N} K{K}: max|err|/max|ref| {err.max() / np.abs(ref).max():.2e}   max over rows of |err|/rowmax "
          f"{(err / rowmax).max():.2e}   (numpy fp32 matmul: {(np.abs(ref32 - ref) / rowmax).max():.2e})", flush=True)


for kind in ("uniform", "positive", "outlier"):
    for (M, N, K) in [(16, 16, 128), (64, 64, 256), (256, 256, 4096), (300, 272, 11008)]:
        run(M, N, K, kind)
