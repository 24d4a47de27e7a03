#!/usr/bin/env python
"""Yardstick for the prefill GEMM: the vendor library (torch.matmul -> hipBLASLt / rocBLAS, bf16 in, fp32 accumulate) against
this repository's 8-phase MFMA GEMM (vck_gemm) on the SAME shapes and random operands — the decoder linears of the benchmarked
batch (M = 8 x 1216) and the ViT linears.  Says how much of the gap to the 2.5 PFLOP/s dense peak is specific to our kernel and
how much any bf16 GEMM pays on this part with random (bit-toggling) data.  usage: python tools/experiments/gemm_yardstick.py"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from vcoder_amd import _lib  # noqa: E402

lib = _lib.load()
dev = torch.device("cuda:0")
SHAPES = [("7b qkv", 9728, 12288, 4096), ("7b o", 9728, 4096, 4096), ("7b gate/up", 9728, 22016, 4096), ("7b down", 9728, 4096, 11008),
          ("13b qkv", 19456, 15360, 5120), ("13b gate/up", 19456, 27648, 5120), ("vit qkv", 13848, 3072, 1024),
          ("vit fc1", 13848, 4096, 1024), ("vit fc2", 13848, 1024, 4096)]


def timed(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3   # us


print(f"{'shape':14s} {'M':>6s} {'N':>6s} {'K':>6s} | {'torch.matmul us':>16s} {'TFLOP/s':>8s} | {'vck_gemm us':>12s} {'TFLOP/s':>8s} | const-operand vck us")
ws = torch.zeros(64 << 18, dtype=torch.float32, device=dev)
for name, M, N, K in SHAPES:
    g = torch.Generator(device=dev).manual_seed(1)
    A = torch.randn(M, K, device=dev, generator=g).to(torch.bfloat16)
    W = (torch.randn(N, K, device=dev, generator=g) * 0.05).to(torch.bfloat16)
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    t_lib = timed(lambda: torch.matmul(A, W.t(), out=out))
    s = torch.cuda.current_stream().cuda_stream

    def ours(a=A, w=W):
        lib.vck_gemm_ws(C.c_void_p(a.data_ptr()), C.c_void_p(w.data_ptr()), None, C.c_void_p(out.data_ptr()), M, N, K, K, K, N, 0,
                        C.c_void_p(ws.data_ptr()), C.c_size_t(64 << 20), C.c_void_p(s))
    t_our = timed(ours)
    Ac, Wc = torch.full_like(A, 0.5), torch.full_like(W, 0.25)
    t_const = timed(lambda: ours(Ac, Wc))
    fl = 2.0 * M * N * K
    print(f"{name:14s} {M:6d} {N:6d} {K:6d} | {t_lib:16.1f} {fl / t_lib / 1e6:8.1f} | {t_our:12.1f} {fl / t_our / 1e6:8.1f} | {t_const:8.1f} ({fl / t_const / 1e6:.0f})")
