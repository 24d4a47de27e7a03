#!/usr/bin/env python
"""Per-kernel micro-benchmark on an MI355X through the vck_* C ABI (include/vcoder_kernels.h), timed with events on the
default stream.  usage: python tools/kbench.py [gemm] [gemv] [attn] [dattn]   (env knobs are read by the library)"""
import ctypes as C
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from vcoder_amd import _lib  # noqa: E402

lib = _lib.load()
dev = torch.device("cuda:0")
P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3  # us


def bf16(*shape, scale=1.0):
    return (torch.randn(*shape, device=dev) * scale).to(torch.bfloat16)


def bench_gemm():
    for (M, N, K, epi, name) in [(9728, 12288, 4096, 0, "llm qkv"), (9728, 4096, 4096, 4, "llm o"),
                                 (9728, 22016, 4096, 5, "llm gate-up"), (9728, 22016, 4096, 0, "gate-up/bf16"),
                                 (9728, 22016, 4096, 3, "gate-up/f32"), (9728, 4096, 11008, 4, "llm down"),
                                 (13848, 3072, 1024, 0, "vit qkv"), (13848, 4096, 1024, 1, "vit fc1"),
                                 (13848, 1024, 4096, 4, "vit fc2"), (13824, 4096, 4096, 0, "adapter 2")]:
        A, W = bf16(M, K), bf16(N, K, scale=0.02)
        out = torch.zeros((M, N), dtype=torch.float32 if epi in (3, 4) else torch.bfloat16, device=dev)
        ldo = N // 2 if epi == 5 else N
        us = timeit(lambda: lib.vck_gemm(P(A), P(W), None, P(out), M, N, K, K, K, ldo, epi, None), iters=10)
        print(f"gemm {name:12s} M{M} N{N} K{K} epi{epi}: {us:9.1f} us  {2 * M * N * K / us / 1e6:7.1f} TFLOP/s", flush=True)
        ws = torch.zeros(16 << 20, device=dev)   # 64 MiB fp32 workspace: split-K of a short last round of tiles
        us = timeit(lambda: lib.vck_gemm_ws(P(A), P(W), None, P(out), M, N, K, K, K, ldo, epi, P(ws), C.c_size_t(64 << 20), None),
                    iters=10)
        print(f"gemm+ws {name:9s}                         : {us:9.1f} us  {2 * M * N * K / us / 1e6:7.1f} TFLOP/s", flush=True)


def bench_gemm32():
    """Round 6 go / no-go: the 8-phase 256 x 256 schedule on v_mfma_f32_16x16x32_bf16 (variant 1) against the same schedule on
    v_mfma_f32_32x32x16_bf16 (variant 6), random and constant operands, with the split-K workspace (the engine's form)."""
    ws = torch.zeros(16 << 20, device=dev)
    for (M, N, K, epi, name) in [(9728, 12288, 4096, 0, "llm qkv"), (9728, 4096, 4096, 4, "llm o"),
                                 (9728, 22016, 4096, 5, "llm gate-up"), (9728, 4096, 11008, 4, "llm down"),
                                 (19456, 15360, 5120, 0, "13b qkv"), (19456, 27648, 5120, 5, "13b gate-up"),
                                 (13848, 3072, 1024, 0, "vit qkv"), (13848, 4096, 1024, 1, "vit fc1"),
                                 (13848, 1024, 4096, 4, "vit fc2"), (13824, 4096, 4096, 0, "adapter 2")]:
        for fill in ("random", "const"):
            if fill == "random":
                A, W = bf16(M, K), bf16(N, K, scale=0.02)
            else:
                A, W = torch.full((M, K), 0.5, device=dev, dtype=torch.bfloat16), torch.full((N, K), 0.01, device=dev, dtype=torch.bfloat16)
            out = torch.zeros((M, N), dtype=torch.float32 if epi in (3, 4) else torch.bfloat16, device=dev)
            ldo = N // 2 if epi == 5 else N
            res = []
            for v in (1, 6, 1, 6):
                lib.vck_set_gemm_variant(v)
                us = timeit(lambda: lib.vck_gemm_ws(P(A), P(W), None, P(out), M, N, K, K, K, ldo, epi, P(ws), C.c_size_t(64 << 20), None),
                            iters=10)
                res.append(us)
            lib.vck_set_gemm_variant(-1)
            f = lambda us: 2 * M * N * K / us / 1e6
            print(f"gemm32 {name:12s} {fill:6s} M{M} N{N} K{K} epi{epi}: 16x16x32 {res[0]:8.1f} / {res[2]:8.1f} us ({f(min(res[0], res[2])):6.1f} TF/s)   "
                  f"32x32x16 {res[1]:8.1f} / {res[3]:8.1f} us ({f(min(res[1], res[3])):6.1f} TF/s)   ratio {min(res[1], res[3]) / min(res[0], res[2]):.3f}", flush=True)


def bench_gemm_qkv():
    """Round 6: the QKV projection of a prefill layer as GEMM (EPI_BF16) + qkv_split against the GEMM with RoPE + head split + KV write
    in its epilogue (vck_gemm_qkv), with / without the split-K workspace."""
    for (B, T, H, K, name) in [(8, 1216, 32, 4096, "7b B=8"), (16, 1216, 40, 5120, "13b B=16"), (8, 1217, 32, 4096, "7b B=8 S=1217")]:
        D, M = H * 128, B * T
        Ts, S_cap = (T + 63) // 64 * 64, 1344
        A, W = bf16(M, K), bf16(3 * D, K, scale=0.02)
        qkv = torch.zeros((M, 3 * D), dtype=torch.bfloat16, device=dev)
        q, vt = torch.zeros((B, H, Ts, 128), dtype=torch.bfloat16, device=dev), torch.zeros((B, H, 128, Ts), dtype=torch.bfloat16, device=dev)
        k, v = torch.zeros((B, H, S_cap, 128), dtype=torch.bfloat16, device=dev), torch.zeros((B, H, S_cap, 128), dtype=torch.bfloat16, device=dev)
        cos, sin = torch.rand(S_cap, 64, device=dev), torch.rand(S_cap, 64, device=dev)
        ws = torch.zeros(16 << 20, device=dev)
        for wsb in (64 << 20, 0):
            wp = P(ws) if wsb else None
            g = timeit(lambda: lib.vck_gemm_ws(P(A), P(W), None, P(qkv), M, 3 * D, K, K, K, 3 * D, 0, wp, C.c_size_t(wsb), None), iters=10)
            sp = timeit(lambda: lib.vck_qkv_split_kv(P(qkv), P(q), P(k), P(v), P(vt), B, T, H, 128, Ts, S_cap, Ts, P(cos), P(sin), None), iters=10)
            f = timeit(lambda: lib.vck_gemm_qkv(P(A), None, P(W), None, None, B, T, H, K, K, P(q), P(k), P(v), P(vt), None, None, Ts, S_cap,
                                                 Ts, 0, P(cos), P(sin), 0, wp, C.c_size_t(wsb), None), iters=10)
            print(f"gemm_qkv {name:14s} ws={wsb >> 20:2d}MiB: gemm {g:8.1f} + split {sp:6.1f} = {g + sp:8.1f} us   fused {f:8.1f} us   "
                  f"({2 * M * 3 * D * K / f / 1e6:6.1f} TF/s)", flush=True)


def bench_gemm_f8():
    """W8A8 prefill linears (weight format 2): the activation quantiser and the e4m3 x e4m3 GEMM, next to the bf16 GEMM of the
    same shape (7b B=8 and 13b B=16 prefill shapes)."""
    ws = torch.zeros(16 << 20, device=dev)
    for (M, N, K, epi, name) in [(9728, 12288, 4096, 0, "7b qkv"), (9728, 4096, 4096, 4, "7b o"),
                                 (9728, 22016, 4096, 5, "7b gate-up"), (9728, 4096, 11008, 4, "7b down"),
                                 (19456, 15360, 5120, 0, "13b qkv"), (19456, 5120, 5120, 4, "13b o"),
                                 (19456, 27648, 5120, 5, "13b gate-up"), (19456, 5120, 13824, 4, "13b down")]:
        A, W = bf16(M, K), bf16(N, K, scale=0.02)
        out = torch.zeros((M, N), dtype=torch.float32 if epi in (3, 4) else torch.bfloat16, device=dev)
        ldo = N // 2 if epi == 5 else N
        us16 = timeit(lambda: lib.vck_gemm_ws(P(A), P(W), None, P(out), M, N, K, K, K, ldo, epi, P(ws), C.c_size_t(64 << 20), None),
                      iters=10)
        Q = torch.zeros((M, K), dtype=torch.uint8, device=dev)
        sa = torch.zeros(M, device=dev)
        Wq = torch.zeros(N * K, dtype=torch.uint8, device=dev)
        Wrow = torch.zeros((N, K), dtype=torch.uint8, device=dev)
        sw = torch.zeros(N, device=dev)
        lib.vck_quantize_fp8_rows(P(W), P(Wq), P(sw), P(Wrow), N, K, None)
        usq = timeit(lambda: lib.vck_quant_act_rows(P(A), K, P(Q), P(sa), M, K, None), iters=10)
        us8 = timeit(lambda: lib.vck_gemm_f8(P(Q), P(sa), P(Wrow), P(sw), P(out), M, N, K, ldo, epi, P(ws), C.c_size_t(64 << 20),
                                             None), iters=10)
        fl = 2 * M * N * K / 1e6
        print(f"gemm_f8 {name:12s} M{M} N{N} K{K}: bf16 {us16:8.1f} us {fl / us16:7.1f} TF | quant {usq:6.1f} us "
              f"({M * K * 3 / usq / 1e3:6.0f} GB/s) + e4m3 {us8:8.1f} us {fl / us8:7.1f} TF  => x{us16 / (usq + us8):.2f}", flush=True)


def bench_gemv():
    M = 8
    for (N, K, epi, name) in [(12288, 4096, 0, "qkv"), (4096, 4096, 2, "o"), (22016, 4096, 3, "gate-up"),
                              (4096, 11008, 2, "down"), (32000, 4096, 1, "lm_head")]:
        X = bf16(M, K)
        # rotate over 8 weight copies so the 256 MB Infinity Cache cannot hold them
        Ws = [bf16(N * K, scale=0.02) for _ in range(8)]
        out = torch.zeros((M, N), dtype=torch.float32 if epi in (1, 2) else torch.bfloat16, device=dev)
        ldo = N // 2 if epi == 3 else N
        it = [0]

        def f():
            it[0] += 1
            lib.vck_gemv(P(X), P(Ws[it[0] % 8]), P(out), M, N, K, ldo, epi, None)
        us = timeit(f, iters=40)
        print(f"gemv {name:8s} N{N} K{K} epi{epi}: {us:7.1f} us  {2 * N * K / us / 1e3:7.1f} GB/s", flush=True)
        if epi == 2:
            scratch = torch.zeros(4 * (N // 16) * 2 * 256, device=dev)
            counters = torch.zeros(N // 16 * 2, dtype=torch.int32, device=dev)
            for ks in (2, 3, 4):
                def h():
                    it[0] += 1
                    lib.vck_gemv_ex(P(X), P(Ws[it[0] % 8]), None, P(out), None, None, None, None, 16, C.c_float(1e-5), P(scratch),
                                    P(counters), ks, M, N, K, ldo, epi, None)
                us = timeit(h, iters=40)
                print(f"gemv {name:8s} split-K {ks}: {us:7.1f} us  {2 * N * K / us / 1e3:7.1f} GB/s", flush=True)
        if epi in (0, 1, 3):
            xf = torch.randn(16, K, device=dev)
            w = torch.rand(K, device=dev) + 0.5
            npart = (K // 16 + 15) // 16 * 16
            ssq = torch.rand(16, npart, device=dev)

            def g():
                it[0] += 1
                lib.vck_gemv_ex(P(X), P(Ws[it[0] % 8]), None, P(out), P(ssq), None, None, None, npart, C.c_float(1e-5), None, None,
                                0, M, N, K, ldo, epi, None)
            us = timeit(g, iters=40)
            print(f"gemv+rstd {name:8s}          : {us:7.1f} us  {2 * N * K / us / 1e3:7.1f} GB/s", flush=True)


def bench_gemv_rows():
    """The decode pool's question: what does a weight pass cost at M = 8 / 16 (one row group) and M = 24 / 32 (two)?
    7b shapes with the folded-RMSNorm operands the engine uses (consumers scale by rstd; producers publish partials + xg)."""
    for M in (8, 16, 24, 32):
        tot = 0.0
        for (N, K, epi, name, cnt) in [(12288, 4096, 0, "qkv", 32), (4096, 4096, 2, "o", 32), (22016, 4096, 3, "gate-up", 32),
                                       (4096, 11008, 2, "down", 32), (32000, 4096, 1, "lm_head", 1)]:
            X = bf16(32, K)
            Ws = [bf16(N * K, scale=0.02) for _ in range(8)]
            out = torch.zeros((32, N), dtype=torch.float32 if epi in (1, 2) else torch.bfloat16, device=dev)
            ldo = N // 2 if epi == 3 else N
            npart = (4096 // 16 + 15) // 16 * 16
            ssq = torch.rand(32, npart, device=dev)
            gw = torch.rand(N, device=dev) + 0.5
            xg = torch.zeros((32, N), dtype=torch.bfloat16, device=dev)
            scratch = torch.zeros(4 * (N // 16) * 2 * 256, device=dev)
            counters = torch.zeros(N // 16 * 2, dtype=torch.int32, device=dev)
            it = [0]

            def f():
                it[0] += 1
                if epi == 2:
                    lib.vck_gemv_ex(P(X), P(Ws[it[0] % 8]), None, P(out), None, P(ssq), P(gw), P(xg), npart, C.c_float(1e-5),
                                    P(scratch), P(counters), 0, M, N, K, ldo, epi, None)
                else:
                    lib.vck_gemv_ex(P(X), P(Ws[it[0] % 8]), None, P(out), P(ssq), None, None, None, npart, C.c_float(1e-5),
                                    None, None, 0, M, N, K, ldo, epi, None)
            us = timeit(f, iters=40)
            tot += us * cnt
            print(f"gemv_rows M{M:2d} {name:8s} N{N} K{K}: {us:7.1f} us  {2 * N * K / us / 1e3:7.1f} GB/s", flush=True)
        print(f"gemv_rows M{M:2d} all GEMVs of a 7b decode step: {tot / 1e3:6.3f} ms = {tot / 1e3 / M:6.4f} ms per row", flush=True)


def bench_gemv_wg():
    """the GEMV of precision mode "split" (gemv_wg_kernel: hi / lo activation rows in one weight pass) at the 7b shapes, 8 and 32 rows:
    four waves per workgroup everywhere (vck_set_gemv_variant(2), rounds 4-5) against the launcher's choice (six where that balances
    the launch: qkv, gate / up)"""
    lib.vck_gemv_full.restype = None
    for M, G in ((8, 8), (32, 32)):
        tot = {2: 0.0, -1: 0.0}
        for (N, K, epi, name, cnt) in [(12288, 4096, 1, "qkv", 32), (4096, 4096, 2, "o", 32), (22016, 4096, 3, "gate-up", 32),
                                       (4096, 11008, 2, "down", 32), (32000, 4096, 1, "lm_head", 1)]:
            X = bf16(2 * G, K)
            Ws = [bf16(N * K, scale=0.02) for _ in range(8)]
            rows_out = 2 * G if epi == 3 else 32
            out = torch.zeros((max(rows_out, 32), N), dtype=torch.float32 if epi in (1, 2) else torch.bfloat16, device=dev)
            ldo = N // 2 if epi == 3 else N
            npart = (max(K, N) // 16 + 15) // 16 * 16
            ssq = torch.rand(32, npart, device=dev)
            ssq_out = torch.zeros(32, npart, device=dev)
            gw = torch.rand(N, device=dev) + 0.5
            xg = torch.zeros((2 * G, N), dtype=torch.bfloat16, device=dev)
            nsk = 4 * (N // 16) * 2 * 256
            scratch = torch.zeros(nsk, device=dev)
            counters = torch.zeros(N // 16 * 2, dtype=torch.int32, device=dev)
            row = []
            for v in (2, -1):
                lib.vck_set_gemv_variant(v)
                it = [0]

                def f():
                    it[0] += 1
                    prod = epi == 2
                    lib.vck_gemv_full(P(X), P(Ws[it[0] % 8]), None, P(out), None if prod else P(ssq), P(ssq_out) if prod else None,
                                      P(gw) if prod else None, P(xg) if prod else None, C.c_int(npart), C.c_float(1e-5), P(scratch),
                                      C.c_ulonglong(nsk), P(counters), C.c_int(N // 16 * 2), C.c_int(0), M, N, K, ldo, epi, G, None)
                us = timeit(f, iters=60)
                tot[v] += us * cnt
                row.append(f"{'4 waves' if v == 2 else 'chosen '} {us:6.2f} us {2 * N * K / us / 1e3:6.0f} GB/s")
            print(f"gemv_wg M{M:2d} {name:8s} N{N} K{K}: " + " | ".join(row), flush=True)
        print(f"gemv_wg M{M:2d} all GEMVs of a 7b split step: 4 waves {tot[2] / 1e3:6.3f} ms | chosen {tot[-1] / 1e3:6.3f} ms", flush=True)
    lib.vck_set_gemv_variant(-1)


def bench_gemv_wide():
    """ring-kernel GEMV, pair geometry (vck_set_gemv_wide 0) vs the "wide" one (2: ceil(tiles / 256) tiles per workgroup, one deep
    ring per CU, every class) on every > 512-tile matrix of the 7b and 13b models at 8 / 16 / 24 / 32 rows; same bits?"""
    for (N, K, epi, name) in [(12288, 4096, 0, "7b qkv"), (22016, 4096, 3, "7b gate-up"), (15360, 5120, 0, "13b qkv"),
                              (27648, 5120, 3, "13b gate-up")]:
        X = bf16(32, K)
        Ws = [bf16(N * K, scale=0.02) for _ in range(6)]
        ldo = N // 2 if epi == 3 else N
        npart = (K // 16 + 15) // 16 * 16
        ssq = torch.rand(32, npart, device=dev)
        it = [0]
        for M in (8, 16, 24, 32):
            outs, t = {}, {}
            for wide in (0, 2):
                lib.vck_set_gemv_wide(wide)
                out = torch.zeros((32, ldo), dtype=torch.float32 if epi == 1 else torch.bfloat16, device=dev)

                def f():
                    it[0] += 1
                    lib.vck_gemv_ex(P(X), P(Ws[it[0] % 6]), None, P(out), P(ssq), None, None, None, npart, C.c_float(1e-5),
                                    None, None, 0, M, N, K, ldo, epi, None)
                t[wide] = timeit(f, iters=40)
                it[0] = 5
                f()
                torch.cuda.synchronize()
                outs[wide] = out.clone()
            same = torch.equal(outs[0].view(torch.int32 if epi == 1 else torch.int16), outs[2].view(torch.int32 if epi == 1 else torch.int16))
            print(f"gemv_wide M{M:2d} {name:12s}: default {t[0]:6.1f} us ({2 * N * K / t[0] / 1e3:6.0f} GB/s)  wide {t[2]:6.1f} us "
                  f"({2 * N * K / t[2] / 1e3:6.0f} GB/s)  same bits {same}", flush=True)
    lib.vck_set_gemv_wide(-1)


def bench_gemm_chunk():
    """go / no-go of the chunked-prefill idea: the decoder GEMMs at M = 288 / 320 / 384 rows (a prefill chunk plus
    the pool's 32 decode rows) with the weights streamed from HBM (4 rotating copies); the sum per layer against 150 us."""
    for M in (288, 320, 384, 512):
        tot = 0.0
        for (N, K, epi, name) in [(12288, 4096, 0, "qkv"), (4096, 4096, 4, "o"), (22016, 4096, 5, "gate-up"), (4096, 11008, 4, "down")]:
            A = bf16(M, K)
            Ws = [bf16(N, K, scale=0.02) for _ in range(4)]
            out = torch.zeros((M, N), dtype=torch.float32 if epi in (3, 4) else torch.bfloat16, device=dev)
            ldo = N // 2 if epi == 5 else N
            ws = torch.zeros(16 << 20, device=dev)
            it = [0]

            def f():
                it[0] += 1
                lib.vck_gemm_ws(P(A), P(Ws[it[0] % 4]), None, P(out), M, N, K, K, K, ldo, epi, P(ws), C.c_size_t(64 << 20), None)
            us = timeit(f, iters=20)
            tot += us
            print(f"gemm_chunk M{M} {name:8s}: {us:7.1f} us  {2 * M * N * K / us / 1e6:7.1f} TFLOP/s  weights {2 * N * K / us / 1e3:6.0f} GB/s", flush=True)
        print(f"gemm_chunk M{M} layer sum: {tot:7.1f} us (go if <= 150)", flush=True)


def bench_gemv_rows8():
    """W8A16 weights at 17..32 rows (the pool's rows under the fp8 weight formats): qkv and gate / up of the 7b and 13b models — the pair
    geometry (vck_set_gemv_wide(0)) against the one-workgroup-per-CU geometry (default since round 6 for > 512 tiles)"""
    for M in (12, 16, 24, 32):
        for (N, K, epi, name) in [(12288, 4096, 0, "7b qkv"), (22016, 4096, 3, "7b gate-up"), (15360, 5120, 0, "13b qkv"),
                                  (27648, 5120, 3, "13b gate-up")]:
            X = bf16(32, K)
            Ws = []
            sc = torch.zeros(N, device=dev)
            for _ in range(6):
                W = bf16(N, K, scale=0.02)
                Wq = torch.zeros(N * K, dtype=torch.uint8, device=dev)
                lib.vck_quantize_fp8(P(W), P(Wq), P(sc), N, K, None)
                Ws.append(Wq)
            torch.cuda.synchronize()
            out = torch.zeros((32, N), dtype=torch.bfloat16, device=dev)
            ldo = N // 2 if epi == 3 else N
            npart = (K // 16 + 15) // 16 * 16
            ssq = torch.rand(32, npart, device=dev)
            row = []
            for wide in (0, -1, 2):
                lib.vck_set_gemv_wide(wide)
                it = [0]

                def f():
                    it[0] += 1
                    lib.vck_gemv_ex(P(X), P(Ws[it[0] % 6]), P(sc), P(out), P(ssq), None, None, None, npart, C.c_float(1e-5),
                                    None, None, 0, M, N, K, ldo, epi, None)
                us = timeit(f, iters=60)
                row.append(f"{'pairs' if wide == 0 else 'default' if wide < 0 else 'wide everywhere'} {us:6.1f} us {N * K / us / 1e3:6.0f} GB/s")
            print(f"gemv_rows8 M{M:2d} {name:12s} N{N} K{K}: " + " | ".join(row), flush=True)
    lib.vck_set_gemv_wide(-1)


def bench_gemv_fp8_ks():
    """13b o_proj / down with W8A16 weights (320 tiles: the 257..512-tile class, tile pairs x KS_MID = 3 K-slices by default — a choice
    measured with bf16 weights): explicit K-slice counts at 8 / 16 / 32 rows"""
    for (N, K, name) in [(5120, 5120, "13b o"), (5120, 13824, "13b down")]:
        X = bf16(32, K)
        Ws = []
        sc = torch.zeros(N, device=dev)
        for _ in range(6):
            W = bf16(N, K, scale=0.02)
            Wq = torch.zeros(N * K, dtype=torch.uint8, device=dev)
            lib.vck_quantize_fp8(P(W), P(Wq), P(sc), N, K, None)
            Ws.append(Wq)
        torch.cuda.synchronize()
        out = torch.zeros((32, N), dtype=torch.float32, device=dev)
        npart = (max(K, N) // 16 + 15) // 16 * 16
        ssq_out = torch.zeros(32, npart, device=dev)
        gw = torch.rand(N, device=dev) + 0.5
        xg = torch.zeros((32, N), dtype=torch.bfloat16, device=dev)
        scratch = torch.zeros(8 * (N // 16) * 2 * 256, device=dev)
        counters = torch.zeros(N // 16 * 2, dtype=torch.int32, device=dev)
        for M in (8, 16, 32):
            row = []
            for ks in (0, 1, 2, 3, 4, 6):
                it = [0]

                def f():
                    it[0] += 1
                    lib.vck_gemv_ex(P(X), P(Ws[it[0] % 6]), P(sc), P(out), None, P(ssq_out), P(gw), P(xg), npart, C.c_float(1e-5),
                                    P(scratch), P(counters), ks, M, N, K, N, 2, None)
                us = timeit(f, iters=60)
                row.append(f"ks {'default' if ks == 0 else ks}: {us:5.1f} us")
            print(f"gemv_fp8_ks {name:8s} N{N} K{K} M{M:2d}: " + " | ".join(row), flush=True)


def bench_dattn_rows():
    """decode attention over B rows with per-row positions (the pool's form), ctx ~1280"""
    H, hd, S = 32, 128, 2048
    D = H * hd
    cos, sin = torch.rand(S, hd // 2, device=dev), torch.rand(S, hd // 2, device=dev)
    for B in (8, 16, 24, 32):
        qkv = bf16(B, 3 * D)
        ks = [bf16(B, H, S, hd) for _ in range(2)]
        vs = [bf16(B, H, S, hd) for _ in range(2)]
        out = torch.zeros((B, D), dtype=torch.bfloat16, device=dev)
        pos = torch.tensor([1216 + (7 * b) % 128 for b in range(B)], dtype=torch.int32, device=dev)
        act = torch.ones(B, dtype=torch.int32, device=dev)
        it = [0]

        def f():
            it[0] += 1
            lib.vck_attention_decode_rows(P(qkv), P(ks[it[0] % 2]), P(vs[it[0] % 2]), P(out), B, H, hd, S, P(pos), 1, P(act),
                                          P(cos), P(sin), C.c_float(1 / math.sqrt(hd)), None)
        us = timeit(f, iters=40)
        byts = 4.0 * float((pos + 1).sum().item()) * D
        print(f"decode attention rows B={B:2d} ctx~1280: {us:7.1f} us  {byts / us / 1e3:7.1f} GB/s", flush=True)


def bench_dattn_split():
    """decode attention of precision mode split: fp32 caches vs fp24 caches (3 bytes per element), rows as the pool spans them"""
    H, hd, S = 32, 128, 2048
    D = H * hd
    cos, sin = torch.rand(S, hd // 2, device=dev), torch.rand(S, hd // 2, device=dev)
    for B in (8, 16, 32):
        G = 8 if B <= 8 else (16 if B <= 16 else 32)
        qkv = torch.randn(B, 3 * D, device=dev)
        out = torch.zeros((2 * G, D), dtype=torch.bfloat16, device=dev)
        pos = torch.tensor([1216 + (7 * b) % 128 for b in range(B)], dtype=torch.int32, device=dev)
        act = torch.ones(B, dtype=torch.int32, device=dev)
        keys = float((pos + 1).sum().item())
        for name, es, fn in (("fp32", 4, lib.vck_attention_decode_kv32), ("fp24", 3, lib.vck_attention_decode_kv24)):
            ks = [torch.randint(0, 255, (B, H, S, hd * es), dtype=torch.uint8, device=dev) for _ in range(2)]
            vs = [torch.randint(0, 255, (B, H, S, hd * es), dtype=torch.uint8, device=dev) for _ in range(2)]
            for t in ks + vs:   # keep exponents sane: clear the top exponent bits of every element's high byte
                t.view(B, H, S, -1)[...] &= 0x3F
            it = [0]

            def f():
                it[0] += 1
                fn(P(qkv), P(ks[it[0] % 2]), P(vs[it[0] % 2]), P(out), B, H, hd, S, P(pos), 1, P(act), P(cos), P(sin),
                   C.c_float(1 / math.sqrt(hd)), G, None)
            us = timeit(f, iters=30)
            byts = 2.0 * keys * D * es
            print(f"decode attention split {name} B={B:2d} ctx~1280: {us:7.1f} us  {byts / us / 1e3:7.1f} GB/s", flush=True)


def bench_dattn_kv8():
    """decode attention of the bf16 step: bf16 caches vs the e4m3 caches of the fp8 weight format, 13b heads (40 x 128)"""
    H, hd, S = 40, 128, 2048
    D = H * hd
    cos, sin = torch.rand(S, hd // 2, device=dev), torch.rand(S, hd // 2, device=dev)
    for B in (8, 16, 32):
        qkv = bf16(B, 3 * D)
        out = torch.zeros((B, D), dtype=torch.bfloat16, device=dev)
        pos = torch.tensor([1216 + (7 * b) % 128 for b in range(B)], dtype=torch.int32, device=dev)
        act = torch.ones(B, dtype=torch.int32, device=dev)
        keys = float((pos + 1).sum().item())
        for name, es, fn in (("bf16", 2, lib.vck_attention_decode_rows), ("e4m3", 1, lib.vck_attention_decode_kv8)):
            ks = [torch.randint(0, 120, (B, H, S, hd * es), dtype=torch.uint8, device=dev) for _ in range(2)]
            vs = [torch.randint(0, 120, (B, H, S, hd * es), dtype=torch.uint8, device=dev) for _ in range(2)]
            it = [0]

            def f():
                it[0] += 1
                fn(P(qkv), P(ks[it[0] % 2]), P(vs[it[0] % 2]), P(out), B, H, hd, S, P(pos), 1, P(act), P(cos), P(sin),
                   C.c_float(1 / math.sqrt(hd)), None)
            us = timeit(f, iters=30)
            byts = 2.0 * keys * D * es
            print(f"decode attention 13b heads {name} B={B:2d} ctx~1280: {us:7.1f} us  {byts / us / 1e3:7.1f} GB/s", flush=True)


def bench_gemv_pair():
    """Two streams run the SAME GEMV (same weights, different activations / outputs) at the same time: does the second
    reader hit the Infinity Cache / merge with the first?  pair time ~ single time => yes."""
    M = 8
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    for (N, K, epi, name) in [(12288, 4096, 0, "qkv"), (22016, 4096, 3, "gate-up"), (4096, 11008, 1, "down(f32)")]:
        X1, X2 = bf16(M, K), bf16(M, K)
        Ws = [bf16(N * K, scale=0.02) for _ in range(8)]
        dt = torch.float32 if epi in (1, 2) else torch.bfloat16
        o1 = torch.zeros((M, N), dtype=dt, device=dev)
        o2 = torch.zeros((M, N), dtype=dt, device=dev)
        ldo = N // 2 if epi == 3 else N
        it = [0]

        def single():
            it[0] += 1
            lib.vck_gemv(P(X1), P(Ws[it[0] % 8]), P(o1), M, N, K, ldo, epi, C.c_void_p(s1.cuda_stream))

        def pair_same():
            it[0] += 1
            w = Ws[it[0] % 8]
            lib.vck_gemv(P(X1), P(w), P(o1), M, N, K, ldo, epi, C.c_void_p(s1.cuda_stream))
            lib.vck_gemv(P(X2), P(w), P(o2), M, N, K, ldo, epi, C.c_void_p(s2.cuda_stream))

        def pair_diff():
            it[0] += 1
            lib.vck_gemv(P(X1), P(Ws[it[0] % 8]), P(o1), M, N, K, ldo, epi, C.c_void_p(s1.cuda_stream))
            lib.vck_gemv(P(X2), P(Ws[(it[0] + 4) % 8]), P(o2), M, N, K, ldo, epi, C.c_void_p(s2.cuda_stream))

        def wall(fn, iters=40):
            for _ in range(4):
                fn()
            torch.cuda.synchronize()
            import time
            t0 = time.perf_counter()
            for _ in range(iters):
                fn()
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / iters * 1e6
        print(f"pair {name:10s}: single {wall(single):6.1f} us | two streams, same weights {wall(pair_same):6.1f} us | "
              f"two streams, different weights {wall(pair_diff):6.1f} us", flush=True)


def bench_gemv13():
    """VCoder-DS 13b decode shapes at M = 16 and M = 8 (config C3 runs M = 16)."""
    for M in (16, 8):
        for (N, K, epi, name) in [(15360, 5120, 0, "qkv"), (5120, 5120, 2, "o"), (27648, 5120, 3, "gate-up"),
                                  (5120, 13824, 2, "down"), (32000, 5120, 1, "lm_head")]:
            X = bf16(M, K)
            Ws = [bf16(N * K, scale=0.02) for _ in range(4)]
            out = torch.zeros((M, N), dtype=torch.float32 if epi in (1, 2) else torch.bfloat16, device=dev)
            ldo = N // 2 if epi == 3 else N
            it = [0]

            def f():
                it[0] += 1
                lib.vck_gemv(P(X), P(Ws[it[0] % 4]), P(out), M, N, K, ldo, epi, None)
            us = timeit(f, iters=40)
            print(f"gemv13 M{M} {name:8s} N{N} K{K} epi{epi}: {us:7.1f} us  {2 * N * K / us / 1e3:7.1f} GB/s", flush=True)
            if epi == 2:
                scratch = torch.zeros(4 * (N // 16) * 2 * 256, device=dev)
                counters = torch.zeros(N // 16 * 2, dtype=torch.int32, device=dev)
                for ks in (2, 3, 4):
                    def h():
                        it[0] += 1
                        lib.vck_gemv_ex(P(X), P(Ws[it[0] % 4]), None, P(out), None, None, None, None, 16, C.c_float(1e-5),
                                        P(scratch), P(counters), ks, M, N, K, ldo, epi, None)
                    us = timeit(h, iters=40)
                    print(f"gemv13 M{M} {name:8s} split-K {ks}: {us:7.1f} us  {2 * N * K / us / 1e3:7.1f} GB/s", flush=True)


def bench_gemv_fp8():
    """W8A16 form: same shapes, e4m3 bytes (random bytes: the kernel's speed does not depend on the values)."""
    for M in (8, 16):
        for (N, K, epi, name) in [(12288, 4096, 0, "qkv"), (4096, 4096, 2, "o"), (22016, 4096, 3, "gate-up"),
                                  (4096, 11008, 2, "down")]:
            X = bf16(M, K)
            Ws = [torch.randint(0, 120, (N * K,), dtype=torch.uint8, device=dev) for _ in range(8)]
            sc = torch.ones(N, device=dev)
            out = torch.zeros((M, N), dtype=torch.float32 if epi in (1, 2) else torch.bfloat16, device=dev)
            ldo = N // 2 if epi == 3 else N
            it = [0]
            xf = torch.randn(16, K, device=dev)
            w = torch.rand(K, device=dev) + 0.5
            npart = (K // 16 + 15) // 16 * 16
            ssq = torch.rand(16, npart, device=dev)
            norm = epi in (0, 1, 3)

            def f():
                it[0] += 1
                lib.vck_gemv_ex(P(X), P(Ws[it[0] % 8]), P(sc), P(out), P(ssq) if norm else None, None, None, None, npart,
                                C.c_float(1e-5), None, None, 0, M, N, K, ldo, epi, None)
            us = timeit(f, iters=40)
            print(f"gemv_fp8 M{M} {name:8s} N{N} K{K} epi{epi} norm{int(norm)}: {us:7.1f} us  {N * K / us / 1e3:7.1f} GB/s",
                  flush=True)


def bench_attn():
    for (B, H, T, hd, causal, name) in [(8, 32, 1216, 128, 1, "llm prefill"), (24, 16, 577, 64, 0, "vit")]:
        Ts = (T + 63) // 64 * 64
        q, k, vt = bf16(B, H, Ts, hd), bf16(B, H, Ts, hd), bf16(B, H, hd, Ts)
        out = torch.zeros((B * T, H * hd), dtype=torch.bfloat16, device=dev)
        us = timeit(lambda: lib.vck_attention(P(q), P(k), P(vt), P(out), B, H, T, hd, Ts, Ts, causal,
                                              C.c_float(1 / math.sqrt(hd)), None), iters=10)
        fl = 4.0 * B * H * T * T * hd * (0.5 if causal else 1.0)
        print(f"attention {name:12s} variant={os.environ.get('VC_ATTN_VARIANT', '0')}: {us:8.1f} us  "
              f"{fl / us / 1e6:7.1f} TFLOP/s", flush=True)


def bench_dattn():
    B, H, hd, pos = 8, 32, 128, 1280
    S = 1344
    D = H * hd
    qkv = bf16(B, 3 * D)
    ks = [bf16(B, H, S, hd) for _ in range(4)]
    vs = [bf16(B, H, S, hd) for _ in range(4)]
    out = torch.zeros((B, D), dtype=torch.bfloat16, device=dev)
    posd = torch.tensor([pos], dtype=torch.int32, device=dev)
    cos, sin = torch.rand(S, hd // 2, device=dev), torch.rand(S, hd // 2, device=dev)
    it = [0]

    def f():
        it[0] += 1
        lib.vck_attention_decode_fused(P(qkv), P(ks[it[0] % 4]), P(vs[it[0] % 4]), P(out), B, H, hd, S, P(posd), P(cos),
                                       P(sin), C.c_float(1 / math.sqrt(hd)), None)
    us = timeit(f, iters=40)
    print(f"decode attention ctx={pos + 1}: {us:7.1f} us  {4.0 * B * (pos + 1) * D / us / 1e3:7.1f} GB/s", flush=True)


if __name__ == "__main__":
    what = sys.argv[1:] or ["gemm", "gemv", "attn", "dattn"]
    table = {"gemm_qkv": bench_gemm_qkv, "gemm32": bench_gemm32, "gemm": bench_gemm, "gemv": bench_gemv, "attn": bench_attn, "dattn": bench_dattn, "gemv_fp8": bench_gemv_fp8,
             "gemv13": bench_gemv13, "gemv_pair": bench_gemv_pair, "gemv_rows": bench_gemv_rows, "dattn_rows": bench_dattn_rows,
             "gemm_f8": bench_gemm_f8, "gemv_rows8": bench_gemv_rows8, "gemv_wide": bench_gemv_wide, "gemm_chunk": bench_gemm_chunk,
             "dattn_split": bench_dattn_split, "dattn_kv8": bench_dattn_kv8, "gemv_wg": bench_gemv_wg, "gemv_fp8_ks": bench_gemv_fp8_ks}
    for w in what:
        table[w]()
