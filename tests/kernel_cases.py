"""Per-kernel parity cases shared by the CPU-emulator tests (small shapes) and the `-m gpu` tests
(true shapes).  Every case computes the expected result with the oracle (oracle/cpu_ref.py, fp32 torch
ops with bf16 rounding emulation where the kernel rounds) and compares the kernel's output.

Backends expose the same C-ABI symbols (include/vcoder_kernels.h):
  EmuBackend : tests/emu/libvcoder_emu.so   (host pointers, thread-per-lane functional emulator; CPU only)
  HipBackend : vcoder_amd/lib/libvcoder_hip.so (device pointers; real gfx950 kernels)
"""
from __future__ import annotations

import ctypes
import math
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import cpu_ref  # noqa: E402  (oracle: test infrastructure)
from vcoder_amd import synth  # noqa: E402

c_p = ctypes.c_void_p


def bf16_round(x: np.ndarray) -> np.ndarray:
    return synth.round_to_bf16(np.asarray(x, dtype=np.float32))


def _op_bits(a, operands):
    """fp32 array -> the bit patterns of the library's 16-bit operand format (uint16): bfloat16, or IEEE fp16 for the -DVC_F16 build"""
    a = np.asarray(a, dtype=np.float32)
    if operands == "fp16":
        return np.ascontiguousarray(np.clip(a, -65504.0, 65504.0).astype(np.float16)).view(np.uint16).reshape(np.shape(a))
    return synth.to_bf16_bits(a).reshape(np.shape(a))


def _op_decode(bits, operands):
    bits = np.ascontiguousarray(bits).view(np.uint16)
    if operands == "fp16":
        return bits.view(np.float16).astype(np.float32).reshape(bits.shape)
    return synth.from_bf16_bits(bits).reshape(bits.shape)


class EmuBackend:
    name = "emu"

    def __init__(self, operands: str = "bf16"):
        # VC_EMU_LIB: another build of the emulator library (e.g. with the host engine under AddressSanitizer, tools/emu_asan.sh)
        self.operands = operands
        f16 = operands == "fp16"
        path = os.path.join(ROOT, "tests", "emu", "libvcoder_emu_f16.so") if f16 else \
            (os.environ.get("VC_EMU_LIB") or os.path.join(ROOT, "tests", "emu", "libvcoder_emu.so"))
        if not os.path.exists(path):
            import subprocess

            subprocess.check_call([os.path.join(ROOT, "tests", "emu", "build_emu.sh")] + (["f16"] if f16 else []))
        self.lib = ctypes.CDLL(path)

    # every array handed to a kernel ends (to 16 bytes) at an inaccessible page — an out-of-bounds access faults (VC_EMU_GUARD=0:
    # plain numpy arrays); the emulator's hipMalloc does the same for the engine's device buffers (tests/emu/hip_emu.h)
    @staticmethod
    def _guard(a):
        if os.environ.get("VC_EMU_GUARD", "1") == "0":
            return a
        import mmap

        a = np.ascontiguousarray(a)
        page, need = 4096, (max(a.nbytes, 1) + 15) // 16 * 16
        total = (need + page - 1) // page * page + page
        buf = mmap.mmap(-1, total)
        addr = ctypes.addressof(ctypes.c_char.from_buffer(buf))
        front = os.environ.get("VC_EMU_GUARD") == "2"     # the mirror image: the array starts right behind the dead page
        if ctypes.CDLL(None, use_errno=True).mprotect(ctypes.c_void_p(addr + (0 if front else total - page)), ctypes.c_size_t(page), 0) != 0:
            raise OSError("mprotect failed")
        g = np.frombuffer(buf, dtype=a.dtype, count=a.size, offset=page if front else total - page - need).reshape(a.shape)
        g[...] = a
        return g

    # arrays are numpy; bf16 arrays are uint16 bit patterns
    def f32(self, a):
        return self._guard(np.ascontiguousarray(a, dtype=np.float32))

    def bf16(self, a):   # "bf16" = the library's 16-bit operand format
        return self._guard(_op_bits(a, self.operands))

    def i32(self, a):
        return self._guard(np.ascontiguousarray(a, dtype=np.int32))

    def zeros(self, shape, kind):
        return self._guard(np.zeros(shape, dtype={"f32": np.float32, "bf16": np.uint16, "i32": np.int32, "u8": np.uint8}[kind]))

    def ptr(self, a):
        return None if a is None else a.ctypes.data_as(c_p)

    def host_f32(self, a):
        return _op_decode(a, self.operands) if a.dtype == np.uint16 else np.array(a)

    def host_i32(self, a):
        return np.array(a)

    def sync(self):
        pass


class HipBackend:
    name = "hip"

    def __init__(self, operands: str = "bf16"):
        from vcoder_amd import _lib

        self.operands = operands
        self.lib = _lib.load(operands)
        self.dev = torch.device("cuda:0")

    def f32(self, a):
        return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(self.dev)

    def bf16(self, a):   # "bf16" = the library's 16-bit operand format
        return torch.from_numpy(_op_bits(a, self.operands).view(np.int16)).to(self.dev)

    def i32(self, a):
        return torch.from_numpy(np.ascontiguousarray(a, dtype=np.int32)).to(self.dev)

    def zeros(self, shape, kind):
        dt = {"f32": torch.float32, "bf16": torch.int16, "i32": torch.int32, "u8": torch.uint8}[kind]
        return torch.zeros(shape, dtype=dt, device=self.dev)

    def ptr(self, a):
        return None if a is None else c_p(a.data_ptr())

    def host_f32(self, a):
        if a.dtype == torch.int16:
            return _op_decode(a.cpu().numpy(), self.operands).reshape(tuple(a.shape))
        return a.cpu().numpy()

    def host_i32(self, a):
        return a.cpu().numpy()

    def sync(self):
        torch.cuda.synchronize()


def _call(be, fn, *args):
    conv = []
    for a in args:
        if a is None:
            conv.append(None)
        elif isinstance(a, (np.ndarray, torch.Tensor)):
            conv.append(be.ptr(a))
        elif isinstance(a, float):
            conv.append(ctypes.c_float(a))
        elif isinstance(a, int):
            conv.append(ctypes.c_int(a))
        else:
            conv.append(a)
    getattr(be.lib, fn)(*conv, None)  # last arg: stream = default
    be.sync()


def rel_err(got, ref):
    return float(np.abs(got - ref).max() / (np.abs(ref).max() + 1e-12))


# ------------------------------------------------------------------------------------------------
def check_gemm(be, M, N, K, epi, bias=True, seed=0, ws_mb=0):
    """ws_mb > 0: through vck_gemm_ws with an fp32 workspace of that many MiB (enables the split-K remainder round)."""
    rng = np.random.RandomState(seed)
    A = bf16_round(rng.randn(M, K))
    W = bf16_round(rng.randn(N, K) * 0.05)
    b = rng.randn(N).astype(np.float32) * 0.1 if bias else None
    ref = A.astype(np.float64) @ W.T.astype(np.float64) + (b if bias else 0.0)
    t = torch.from_numpy(ref).float()
    if epi in (0, 1, 2):
        out = be.zeros((M, N), "bf16")
        if epi == 1:
            t = cpu_ref.quick_gelu(t)
        if epi == 2:
            t = torch.nn.functional.gelu(t)
    elif epi == 3:
        out = be.zeros((M, N), "f32")
    elif epi == 4:
        r0 = rng.randn(M, N).astype(np.float32)
        out = be.f32(r0.copy())
        t = t + torch.from_numpy(r0)
    else:
        out = be.zeros((M, N // 2), "bf16")
        t = torch.nn.functional.silu(t[:, 0::2]) * t[:, 1::2]
    ldo = N // 2 if epi == 5 else N
    if ws_mb:
        Ad, Wd, bd = be.bf16(A), be.bf16(W), (be.f32(b) if bias else None)
        ws = be.zeros((ws_mb << 18,), "f32")
        be.lib.vck_gemm_ws(be.ptr(Ad), be.ptr(Wd), be.ptr(bd), be.ptr(out), M, N, K, K, K, ldo, epi, be.ptr(ws),
                           ctypes.c_size_t(ws_mb << 20), None)
        be.sync()
    else:
        _call(be, "vck_gemm", be.bf16(A), be.bf16(W), be.f32(b) if bias else None, out, M, N, K, K, K, ldo, epi)
    got = be.host_f32(out)
    ref = t.numpy()
    tol = 2 ** -8 if epi in (0, 1, 2, 5) else 1e-5  # bf16 output rounding vs fp32 accumulate
    e = rel_err(got, ref)
    assert e < tol, f"gemm M{M} N{N} K{K} epi{epi}: rel err {e}"
    return e


def check_gemv(be, M, N, K, epi, seed=0):
    rng = np.random.RandomState(seed)
    X = bf16_round(rng.randn(M, K))
    W = bf16_round(rng.randn(N, K) * 0.05)
    Wb = be.bf16(W)
    Wp = be.zeros((N * K,), "bf16")
    _call(be, "vck_pack_weight", Wb, Wp, N, K)
    t = torch.from_numpy(X.astype(np.float64) @ W.T.astype(np.float64)).float()
    if epi == 0:
        out = be.zeros((M, N), "bf16")
    elif epi == 1:
        out = be.zeros((M, N), "f32")
    elif epi == 2:
        r0 = rng.randn(M, N).astype(np.float32)
        out = be.f32(r0)
        t = t + torch.from_numpy(r0)
    else:
        out = be.zeros((M, N // 2), "bf16")
        t = torch.nn.functional.silu(t[:, 0::2]) * t[:, 1::2]
    _call(be, "vck_gemv", be.bf16(X), Wp, out, M, N, K, N // 2 if epi == 3 else N, epi)
    e = rel_err(be.host_f32(out), t.numpy())
    assert e < (2 ** -8 if epi in (0, 3) else 1e-5), f"gemv M{M} N{N} K{K} epi{epi}: rel err {e}"
    return e



def _gemv_ex(be, X, Wp, wscale, out, ssq_in, ssq_out, xg_w, xg_out, npart, M, N, K, ldo, epi, sk=None, ksplit=0):
    """vck_gemv_ex through raw pointers; every array must stay referenced by the caller until be.sync().
    sk = (scratch f32 [ksplit*N/16*2*256], counters i32 [N/16*2]) enables the split-K finisher."""
    be.lib.vck_gemv_ex(be.ptr(X), be.ptr(Wp), be.ptr(wscale), be.ptr(out), be.ptr(ssq_in), be.ptr(ssq_out), be.ptr(xg_w),
                       be.ptr(xg_out), ctypes.c_int(npart), ctypes.c_float(1e-5), be.ptr(sk[0]) if sk else None,
                       be.ptr(sk[1]) if sk else None, ctypes.c_int(ksplit), M, N, K, ldo, epi, None)
    be.sync()


def _rstd(x, eps=1e-5):
    return 1.0 / np.sqrt((x.astype(np.float64) ** 2).mean(-1, keepdims=True) + eps)

def check_gemv_fp8(be, M, N, K, epi, norm=False, seed=0):
    """W8A16: the device quantiser is bit-identical to vcoder_amd/quant.py (bytes, scales, dequantised bf16 rewrite),
    and the byte-streaming GEMV equals X @ W_eff^T."""
    from vcoder_amd import quant

    rng = np.random.RandomState(seed)
    W = rng.randn(N, K) * 0.05 * np.exp2(rng.randint(-6, 4, size=(N, 1)))  # row magnitudes over 10 octaves
    W[rng.randint(N)] = 0.0                                                 # an all-zero row (scale 1)
    W[rng.randint(N), rng.randint(K)] = 37.0                                # an outlier: the rest of that row underflows
    W = bf16_round(W)
    q, s, w_eff = quant.quantize_rows(W)
    Wb, Wq, sc = be.bf16(W), be.zeros((N * K,), "u8"), be.zeros((N,), "f32")
    _call(be, "vck_quantize_fp8", Wb, Wq, sc, N, K)
    got_q = np.asarray(Wq.cpu().numpy() if hasattr(Wq, "cpu") else Wq)
    assert np.array_equal(be.host_f32(sc), s), "row scales differ from quant.row_scales"
    assert np.array_equal(got_q, quant.pack_supertiles(q)), "e4m3 bytes differ from quant.e4m3_encode"
    assert np.array_equal(be.host_f32(Wb), w_eff), "dequantised rewrite differs"
    assert np.array_equal(bf16_round(w_eff), w_eff)
    npart = 16
    ssq = None
    scale_rows = 1.0
    X = bf16_round(rng.randn(M, K))
    if norm:   # consumer form: X is the producer's bf16(x * g); the output is scaled by rstd from the partials
        x32 = (rng.randn(M, K) * 1.5).astype(np.float32)
        g = (rng.rand(K) + 0.5).astype(np.float32)
        X = bf16_round(x32 * g)
        part = np.zeros((16, npart), np.float32)
        part[:M, : npart // 2] = ((x32.astype(np.float64) ** 2).sum(-1) / (npart // 2))[:, None]
        ssq = be.f32(part)
        scale_rows = _rstd(x32)
    Xd = be.bf16(X)
    t = torch.from_numpy((X.astype(np.float64) @ w_eff.T.astype(np.float64)) * scale_rows).float()
    if epi == 0:
        out = be.zeros((M, N), "bf16")
    elif epi == 1:
        out = be.zeros((M, N), "f32")
    elif epi == 2:
        r0 = rng.randn(M, N).astype(np.float32)
        out = be.f32(r0)
        t = t + torch.from_numpy(r0)
    else:
        out = be.zeros((M, N // 2), "bf16")
        t = torch.nn.functional.silu(t[:, 0::2]) * t[:, 1::2]
    _gemv_ex(be, Xd, Wq, sc, out, ssq, None, None, None, npart, M, N, K, N // 2 if epi == 3 else N, epi)
    e = rel_err(be.host_f32(out), t.numpy())
    tol = 2 ** -8 if epi in (0, 3) else 2e-5
    assert e < tol, f"gemv_fp8 M{M} N{N} K{K} epi{epi} norm{norm}: rel err {e}"
    return e


def check_gemm_f8(be, M, N, K, epi, seed=0, ws_mb=0):
    """W8A8 prefill GEMM: the device's activation quantiser equals vcoder_amd/quant.py row for row (bytes + scales), the
    row-major weight bytes equal quant.quantize_rows, and the e4m3 x e4m3 MFMA GEMM equals A_eff @ W_eff^T (exact
    products, fp32 accumulation) through the epilogue."""
    from vcoder_amd import quant

    rng = np.random.RandomState(seed)
    A = rng.randn(M, K) * np.exp2(rng.randint(-4, 5, size=(M, 1)))
    A[rng.randint(M)] = 0.0
    A[rng.randint(M), rng.randint(K)] = 300.0     # a massive activation: the rest of the row loses bits
    A = bf16_round(A)
    W = bf16_round(rng.randn(N, K) * 0.05 * np.exp2(rng.randint(-3, 3, size=(N, 1))))
    qa, sa, a_eff = quant.quantize_rows(A)
    qw, sw, w_eff = quant.quantize_rows(W)
    Ad, Q, sad = be.bf16(A), be.zeros((M, K), "u8"), be.zeros((M,), "f32")
    _call(be, "vck_quant_act_rows", Ad, K, Q, sad, M, K)
    Wb, Wq, swd, Wrow = be.bf16(W), be.zeros((N * K,), "u8"), be.zeros((N,), "f32"), be.zeros((N, K), "u8")
    _call(be, "vck_quantize_fp8_rows", Wb, Wq, swd, Wrow, N, K)
    host = lambda t: np.asarray(t.cpu().numpy() if hasattr(t, "cpu") else t)
    assert np.array_equal(host(Q), qa), "activation bytes differ from quant.quantize_rows"
    assert np.array_equal(be.host_f32(sad), sa), "activation row scales differ"
    assert np.array_equal(host(Wrow), qw), "row-major weight bytes differ from quant.quantize_rows"
    assert np.array_equal(host(Wq), quant.pack_supertiles(qw))
    t = torch.from_numpy(a_eff.astype(np.float64) @ w_eff.T.astype(np.float64)).float()
    if epi == 0:
        out = be.zeros((M, N), "bf16")
    elif epi == 4:
        r0 = rng.randn(M, N).astype(np.float32)
        out = be.f32(r0.copy())
        t = t + torch.from_numpy(r0)
    else:
        out = be.zeros((M, N // 2), "bf16")
        t = torch.nn.functional.silu(t[:, 0::2]) * t[:, 1::2]
    ws = be.zeros((max(ws_mb, 1) << 18,), "f32")
    be.lib.vck_gemm_f8(be.ptr(Q), be.ptr(sad), be.ptr(Wrow), be.ptr(swd), be.ptr(out), M, N, K, N // 2 if epi == 5 else N, epi,
                       be.ptr(ws) if ws_mb else None, ctypes.c_size_t(ws_mb << 20), None)
    be.sync()
    e = rel_err(be.host_f32(out), t.numpy())
    # fp32 outputs: the emulator's MFMA sums the 128 exact products of a step in fp32; the hardware's scaled MFMA aligns them
    # to the largest and keeps ~13 bits below it (tools/experiments/f8_precision.py: 2e-5 of the row maximum on random
    # operands, 1e-4 when one product dominates; measured here 3.2e-5 ... 3.9e-5) -> 1e-4
    tol = 2 ** -8 if epi in (0, 5) else (2e-5 if be.name == "emu" else 1e-4)
    print(f"gemm_f8 M{M} N{N} K{K} epi{epi}: rel err {e:.3e}")
    assert e < tol, f"gemm_f8 M{M} N{N} K{K} epi{epi}: rel err {e}"
    return e


def check_gemv_splitk(be, M, N, K, ksplit, seed=0):
    """Split-K RESID form (o_proj / down): x += h @ W^T with ssq partials and the xg operand published by the tile's last
    arriver; equal to the unsplit launch up to fp32 summation order, bit-identical across repeated launches, and the
    arrival counters are left at zero."""
    rng = np.random.RandomState(seed)
    npart = (N // 16 + 15) // 16 * 16
    X = bf16_round(rng.randn(M, K))
    W = bf16_round(rng.randn(N, K) * 0.05)
    g = (rng.rand(N) + 0.5).astype(np.float32)
    r0 = rng.randn(M, N).astype(np.float32)
    Xd, Wd, gd = be.bf16(X), be.bf16(W), be.f32(g)
    Wp = be.zeros((N * K,), "bf16")
    _call(be, "vck_pack_weight", Wd, Wp, N, K)
    ref = r0.astype(np.float64) + X.astype(np.float64) @ W.T.astype(np.float64)
    scratch, counters = be.zeros((max(ksplit, 1) * (N // 16) * 2 * 256,), "f32"), be.zeros((N // 16 * 2,), "i32")
    outs = []
    for it in range(3):
        out, ssq, xg = be.f32(r0.copy()), be.zeros((16, npart), "f32"), be.zeros((M, N), "bf16")   # f32() may alias its input
        _gemv_ex(be, Xd, Wp, None, out, None, ssq, gd, xg, npart, M, N, K, N, 2, sk=(scratch, counters), ksplit=ksplit)
        got = be.host_f32(out)
        assert rel_err(got, ref) < 1e-5, f"split-K {ksplit}: rel err {rel_err(got, ref)}"
        assert np.array_equal(be.host_f32(xg), bf16_round(got * g))
        assert np.abs(be.host_f32(ssq)[:M, : N // 16].sum(-1) / (ref ** 2).sum(-1) - 1).max() < 1e-5
        assert not be.host_i32(counters).any(), "arrival counters must be re-armed"
        outs.append(got)
    assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[0], outs[2]), "split-K result depends on arrival order"


def check_interleave(be, F, K):
    rng = np.random.RandomState(1)
    g, u = bf16_round(rng.randn(F, K)), bf16_round(rng.randn(F, K))
    out = be.zeros((2 * F, K), "bf16")
    _call(be, "vck_interleave_rows", be.bf16(g), be.bf16(u), out, F, K)
    got = be.host_f32(out)
    assert np.array_equal(got[0::2], g) and np.array_equal(got[1::2], u)


def check_layernorm(be, rows, D, seed=0):
    rng = np.random.RandomState(seed)
    x = (rng.randn(rows, D) * 2 + 0.3).astype(np.float32)
    w, b = (rng.rand(D) + 0.5).astype(np.float32), (rng.randn(D) * 0.1).astype(np.float32)
    y = be.zeros((rows, D), "bf16")
    _call(be, "vck_layernorm", be.f32(x), be.f32(w), be.f32(b), y, rows, D, 1e-5)
    ref = cpu_ref.layer_norm(torch.from_numpy(x), torch.from_numpy(w), torch.from_numpy(b), 1e-5).numpy()
    e = rel_err(be.host_f32(y), ref)
    assert e < 2 ** -8, f"layernorm rel err {e}"


def check_rmsnorm(be, rows, D, gather=False, seed=0):
    rng = np.random.RandomState(seed)
    x = (rng.randn(rows * (3 if gather else 1), D) * 3).astype(np.float32)
    w = (rng.rand(D) + 0.5).astype(np.float32)
    idx = rng.permutation(x.shape[0])[:rows].astype(np.int32) if gather else None
    y = be.zeros((rows, D), "bf16")
    _call(be, "vck_rmsnorm", be.f32(x), be.i32(idx) if gather else None, be.f32(w), y, rows, D, 1e-5)
    src = x[idx] if gather else x
    ref = cpu_ref.rms_norm(torch.from_numpy(src), torch.from_numpy(w), 1e-5).numpy()
    e = rel_err(be.host_f32(y), ref)
    assert e < 2 ** -8, f"rmsnorm rel err {e}"


def check_quant_act_rows_exhaustive(be):
    """vck_quant_act_rows over EVERY finite bf16 value v with |v| <= 448 * s, at three power-of-two row scales s (each row carries
    448 * s so that its scale is s): the bytes the device writes — since round 6 the hardware's v_cvt_pk_fp8_f32 in the register-resident
    kernel — equal the host restatement's (vcoder_amd/quant.py, the software round-to-nearest-even encode), ties, subnormals and
    the largest finite value included."""
    from vcoder_amd import quant

    bits = np.arange(65536, dtype=np.uint32).astype(np.uint32) << 16
    vals = bits.view(np.float32)
    vals = vals[np.isfinite(vals) & (np.abs(vals) <= 448.0) & ((np.abs(vals) >= 2.0 ** -60) | (vals == 0))]   # (the bf16 denormals would not stay bf16 when scaled down)
    K = 4096
    rows = []
    for sc in (1.0, 32.0, 0.125):
        v = (vals * np.float32(sc)).astype(np.float32)            # exact: a power of two
        n = -(-len(v) // (K - 1))
        for r in range(n):
            row = np.zeros(K, np.float32)
            chunk = v[r * (K - 1):(r + 1) * (K - 1)]
            row[0] = 448.0 * sc                                   # pins the row's scale to sc
            row[1:1 + len(chunk)] = chunk
            rows.append(row)
    A = np.stack(rows, 0)
    assert np.array_equal(bf16_round(A), A)
    qa, sa, _ = quant.quantize_rows(A)
    M = A.shape[0]
    Ad, Q, sad = be.bf16(A), be.zeros((M, K), "u8"), be.zeros((M,), "f32")
    _call(be, "vck_quant_act_rows", Ad, K, Q, sad, M, K)
    host = lambda t: np.asarray(t.cpu().numpy() if hasattr(t, "cpu") else t)
    assert np.array_equal(be.host_f32(sad), sa), "row scales differ"
    got = host(Q)
    bad = np.argwhere(got != qa)
    assert bad.size == 0, f"{len(bad)} bytes differ, first at {bad[0]}: value {A[tuple(bad[0])]!r} scale {sa[bad[0][0]]} device {got[tuple(bad[0])]:#x} host {qa[tuple(bad[0])]:#x}"
    return M


def check_rmsnorm_q8(be, rows, D, seed=0):
    """RMSNorm straight into the e4m3 operand: bit-identical to vck_rmsnorm followed by vck_quant_act_rows, and to
    vcoder_amd/quant.py on the kernel's own bf16 row."""
    from vcoder_amd import quant

    rng = np.random.RandomState(seed)
    x = (rng.randn(rows, D) * np.exp2(rng.randint(-3, 4, size=(rows, 1)))).astype(np.float32)
    x[rng.randint(rows), rng.randint(D)] = 500.0
    w = (rng.rand(D) + 0.5).astype(np.float32)
    xd, wd = be.f32(x), be.f32(w)
    y = be.zeros((rows, D), "bf16")
    _call(be, "vck_rmsnorm", xd, None, wd, y, rows, D, 1e-5)
    q2, s2 = be.zeros((rows, D), "u8"), be.zeros((rows,), "f32")
    _call(be, "vck_quant_act_rows", y, D, q2, s2, rows, D)
    q1, s1 = be.zeros((rows, D), "u8"), be.zeros((rows,), "f32")
    _call(be, "vck_rmsnorm_q8", xd, wd, q1, s1, rows, D, 1e-5)
    host = lambda t: np.asarray(t.cpu().numpy() if hasattr(t, "cpu") else t)
    assert np.array_equal(host(q1), host(q2)) and np.array_equal(be.host_f32(s1), be.host_f32(s2)), "fused != two passes"
    qa, sa, _ = quant.quantize_rows(be.host_f32(y))
    assert np.array_equal(host(q1), qa) and np.array_equal(be.host_f32(s1), sa)


class _VitCfg:
    def __init__(self, image, patch):
        self.vit_image_size, self.vit_patch_size = image, patch


def check_im2col(be, n_img, image, patch, Kpad):
    rng = np.random.RandomState(3)
    px = rng.randn(n_img, 3, image, image).astype(np.float32)
    g = image // patch
    cols = be.zeros((n_img * g * g, Kpad), "bf16")
    _call(be, "vck_im2col", be.f32(px), cols, n_img, image, patch, Kpad)
    sd = {"embeddings.patch_embedding.weight": torch.zeros(4, 3, patch, patch)}
    ref, _ = cpu_ref.vit_embed(torch.from_numpy(px), sd, "", _VitCfg(image, patch))
    got = be.host_f32(cols)
    assert np.array_equal(got[:, :3 * patch * patch], bf16_round(ref.numpy()))
    assert not got[:, 3 * patch * patch:].any()


def check_vit_embed_ln(be, n_img, T, D):
    rng = np.random.RandomState(4)
    patches = rng.randn(n_img * (T - 1), D).astype(np.float32)
    cls, pos = rng.randn(D).astype(np.float32), rng.randn(T, D).astype(np.float32)
    w, b = (rng.rand(D) + 0.5).astype(np.float32), (rng.randn(D) * 0.1).astype(np.float32)
    x = be.zeros((n_img, T, D), "f32")
    _call(be, "vck_vit_embed_ln", be.f32(patches), be.f32(cls), be.f32(pos), be.f32(w), be.f32(b), x, n_img, T, D, 1e-5)
    pre = np.concatenate([np.broadcast_to(cls, (n_img, 1, D)), patches.reshape(n_img, T - 1, D)], 1) + pos[None]
    ref = cpu_ref.layer_norm(torch.from_numpy(pre), torch.from_numpy(w), torch.from_numpy(b), 1e-5).numpy()
    e = rel_err(be.host_f32(x), ref)
    assert e < 1e-5, f"vit_embed_ln rel err {e}"


def check_select_rows(be, n_img, T, D):
    rng = np.random.RandomState(5)
    x = rng.randn(n_img, T, D).astype(np.float32)
    y = be.zeros((n_img * (T - 1), D), "bf16")
    _call(be, "vck_select_rows_bf16", be.f32(x), y, n_img, T, 1, D)
    assert np.array_equal(be.host_f32(y), bf16_round(x[:, 1:].reshape(-1, D)))


def rope_tables(max_pos, hd, theta=10000.0):
    cos, sin = cpu_ref.rope_cos_sin(torch.arange(max_pos), hd, theta)
    return cos[:, : hd // 2].contiguous().numpy(), sin[:, : hd // 2].contiguous().numpy()


def _split_ref(qkv, B, T, H, hd, rope, pos0=0):
    D = H * hd
    t = torch.from_numpy(qkv).reshape(B, T, 3, H, hd).permute(2, 0, 3, 1, 4)  # [3,B,H,T,hd]
    q, k, v = t[0], t[1], t[2]
    if rope:
        cos, sin = cpu_ref.rope_cos_sin(torch.arange(pos0, pos0 + T), hd, 10000.0)
        q = cpu_ref.apply_rope(q, cos, sin)
        k = cpu_ref.apply_rope(k, cos, sin)
    return bf16_round(q.numpy()), bf16_round(k.numpy()), v.numpy()


def vt_key_pos(Ts):
    """key order of the flash kernel's V^T scratch (csrc/attn.hip vt_chunk_key0): stored[..., p] = natural[..., vt_key_pos(Ts)[p]];
    inside every aligned 32-key block position 8c + e holds key 4c + (e & 3) + 16 (e >> 2)."""
    pos = np.arange(Ts)
    c, e = (pos % 32) // 8, pos % 8
    return (pos // 32) * 32 + 4 * c + (e & 3) + 16 * (e >> 2)


def check_qkv_split(be, B, T, H, hd, rope):
    rng = np.random.RandomState(6)
    D = H * hd
    qkv = bf16_round(rng.randn(B * T, 3 * D))
    Ts = (T + 63) // 64 * 64
    q, k, vt = be.zeros((B, H, Ts, hd), "bf16"), be.zeros((B, H, Ts, hd), "bf16"), be.zeros((B, H, hd, Ts), "bf16")
    cos, sin = rope_tables(Ts, hd)
    _call(be, "vck_qkv_split", be.bf16(qkv), q, k, vt, B, T, H, hd, Ts, Ts, None,
          be.f32(cos) if rope else None, be.f32(sin) if rope else None)
    rq, rk, rv = _split_ref(qkv, B, T, H, hd, rope)
    gq, gk, gv = be.host_f32(q), be.host_f32(k), be.host_f32(vt)
    assert np.abs(gq[:, :, :T] - rq).max() <= 2 ** -7 * np.abs(rq).max()
    assert np.abs(gk[:, :, :T] - rk).max() <= 2 ** -7 * np.abs(rk).max()
    if not rope:
        assert np.array_equal(gq[:, :, :T], rq) and np.array_equal(gk[:, :, :T], rk)
    rvt = np.zeros_like(gv)
    rvt[:, :, :, :T] = rv.transpose(0, 1, 3, 2)
    assert np.array_equal(gv, rvt[..., vt_key_pos(Ts)])              # the scratch's key order; zeros behind T
    assert not gq[:, :, T:].any()
    # the LLM prefill form: K and V rows into a cache of S_cap keys, V^T into a scratch of its own stride
    S_cap = Ts + 64
    k2, v2 = be.zeros((B, H, S_cap, hd), "bf16"), be.zeros((B, H, S_cap, hd), "bf16")
    q2, vt2 = be.zeros((B, H, Ts, hd), "bf16"), be.zeros((B, H, hd, Ts), "bf16")
    qd, cd, sd = be.bf16(qkv), (be.f32(cos) if rope else None), (be.f32(sin) if rope else None)
    be.lib.vck_qkv_split_kv(be.ptr(qd), be.ptr(q2), be.ptr(k2), be.ptr(v2), be.ptr(vt2), B, T, H, hd, Ts, S_cap, Ts, be.ptr(cd),
                            be.ptr(sd), None)
    be.sync()
    assert np.array_equal(be.host_f32(k2)[:, :, :T], gk[:, :, :T]) and np.array_equal(be.host_f32(q2), gq)
    assert np.array_equal(be.host_f32(v2)[:, :, :T], rv) and not be.host_f32(v2)[:, :, T:].any()
    assert np.array_equal(be.host_f32(vt2), gv)


def check_gemm_qkv_fused(be, B, T, H, K, bias=False, ws_mb=0, f8=False, kv8=False, seed=0):
    """EPI_QKV (round 6): the QKV GEMM whose epilogue applies RoPE, splits the heads and writes Q, the K / V cache rows
    (bf16 and / or e4m3) and the V^T scratch — against the two launches it replaces (vck_gemm EPI_BF16 + vck_qkv_split_kv / _kv8):
    every output BIT FOR BIT, nothing written behind T, for sample lengths that are and are not multiples of 32 (padded token
    rows), ragged last tiles, split-K remainder rounds, the e4m3 operand form."""
    from vcoder_amd import quant

    rng = np.random.RandomState(seed)
    hd, D = 128, H * 128
    M = B * T
    A = bf16_round(rng.randn(M, K))
    W = bf16_round(rng.randn(3 * D, K) * 0.05)
    b = (rng.randn(3 * D) * 0.1).astype(np.float32) if bias else None
    Ts = (T + 63) // 64 * 64
    S_cap = Ts + 64
    cos, sin = rope_tables(S_cap, hd)
    cd, sd = be.f32(cos), be.f32(sin)
    ws = be.zeros((max(ws_mb, 1) << 18,), "f32")
    wsp, wsb = (be.ptr(ws) if ws_mb else None), ctypes.c_size_t(ws_mb << 20)
    qkv = be.zeros((M, 3 * D), "bf16")
    if f8:
        Ad, Q, sad = be.bf16(A), be.zeros((M, K), "u8"), be.zeros((M,), "f32")
        _call(be, "vck_quant_act_rows", Ad, K, Q, sad, M, K)
        Wb, Wq, swd, Wrow = be.bf16(W), be.zeros((3 * D * K,), "u8"), be.zeros((3 * D,), "f32"), be.zeros((3 * D, K), "u8")
        _call(be, "vck_quantize_fp8_rows", Wb, Wq, swd, Wrow, 3 * D, K)
        be.lib.vck_gemm_f8(be.ptr(Q), be.ptr(sad), be.ptr(Wrow), be.ptr(swd), be.ptr(qkv), M, 3 * D, K, 3 * D, 0, wsp, wsb, None)
        Aop, Wop, asc, wsc = Q, Wrow, sad, swd
    else:
        Aop, Wop, asc, wsc = be.bf16(A), be.bf16(W), None, None
        bd = be.f32(b) if bias else None
        be.lib.vck_gemm_ws(be.ptr(Aop), be.ptr(Wop), be.ptr(bd), be.ptr(qkv), M, 3 * D, K, K, K, 3 * D, 0, wsp, wsb, None)
    be.sync()
    mk = lambda: (be.zeros((B, H, Ts, hd), "bf16"), be.zeros((B, H, S_cap, hd), "bf16"), be.zeros((B, H, S_cap, hd), "bf16"),
                  be.zeros((B, H, hd, Ts), "bf16"), be.zeros((B, H, S_cap, hd), "u8"), be.zeros((B, H, S_cap, hd), "u8"))
    q1, k1, v1, vt1, k81, v81 = mk()
    q2, k2, v2, vt2, k82, v82 = mk()
    if kv8:   # e4m3 cache rows; the bf16 K rows go to a scratch of stride Ts (the flash kernel's operand)
        k1, k2 = be.zeros((B, H, Ts, hd), "bf16"), be.zeros((B, H, Ts, hd), "bf16")
        be.lib.vck_qkv_split_kv8(be.ptr(qkv), be.ptr(q1), be.ptr(k1), be.ptr(k81), be.ptr(v81), be.ptr(vt1), B, T, H, hd, Ts, Ts, Ts,
                                 S_cap, be.ptr(cd), be.ptr(sd), None)
    else:
        be.lib.vck_qkv_split_kv(be.ptr(qkv), be.ptr(q1), be.ptr(k1), be.ptr(v1), be.ptr(vt1), B, T, H, hd, Ts, S_cap, Ts, be.ptr(cd),
                                be.ptr(sd), None)
    be.sync()
    bd = be.f32(b) if (bias and not f8) else None
    be.lib.vck_gemm_qkv(be.ptr(Aop), be.ptr(asc), be.ptr(Wop), be.ptr(wsc), be.ptr(bd), B, T, H, K, K, be.ptr(q2), be.ptr(k2),
                        None if kv8 else be.ptr(v2), be.ptr(vt2), be.ptr(k82) if kv8 else None, be.ptr(v82) if kv8 else None,
                        Ts, Ts if kv8 else S_cap, Ts, S_cap, be.ptr(cd), be.ptr(sd), int(f8), wsp, wsb, None)
    be.sync()
    host = lambda t: np.asarray(t.cpu().numpy() if hasattr(t, "cpu") else t)
    # With a split-K workspace and T off the 32-token grid the padded and the plain token rows sit in different tiles, so a row may
    # be summed in K-slices by one launch and in one piece by the other: fp32 summation order, i.e. single bf16 roundings, may differ
    exact = not (ws_mb and T % 32)
    for name, a1, a2 in (("q", q1, q2), ("k", k1, k2), ("v", v1, v2), ("vt", vt1, vt2)):
        g1, g2 = be.host_f32(a1), be.host_f32(a2)
        if exact:
            assert np.array_equal(g1, g2), f"fused QKV epilogue: {name} differs from gemm + qkv_split (B{B} T{T} H{H} K{K} f8={f8} kv8={kv8}): " \
                                           f"{int((g1 != g2).sum())} of {g1.size} elements, max {np.abs(g1 - g2).max()}"
        else:
            assert np.array_equal(g1 == 0, g2 == 0) or np.abs(g1 - g2).max() <= 2.0 ** -7 * np.abs(g1).max()
            assert np.abs(g1 - g2).max() <= 2.0 ** -7 * np.abs(g1).max() and (g1 != g2).mean() < 0.02, \
                f"fused QKV epilogue: {name}: {(g1 != g2).mean():.4f} of the elements differ, max {np.abs(g1 - g2).max()}"
    if exact:
        assert np.array_equal(host(k81), host(k82)) and np.array_equal(host(v81), host(v82)), "fused QKV epilogue: e4m3 cache rows differ"
    else:
        assert (host(k81) != host(k82)).mean() < 0.02 and (host(v81) != host(v82)).mean() < 0.02
    assert be.host_f32(q2).any() and be.host_f32(vt2).any()


def check_attention(be, B, H, T, hd, causal, seed=0, spike=False):
    rng = np.random.RandomState(seed)
    Ts = (T + 63) // 64 * 64
    q = bf16_round(rng.randn(B, H, T, hd))
    k = bf16_round(rng.randn(B, H, T, hd))
    v = bf16_round(rng.randn(B, H, T, hd))
    if spike:  # force a late running-max jump (online-softmax rescale branch)
        k[:, :, T - 3] = q[:, :, T - 1] * 4
    qp, kp = np.zeros((B, H, Ts, hd), np.float32), np.zeros((B, H, Ts, hd), np.float32)
    vtp = np.zeros((B, H, hd, Ts), np.float32)
    qp[:, :, :T], kp[:, :, :T], vtp[:, :, :, :T] = q, k, v.transpose(0, 1, 3, 2)
    out = be.zeros((B * T, H * hd), "bf16")
    scale = 1.0 / math.sqrt(hd)
    vtp = vtp[..., vt_key_pos(Ts)]
    _call(be, "vck_attention", be.bf16(qp), be.bf16(kp), be.bf16(vtp), out, B, H, T, hd, Ts, Ts, int(causal), scale)
    ref = cpu_ref.softmax_attention(torch.from_numpy(q), torch.from_numpy(k), torch.from_numpy(v), scale, causal,
                                    cpu_ref.Rounder(True))
    ref = ref.transpose(1, 2).reshape(B * T, H * hd).numpy()
    got = be.host_f32(out)
    err = np.abs(got - ref).max()
    assert err < 2 ** -7 * max(1.0, np.abs(ref).max()), f"attention B{B} H{H} T{T} hd{hd} causal{causal}: abs err {err}"
    return err


def check_splice(be, D):
    rng = np.random.RandomState(8)
    V, NF = 50, 12
    embed, feats = bf16_round(rng.randn(V, D)), bf16_round(rng.randn(NF, D))
    rows = [(0, 3), (0, 49), (1, 0), (1, 11), (2, 0), (0, 7), (1, 5)]
    x = be.f32(rng.randn(len(rows), D))
    _call(be, "vck_splice", be.i32(np.array(rows).reshape(-1)), len(rows), be.bf16(embed), be.bf16(feats), x, D)
    got = be.host_f32(x)
    for r, (kind, src) in enumerate(rows):
        exp = embed[src] if kind == 0 else feats[src] if kind == 1 else np.zeros(D, np.float32)
        assert np.array_equal(got[r], exp)
    tok = [4, 9, 0]
    x2 = be.zeros((3, D), "f32")
    _call(be, "vck_embed_tokens", be.i32(tok), be.bf16(embed), x2, 3, D)
    assert np.array_equal(be.host_f32(x2), embed[tok])


def check_greedy(be, B, V):
    rng = np.random.RandomState(9)
    lg = rng.randn(B, V).astype(np.float32)
    lg[0, 17] = lg[0, 5] = lg[0].max() + 1.0       # tie -> lowest index
    lg[1, V - 1] = lg[1].max() + 2.0
    eos, pad, max_new = 2, 0, 4
    if B > 2:
        lg[2, eos] = lg[2].max() + 1.0               # row 2 emits EOS at step 0 then pads
    nxt, out, fin = be.zeros((B,), "i32"), be.zeros((B, max_new), "i32"), be.zeros((B,), "i32")
    step = be.i32([0])
    for _ in range(2):
        _call(be, "vck_greedy", be.f32(lg), nxt, out, fin, step, B, V, max_new, eos, pad)
        _call(be, "vck_advance", step, None, None)
    o = be.host_i32(out)
    exp = torch.argmax(torch.from_numpy(lg), -1).numpy()
    assert o[0, 0] == 5 and o[1, 0] == V - 1
    for b in range(B):
        assert o[b, 0] == exp[b]
        assert o[b, 1] == (pad if exp[b] == eos else exp[b])
    assert be.host_i32(step)[0] == 2


def check_synth(be, name="model.layers.3.mlp.up_proj.weight", n=5000):
    ts = synth.tensor_seed(name, 42)
    hw = 0.02 * math.sqrt(3)
    ob, of = be.zeros((n,), "bf16"), be.zeros((n,), "f32")
    lib = be.lib
    lib.vck_synth_bf16(be.ptr(ob), ctypes.c_uint64(n), ctypes.c_uint32(ts), ctypes.c_float(0.0), ctypes.c_float(hw), None)
    lib.vck_synth_f32(be.ptr(of), ctypes.c_uint64(n), ctypes.c_uint32(ts), ctypes.c_float(1.0), ctypes.c_float(0.1), None)
    be.sync()
    assert np.array_equal(be.host_f32(ob), synth.synth_tensor(name, (n,), 42, 0.0, hw))
    assert np.array_equal(be.host_f32(of), synth.synth_tensor(name, (n,), 42, 1.0, 0.1))
    # the value classes of the reference's checkpoints: fp16-valued (normal range, and a width that reaches fp16 subnormals) / fp32
    for rounding, off, w in (("fp16", 0.0, hw), ("fp16", 0.0, 1e-4), ("fp16", 1.0, 0.1), ("fp32", 0.0, hw)):
        o = be.zeros((n,), "f32")
        lib.vck_synth_f32_rounded(be.ptr(o), ctypes.c_uint64(n), ctypes.c_uint32(ts), ctypes.c_float(off), ctypes.c_float(w),
                                  synth.ROUNDING_CODE[rounding], None)
        be.sync()
        assert np.array_equal(be.host_f32(o), synth.synth_tensor(name, (n,), 42, off, w, rounding)), (rounding, off, w)


# ---- fused decode-step kernels ------------------------------------------------------------------------------------
def check_gemv_norm_chain(be, M, D, N, seed=0):
    """The decode-step chain with RMSNorm folded across producer and consumer:
    embed_tokens_ssq (x, partials, xg = bf16(x*g1)) -> consumer GEMV (rstd * xg @ W1^T) -> RESID GEMV (x += h @ Wo^T,
    new partials, xg = bf16(x*g2)) -> consumer GEMV on the new rows."""
    rng = np.random.RandomState(seed)
    V = 64
    npart = (D // 16 + 15) // 16 * 16
    embed = bf16_round(rng.randn(V, D))
    tok = rng.randint(0, V, size=M).astype(np.int32)
    g1 = (rng.rand(D) + 0.5).astype(np.float32)
    g2 = (rng.rand(D) + 0.5).astype(np.float32)
    W1 = bf16_round(rng.randn(N, D) * 0.05)
    Wo = bf16_round(rng.randn(D, N) * 0.05)
    Mp = (M + 15) // 16 * 16
    x = be.zeros((Mp, D), "f32")
    xg = be.bf16(rng.randn(Mp, D))              # garbage: the kernels must fully overwrite the valid rows
    ssq = be.f32(rng.randn(Mp, npart))
    tokd, embd, g1d, g2d = be.i32(tok), be.bf16(embed), be.f32(g1), be.f32(g2)
    be.lib.vck_embed_tokens_ssq(be.ptr(tokd), be.ptr(embd), be.ptr(x), be.ptr(ssq), be.ptr(g1d), be.ptr(xg), M, D, npart, None)
    be.sync()
    x0 = embed[tok]
    assert np.array_equal(be.host_f32(x)[:M], x0)
    assert np.array_equal(be.host_f32(xg)[:M], bf16_round(x0 * g1)), "xg operand of the embedding kernel"
    W1p, Wop = be.zeros((N * D,), "bf16"), be.zeros((N * D,), "bf16")
    W1d, Wod = be.bf16(W1), be.bf16(Wo)
    _call(be, "vck_pack_weight", W1d, W1p, N, D)
    _call(be, "vck_pack_weight", Wod, Wop, D, N)
    # consumer: h = rstd * (xg @ W1^T)
    h = be.zeros((M, N), "bf16")
    _gemv_ex(be, xg, W1p, None, h, ssq, None, None, None, npart, M, N, D, N, 0)
    ref_h = (bf16_round(x0 * g1).astype(np.float64) @ W1.T.astype(np.float64)) * _rstd(x0)
    e = rel_err(be.host_f32(h), ref_h)
    assert e < 2 ** -8, f"consumer gemv rel err {e}"
    # producer: x += h @ Wo^T, partials of the new rows, xg = bf16(x_new * g2)
    hb = be.host_f32(h)
    _gemv_ex(be, h, Wop, None, x, None, ssq, g2d, xg, npart, M, D, N, D, 2)
    x_new = x0.astype(np.float64) + hb.astype(np.float64) @ Wo.T.astype(np.float64)
    got_x = be.host_f32(x)[:M]
    assert rel_err(got_x, x_new) < 1e-5
    got_ss = be.host_f32(ssq)[:M, : D // 16].sum(-1)
    assert np.abs(got_ss / (x_new ** 2).sum(-1) - 1).max() < 1e-5
    assert np.array_equal(be.host_f32(xg)[:M], bf16_round(got_x * g2)), "xg operand of the RESID epilogue"
    # second consumer on the published partials
    y = be.zeros((M, N), "f32")
    _gemv_ex(be, xg, W1p, None, y, ssq, None, None, None, npart, M, N, D, N, 1)
    ref_y = (bf16_round(got_x * g2).astype(np.float64) @ W1.T.astype(np.float64)) * _rstd(got_x)
    e = rel_err(be.host_f32(y), ref_y)
    assert e < 2e-5, f"second consumer gemv rel err {e}"


def check_attention_decode_fused(be, B, H, hd, pos, seed=0, per_row=False):
    """fused decode attention (RoPE + append + attention over the key-major K / V cache).  per_row: every row at its own
    position, one row inactive (the decode pool's form, vck_attention_decode_rows)."""
    rng = np.random.RandomState(seed)
    D = H * hd
    poss = [max(1, pos - 13 * b) for b in range(B)] if per_row else [pos] * B
    S = (pos + 1 + 63) // 64 * 64 + 64
    qkv = bf16_round(rng.randn(B, 3 * D))
    k_old = bf16_round(rng.randn(B, H, S, hd))
    v_old = bf16_round(rng.randn(B, H, S, hd))          # beyond a row's position: stale garbage that must not be read
    kd, vd = be.bf16(k_old), be.bf16(v_old)
    out = be.zeros((B, D), "bf16")
    cos, sin = rope_tables(S, hd)
    scale = 1.0 / math.sqrt(hd)
    qd, cd, sd = be.bf16(qkv), be.f32(cos), be.f32(sin)   # keep alive until sync()
    inactive = B - 1 if (per_row and B > 1) else -1
    if per_row:
        rows = np.zeros((B, 4), np.int32)
        rows[:, 0] = 1
        rows[:, 1] = poss
        if inactive >= 0:
            rows[inactive, 0] = 0
        rd = be.i32(rows)
        base = rd.ctypes.data if isinstance(rd, np.ndarray) else rd.data_ptr()
        be.lib.vck_attention_decode_rows(be.ptr(qd), be.ptr(kd), be.ptr(vd), be.ptr(out), B, H, hd, S, c_p(base + 4), 4,
                                         c_p(base), be.ptr(cd), be.ptr(sd), ctypes.c_float(scale), None)
    else:
        pd = be.i32([pos])
        be.lib.vck_attention_decode_fused(be.ptr(qd), be.ptr(kd), be.ptr(vd), be.ptr(out), B, H, hd, S, be.ptr(pd),
                                          be.ptr(cd), be.ptr(sd), ctypes.c_float(scale), None)
    be.sync()
    gk, gv, got = be.host_f32(kd), be.host_f32(vd), be.host_f32(out)
    for b in range(B):
        pb = poss[b]
        if b == inactive:
            assert np.array_equal(gk[b], k_old[b]) and np.array_equal(gv[b], v_old[b]) and not got[b].any()
            continue
        rq, rk, rv = _split_ref(qkv[b:b + 1], 1, 1, H, hd, True, pos0=pb)   # roped+rounded q,k and raw v of the new token
        assert np.abs(gk[b, :, pb] - rk[0, :, 0]).max() <= 2 ** -7 * np.abs(rk).max()
        assert np.array_equal(gv[b, :, pb], rv[0, :, 0])
        keep = np.ones(S, bool)
        keep[pb] = False
        assert np.array_equal(gk[b][:, keep], k_old[b][:, keep]) and np.array_equal(gv[b][:, keep], v_old[b][:, keep])
        k_all = np.concatenate([k_old[b:b + 1, :, :pb], gk[b:b + 1, :, pb:pb + 1]], 2)
        v_all = np.concatenate([v_old[b:b + 1, :, :pb], rv], 2)
        ref = cpu_ref.softmax_attention(torch.from_numpy(rq), torch.from_numpy(k_all), torch.from_numpy(v_all), scale, False,
                                        cpu_ref.Rounder(False))
        ref = ref.transpose(1, 2).reshape(D).numpy()
        err = np.abs(got[b] - ref).max()
        assert err < 2 ** -7 * max(1.0, np.abs(ref).max()), f"attention_decode_fused row {b} pos {pb}: abs err {err}"


RS = dict(ACTIVE=0, FINISHED=1, STEP=2, POS=3, MAXNEW=4, EOS=5, PAD=6, NSTOP=7, SAMPLE=8, INVTEMP=9, TOPK=10, TOPP=11,
          SEED_LO=12, SEED_HI=13, OUT_OFF=14, TAIL=16, STOP=24, STRIDE=128)   # csrc/kernels.h RowStateField


def _f32_bits(x: float) -> int:
    return int(np.float32(x).view(np.int32))


def make_rows(n, **fields):
    """host-side RowState records [n, 128] int32; a field value is a scalar or a per-row list"""
    r = np.zeros((n, RS["STRIDE"]), dtype=np.int32)
    r[:, RS["ACTIVE"]] = 1
    r[:, RS["EOS"]] = -1
    r[:, RS["INVTEMP"]] = _f32_bits(1.0)
    r[:, RS["TOPP"]] = _f32_bits(1.0)
    r[:, RS["TAIL"]:RS["TAIL"] + 7] = np.iinfo(np.int32).min
    for k, v in fields.items():
        r[:, RS[k]] = v
    return r


def _select(be, lg, rows, out, D, V, embed=None, advance=3, gw=None):
    """one launch of vck_select_embed; returns (next_tok, rows, x, xg, ssq) on the host"""
    n = rows.shape[0]
    npart = (D // 16 + 15) // 16 * 16
    nxt = be.zeros((n,), "i32")
    x, ssq, xg = be.zeros((n, D), "f32"), be.zeros((n, npart), "f32"), be.zeros((n, D), "bf16")
    lgd, rd = be.f32(lg), be.i32(rows)
    embd = be.bf16(embed) if embed is not None else None
    gwd = be.f32(gw if gw is not None else np.ones(D, np.float32))
    be.lib.vck_select_embed(be.ptr(lgd), V, be.ptr(rd), be.ptr(nxt), be.ptr(out), be.ptr(embd), be.ptr(x), be.ptr(ssq),
                            be.ptr(gwd), be.ptr(xg), D, npart, V, n, advance, None)
    be.sync()
    return be.host_i32(nxt), be.host_i32(rd), be.host_f32(x), be.host_f32(xg), be.host_f32(ssq)


def check_select_embed(be, B, V, D):
    """greedy selection (lowest index on ties), EOS -> pad bookkeeping, inactive rows, per-row step / position advance,
    embedding + RMSNorm partials + xg of the selected token ([HF] generation/utils.py:2894,2925-2929)."""
    assert int(be.lib.vck_row_state_stride()) == RS["STRIDE"]
    rng = np.random.RandomState(11)
    lg = rng.randn(B, V).astype(np.float32)
    lg[0, 17] = lg[0, 5] = lg[0].max() + 1.0
    eos, pad, max_new = 2, 0, 4
    if B > 2:
        lg[2, eos] = lg[2].max() + 1.0
    embed = bf16_round(rng.randn(V, D))
    gw = (rng.rand(D) + 0.5).astype(np.float32)
    rows = make_rows(B, MAXNEW=max_new, EOS=eos, PAD=pad, POS=[100 + 3 * b for b in range(B)],
                     OUT_OFF=[b * max_new for b in range(B)])
    if B > 1:
        rows[1, RS["ACTIVE"]] = 0        # a free row: nothing of it may change
    out = be.zeros((B, max_new), "i32")
    nxt, r1, x, xg, ssq = _select(be, lg, rows, out, D, V, embed, advance=1, gw=gw)
    nxt, r2, x, xg, ssq = _select(be, lg, r1, out, D, V, embed, advance=3, gw=gw)
    o = be.host_i32(out)
    exp = np.argmax(lg, -1)
    assert o[0, 0] == 5
    for b in range(B):
        if B > 1 and b == 1:
            assert (o[b] == 0).all() and np.array_equal(r2[b], rows[b])
            continue
        assert o[b, 0] == exp[b] and o[b, 1] == (pad if exp[b] == eos else exp[b])
        assert r2[b, RS["STEP"]] == 2 and r2[b, RS["POS"]] == 100 + 3 * b + 1
        assert r2[b, RS["FINISHED"]] == int(exp[b] == eos)
        last = pad if exp[b] == eos else exp[b]
        assert nxt[b] == last
        assert np.array_equal(x[b], embed[last])
        assert np.array_equal(xg[b], bf16_round(embed[last] * gw))
        assert abs(ssq[b].sum() / (embed[last] ** 2).sum() - 1) < 1e-5


def sample_reference_probs(lg, temperature, top_k, top_p):
    """HF warper order (logits_process.py): temperature -> top-k -> top-p, then softmax; fp64."""
    z = lg.astype(np.float64) / temperature
    if top_k and 0 < top_k < z.shape[-1]:
        kth = np.sort(z)[-top_k]
        z = np.where(z < kth, -np.inf, z)
    if top_p < 1.0:
        order = np.argsort(z, kind="stable")                    # ascending
        p = np.exp(z[order] - z.max())
        p /= p.sum()
        remove = np.cumsum(p) <= (1.0 - top_p)
        remove[-1] = False
        z = z.copy()
        z[order[remove]] = -np.inf
    p = np.exp(z - z.max())
    return p / p.sum()


def check_uniform_extremes(be):
    """the sampler's hash -> uniform map stays strictly inside (0, 1) for every 32-bit hash — including the extremes
    (with 24 random bits, 16777215.5 rounded to 2^24 and u == 1.0 made the Gumbel term +inf: a uniformly random token)"""
    h = np.array([0, 1, 0x1FF, 0x200, 0x7FFFFFFF, 0x80000000, 0xFFFFFE00, 0xFFFFFF00, 0xFFFFFFFE, 0xFFFFFFFF] +
                 list(np.random.RandomState(3).randint(0, 2 ** 32, size=246, dtype=np.uint64)), dtype=np.uint32)
    n = h.shape[0]
    hd = be.i32(h.view(np.int32))
    u, g = be.zeros((n,), "f32"), be.zeros((n,), "f32")
    _call(be, "vck_uniform_probe", hd, u, g, n)
    be.sync()
    u, g = be.host_f32(u), be.host_f32(g)
    assert (u > 0).all() and (u < 1).all(), f"uniform left (0,1): min {u.min()} max {u.max()}"
    assert np.isfinite(g).all(), "Gumbel term not finite"
    exp = ((h >> 9).astype(np.float64) + 0.5) / 8388608.0
    assert np.array_equal(u, exp.astype(np.float32))
    assert u[9] == np.float32(1.0 - 2.0 ** -24) and u[0] == np.float32(2.0 ** -24)


def check_gemv_rows_agree_across_variants(be, N, K, epi, norm=True, ksplit=0, seed=0, fp8=False):
    """The decode pool's promise: a row gets bit-for-bit the same result from a 32-row pass (two MFMA row groups) as from a
    16-row pass — rows 0..15 and 16..31 of an M = 29 launch against two launches of 16 and 13 rows."""
    rng = np.random.RandomState(seed)
    M = 29
    npart = (K // 16 + 15) // 16 * 16
    X = bf16_round(rng.randn(32, K))
    W = bf16_round(rng.randn(N, K) * 0.05)
    wsc = None
    if fp8:   # W8A16: e4m3 super-tiles + row scales; the reference uses the dequantised matrix the quantiser leaves behind
        Wd, Wp, wsc = be.bf16(W), be.zeros((N * K,), "u8"), be.zeros((N,), "f32")
        _call(be, "vck_quantize_fp8", Wd, Wp, wsc, N, K)
        W = be.host_f32(Wd)
    else:
        Wd, Wp = be.bf16(W), be.zeros((N * K,), "bf16")
        _call(be, "vck_pack_weight", Wd, Wp, N, K)
    ssq = np.zeros((32, npart), np.float32)
    ssq[:, : K // 16] = rng.rand(32, K // 16).astype(np.float32) + 0.5
    ssqd = be.f32(ssq) if norm else None
    esz = {0: "bf16", 1: "f32", 2: "f32", 3: "bf16"}[epi]
    No = N // 2 if epi == 3 else N
    r0 = rng.randn(32, No).astype(np.float32)
    mk = lambda: (be.f32(r0.copy()) if epi == 2 else be.zeros((32, No), esz))
    sk = None
    if ksplit:
        sk = (be.zeros((ksplit * (N // 16) * 2 * 256,), "f32"), be.zeros((N // 16 * 2,), "i32"))
    Xd = be.bf16(X)
    big = mk()
    _gemv_ex(be, Xd, Wp, wsc, big, ssqd, None, None, None, npart, M, N, K, No, epi, sk=sk, ksplit=ksplit)
    be.sync()
    lo, hi = mk(), mk()
    _gemv_ex(be, Xd, Wp, wsc, lo, ssqd, None, None, None, npart, 16, N, K, No, epi, sk=sk, ksplit=ksplit)
    be.sync()
    Xh = be.bf16(X[16:])
    ssqh = be.f32(ssq[16:]) if norm else None
    _gemv_ex(be, Xh, Wp, wsc, hi, ssqh, None, None, None, npart, M - 16, N, K, No, epi, sk=sk, ksplit=ksplit)
    be.sync()
    b, l = be.host_f32(big), be.host_f32(lo)
    same = np.array_equal   # the K partition is the same for both row counts
    assert same(b[:16], l[:16]), "rows 0..15 differ between the 32-row and the 16-row pass"
    if epi != 2:   # (the in-place residual form adds into whatever rows the buffer holds: only rows 0..15 line up)
        assert same(b[16:M], be.host_f32(hi)[: M - 16]), "rows 16..28 differ between the 32-row and the 16-row pass"
    ref = X[:M].astype(np.float64) @ W.T.astype(np.float64)
    if norm:
        ref = ref / np.sqrt(ssq[:M, : K // 16].sum(-1, keepdims=True) / K + 1e-5)
    if epi == 3:
        t = torch.from_numpy(ref).float()
        ref = (torch.nn.functional.silu(t[:, 0::2]) * t[:, 1::2]).numpy()
    if epi == 2:
        ref = ref + r0[:M]
    e = rel_err(b[:M], ref)
    assert e < (2 ** -8 if epi in (0, 3) else 2e-5), f"gemv M=29 N{N} K{K} epi{epi}: rel err {e}"


def check_sampling(be, V, temperature, top_k, top_p, draws=2048, seed0=1234):
    """device sampling (vck_select_embed, RS_SAMPLE): (i) the same seed gives the same token, (ii) no draw ever leaves the
    top-k / top-p support of the reference warpers, (iii) the empirical distribution matches softmax of the warped logits
    (total variation < 4 sigma of the multinomial noise), (iv) the step counter is part of the key."""
    rng = np.random.RandomState(5)
    lg1 = (rng.randn(V) * 1.5).astype(np.float32)
    probs = sample_reference_probs(lg1, temperature, top_k, top_p)
    # the nucleus boundary is a floating-point comparison of a cumulative sum (fp32, block order on the device; fp64 here):
    # a token sitting exactly on it may fall on either side, so "left the support" is judged against a hair wider nucleus
    support = sample_reference_probs(lg1, temperature, top_k, min(1.0, top_p + 2e-3)) > 0 if top_p < 1.0 else probs > 0
    R = 16
    lg = np.tile(lg1, (R, 1))
    counts = np.zeros(V, dtype=np.int64)
    first = None
    out = be.zeros((R, 1), "i32")
    for it in range(draws // R):
        rows = make_rows(R, MAXNEW=1, SAMPLE=1, INVTEMP=_f32_bits(1.0 / temperature), TOPK=top_k, TOPP=_f32_bits(top_p),
                         SEED_LO=[(seed0 + 7919 * (it * R + r)) & 0x7FFFFFFF for r in range(R)], SEED_HI=77,
                         OUT_OFF=list(range(R)))
        nxt, _, _, _, _ = _select(be, lg, rows, out, 64, V, None, advance=0)
        if it == 0:
            first = nxt.copy()
            again, _, _, _, _ = _select(be, lg, rows, out, 64, V, None, advance=0)
            assert np.array_equal(first, again), "same seed, same step -> same token"
            rows2 = rows.copy()
            rows2[:, RS["STEP"]] = 1
            other, _, _, _, _ = _select(be, lg, rows2, out, 64, V, None, advance=0)
            if support.sum() > 4:
                assert not np.array_equal(first, other), "the step counter must change the draw"
        assert support[nxt].all(), f"a draw left the top-k/top-p support: {nxt[~support[nxt]]}"
        np.add.at(counts, nxt, 1)
    n = counts.sum()
    tv = 0.5 * np.abs(counts / n - probs).sum()
    # E[TV] of an n-sample multinomial ~ sum_i sqrt(p_i (1 - p_i) / (2 pi n)) ; allow 4x
    bound = 4.0 * np.sqrt(probs * (1 - probs) / (2 * np.pi * n)).sum() + 1e-3
    assert tv < bound, f"sampling T={temperature} k={top_k} p={top_p}: total variation {tv:.4f} > {bound:.4f}"
    return tv


# ---- strict (fp32) kernels ----------------------------------------------------------------------------------------
def check_gemm_f32(be, M, N, K, epi, bias=True, seed=0):
    rng = np.random.RandomState(seed)
    A = rng.randn(M, K).astype(np.float32)
    W = bf16_round(rng.randn(N, K) * 0.05)
    b = (rng.randn(N) * 0.1).astype(np.float32) if bias else None
    t = torch.from_numpy(A.astype(np.float64) @ W.T.astype(np.float64) + (b if bias else 0.0)).float()
    r0 = rng.randn(M, N).astype(np.float32)
    out = be.f32(r0) if epi == 4 else be.zeros((M, N // 2 if epi == 5 else N), "f32")
    if epi == 1:
        t = cpu_ref.quick_gelu(t)
    elif epi == 2:
        t = torch.nn.functional.gelu(t)
    elif epi == 4:
        t = t + torch.from_numpy(r0)
    elif epi == 5:
        t = torch.nn.functional.silu(t[:, 0::2]) * t[:, 1::2]
    Ad, Wd, bd = be.f32(A), be.bf16(W), (be.f32(b) if bias else None)
    _call(be, "vck_gemm_f32", Ad, Wd, bd, out, M, N, K, K, K, N // 2 if epi == 5 else N, epi)
    e = rel_err(be.host_f32(out), t.numpy())
    assert e < 1e-5, f"gemm_f32 M{M} N{N} K{K} epi{epi}: rel err {e}"   # fp32 FMA chain over K (<= 3.5e-7*sqrt-ish growth)


def check_attention_f32(be, B, H, T, hd, causal, decode_pos=None, seed=0):
    rng = np.random.RandomState(seed)
    scale = 1.0 / math.sqrt(hd)
    if decode_pos is None:
        q, k, v = (rng.randn(B, H, T, hd).astype(np.float32) for _ in range(3))
        out = be.zeros((B * T, H * hd), "f32")
        qd, kd, vd = be.f32(q), be.f32(k), be.f32(v)
        _call(be, "vck_attention_f32", qd, kd, vd, out, B, H, T, hd, T, T, int(causal), T, None, scale)
        ref = cpu_ref.softmax_attention(torch.from_numpy(q), torch.from_numpy(k), torch.from_numpy(v), scale, causal,
                                        cpu_ref.Rounder(False)).transpose(1, 2).reshape(B * T, H * hd).numpy()
    else:
        S = decode_pos + 9
        q = rng.randn(B, H, 1, hd).astype(np.float32)
        k, v = (rng.randn(B, H, S, hd).astype(np.float32) for _ in range(2))
        out = be.zeros((B, H * hd), "f32")
        qd, kd, vd, pd = be.f32(q), be.f32(k), be.f32(v), be.i32([decode_pos])
        _call(be, "vck_attention_f32", qd, kd, vd, out, B, H, 1, hd, 1, S, 1, 0, pd, scale)
        n = decode_pos + 1
        ref = cpu_ref.softmax_attention(torch.from_numpy(q), torch.from_numpy(k[:, :, :n]), torch.from_numpy(v[:, :, :n]),
                                        scale, False, cpu_ref.Rounder(False)).transpose(1, 2).reshape(B, H * hd).numpy()
    err = np.abs(be.host_f32(out) - ref).max()
    assert err < 2e-6 * max(1.0, np.abs(ref).max()) + 2e-6, f"attention_f32 abs err {err}"


def check_qkv_rope_f32(be, B, T, H, hd, pos0):
    rng = np.random.RandomState(3)
    D = H * hd
    S = pos0 + T + 3
    qkv = rng.randn(B * T, 3 * D).astype(np.float32)
    q, k, v = be.zeros((B, H, T, hd), "f32"), be.zeros((B, H, S, hd), "f32"), be.zeros((B, H, S, hd), "f32")
    cos, sin = rope_tables(S, hd)
    qd, cd, sd, pd = be.f32(qkv), be.f32(cos), be.f32(sin), be.i32([pos0])
    _call(be, "vck_qkv_rope_f32", qd, q, k, v, B, T, H, hd, T, S, pd, cd, sd)
    t = torch.from_numpy(qkv).reshape(B, T, 3, H, hd).permute(2, 0, 3, 1, 4)
    c, s_ = cpu_ref.rope_cos_sin(torch.arange(pos0, pos0 + T), hd, 10000.0)
    rq, rk = cpu_ref.apply_rope(t[0], c, s_).numpy(), cpu_ref.apply_rope(t[1], c, s_).numpy()
    assert np.abs(be.host_f32(q) - rq).max() < 1e-5
    gk, gv = be.host_f32(k), be.host_f32(v)
    assert np.abs(gk[:, :, pos0:pos0 + T] - rk).max() < 1e-5 and np.array_equal(gv[:, :, pos0:pos0 + T], t[2].numpy())


# ---- precision mode "split": every MFMA operand as bf16 hi + lo ---------------------------------------------------------
def split_hi_lo(x):
    """x (fp32) -> (hi, lo) fp32 arrays holding bf16 values: hi = bf16(x), lo = bf16(x - hi); x - (hi + lo) ~ 2^-17 |x|"""
    x = np.asarray(x, dtype=np.float32)
    hi = bf16_round(x)
    lo = bf16_round(x - hi)
    return hi, lo


def _split_value(be, out, N):
    """[M, >= 2N] device bf16 buffer holding [hi | lo] planes -> fp32 hi + lo"""
    o = be.host_f32(out)
    return o[:, :N].astype(np.float64) + o[:, N:2 * N].astype(np.float64)


def fp16_valued(x):
    """values an fp16 checkpoint holds (11 significant bits: bf16 hi + bf16 lo exactly), as fp32"""
    return np.asarray(x, dtype=np.float32).astype(np.float16).astype(np.float32)


def check_gemm_split(be, M, N, K, epi, bias=True, seed=0, ws_mb=0, pad=64, wlo=False):
    """vck_gemm_split: A = [hi | lo] of an fp32 matrix (row stride 2K + pad), W bf16; against the float64 product of the
    fp32 activations.  fp32 outputs: the split's own error (2^-17 relative per operand) — tolerance 3e-5 of the largest
    output; bf16-valued outputs come back as [hi | lo] and are compared the same way.
    wlo: W is fp16-valued (an inexact checkpoint) and goes in as bf16 hi + lo planes (vck_gemm_split_wlo: a third K segment)."""
    rng = np.random.RandomState(seed)
    A = rng.randn(M, K).astype(np.float32)
    W = fp16_valued(rng.randn(N, K) * 0.05) if wlo else bf16_round(rng.randn(N, K) * 0.05)
    b = rng.randn(N).astype(np.float32) * 0.1 if bias else None
    hi, lo = split_hi_lo(A)
    lda = 2 * K + pad
    Acat = np.zeros((M, lda), np.float32)
    Acat[:, :K], Acat[:, K:2 * K] = hi, lo
    t = torch.from_numpy(A.astype(np.float64) @ W.T.astype(np.float64) + (b if bias else 0.0))
    No = N // 2 if epi == 5 else N
    if epi == 1:
        t = t * torch.sigmoid(1.702 * t)
    elif epi == 2:
        t = torch.nn.functional.gelu(t)
    elif epi == 5:
        t = torch.nn.functional.silu(t[:, 0::2]) * t[:, 1::2]
    if epi in (0, 1, 2, 5):
        ldo = 2 * No + pad
        out = be.zeros((M, ldo), "bf16")
        split_out = No
    elif epi == 3:
        ldo, split_out, out = N, 0, be.zeros((M, N), "f32")
    else:
        r0 = rng.randn(M, N).astype(np.float32)
        ldo, split_out, out = N, 0, be.f32(r0.copy())
        t = t + torch.from_numpy(r0).double()
    Wh = bf16_round(W)
    Ad, Wd, bd = be.bf16(Acat), be.bf16(Wh), (be.f32(b) if bias else None)
    ws = be.zeros((max(ws_mb, 1) << 18,), "f32")
    if wlo:
        Wl = W - Wh
        assert np.array_equal(bf16_round(Wl), Wl) and np.abs(Wl).max() > 0, "an fp16 value is bf16 hi + bf16 lo exactly"
        Wld = be.bf16(Wl)
        be.lib.vck_gemm_split_wlo(be.ptr(Ad), be.ptr(Wd), be.ptr(Wld), be.ptr(bd), be.ptr(out), M, N, K, lda, ldo, epi, split_out,
                                  be.ptr(ws) if ws_mb else None, ctypes.c_size_t((ws_mb << 20) if ws_mb else 0), None)
    else:
        be.lib.vck_gemm_split(be.ptr(Ad), be.ptr(Wd), be.ptr(bd), be.ptr(out), M, N, K, lda, ldo, epi, split_out,
                              be.ptr(ws) if ws_mb else None, ctypes.c_size_t((ws_mb << 20) if ws_mb else 0), None)
    be.sync()
    got = _split_value(be, out, No) if split_out else be.host_f32(out).astype(np.float64)
    e = rel_err(got, t.numpy())
    assert e < 3e-5, f"gemm_split M{M} N{N} K{K} epi{epi} wlo{wlo}: rel err {e}"
    return e


def check_gemm_split_wlo(be, M, N, K, epi, seed=0, ws_mb=0):
    return check_gemm_split(be, M, N, K, epi, bias=epi != 5, seed=seed, wlo=True, ws_mb=ws_mb)


def check_weight_planes(be, n=5000, seed=0):
    """vck_f32_to_bf16_planes: hi = bf16(x), lo = bf16(x - hi), the inexact flag; fp16 values come back exactly as hi + lo"""
    rng = np.random.RandomState(seed)
    for vals, inexact in ((fp16_valued(rng.randn(n) * 0.05), 1), (bf16_round(rng.randn(n) * 0.05), 0), (rng.randn(n).astype(np.float32), 1)):
        x, hi, lo, flag = be.f32(vals), be.zeros((n,), "bf16"), be.zeros((n,), "bf16"), be.zeros((1,), "i32")
        be.lib.vck_f32_to_bf16_planes(be.ptr(x), be.ptr(hi), be.ptr(lo), ctypes.c_uint64(n), be.ptr(flag), None)
        be.sync()
        h, l = be.host_f32(hi), be.host_f32(lo)
        assert np.array_equal(h, bf16_round(vals)) and np.array_equal(l, bf16_round(vals - h))
        assert int(be.host_i32(flag)[0]) == inexact
        if inexact and np.array_equal(fp16_valued(vals), vals):
            assert np.array_equal(h + l, vals), "fp16 values are bf16 hi + bf16 lo exactly"
        else:
            assert np.abs(h.astype(np.float64) + l - vals).max() <= 2.0 ** -16 * np.abs(vals).max()


def check_gemm_f32_wlo(be, M, N, K, epi, seed=0):
    """the strict GEMM over W + W_lo against float64 on the fp16-valued weights"""
    rng = np.random.RandomState(seed)
    A = rng.randn(M, K).astype(np.float32)
    W = fp16_valued(rng.randn(N, K) * 0.05)
    Wh = bf16_round(W)
    out = be.zeros((M, N), "f32")
    Ad, Whd, Wld = be.f32(A), be.bf16(Wh), be.bf16(W - Wh)   # (held until the launch has finished)
    be.lib.vck_gemm_f32_wlo(be.ptr(Ad), be.ptr(Whd), be.ptr(Wld), None, be.ptr(out), M, N, K, K, K, N, epi, None)
    be.sync()
    e = rel_err(be.host_f32(out).astype(np.float64), A.astype(np.float64) @ W.T.astype(np.float64))
    assert e < 2e-5, f"gemm_f32_wlo M{M} N{N} K{K}: rel err {e}"
    return e


def check_gemv_split_wlo(be, M, N, K, epi, G, ksplit=0, seed=0):
    """the split decode GEMV (workgroup-shared form) over the packed hi + lo planes of an fp16-valued weight: against float64 on
    the original values, and clearly better than the hi plane alone"""
    be.lib.vck_set_gemv_variant(1)
    try:
        rng = np.random.RandomState(seed)
        c = _wg_case(be, rng, M, N, K, epi, True, G, w16=True)
        val, _, _ = _wg_run(be, c, M, G, ksplit, wlo=True)
        e = rel_err(val, c["ref"])
        assert e < 3e-5, f"gemv_split_wlo M{M} N{N} K{K} epi{epi} G{G} ks{ksplit}: rel err {e}"
        val0, _, _ = _wg_run(be, c, M, G, ksplit)
        e0 = rel_err(val0, c["ref"])
        assert e0 > 4 * e, f"the hi plane alone should miss the fp16 values: {e0} vs {e}"
        return e
    finally:
        be.lib.vck_set_gemv_variant(-1)


def check_gemv_split(be, M, N, K, epi, norm=True, seed=0, fp8=False):
    """vck_gemv_split at M rows (G = 8 for M <= 8, else 16): stacked hi / lo rows of X against the float64 product of the
    fp32 rows; RMSNorm folding (ssq_in -> rstd, RESID epilogue -> ssq_out + stacked xg_out) as in the bf16 kernel."""
    rng = np.random.RandomState(seed)
    G = 8 if M <= 8 else 16
    npart = (max(K, N) // 16 + 15) // 16 * 16
    Xf = rng.randn(M, K).astype(np.float32)
    hi, lo = split_hi_lo(Xf)
    X = np.zeros((2 * G, K), np.float32)
    X[:M], X[G:G + M] = hi, lo
    X[M:G] = 7.0   # rows of the group beyond M must not be read into any result
    W = bf16_round(rng.randn(N, K) * 0.05)
    wsc = None
    if fp8:
        Wd, Wp, wsc = be.bf16(W), be.zeros((N * K,), "u8"), be.zeros((N,), "f32")
        _call(be, "vck_quantize_fp8", Wd, Wp, wsc, N, K)
        W = be.host_f32(Wd)
    else:
        Wd, Wp = be.bf16(W), be.zeros((N * K,), "bf16")
        _call(be, "vck_pack_weight", Wd, Wp, N, K)
    ssq = np.zeros((16, npart), np.float32)
    ssq[:, : K // 16] = rng.rand(16, K // 16).astype(np.float32) + 0.5
    ref = Xf.astype(np.float64) @ W.T.astype(np.float64)
    if norm:
        ref = ref / np.sqrt(ssq[:M, : K // 16].astype(np.float64).sum(-1, keepdims=True) / K + 1e-5)
    No = N // 2 if epi == 3 else N
    gw = (rng.rand(N).astype(np.float32) + 0.5) if epi == 2 else None
    if epi == 3:
        t = torch.from_numpy(ref)
        ref = (torch.nn.functional.silu(t[:, 0::2]) * t[:, 1::2]).numpy()
    r0 = rng.randn(16, No).astype(np.float32)
    if epi == 2:
        ref = ref + r0[:M]
        out = be.f32(r0.copy())
    elif epi == 1:
        out = be.zeros((16, No), "f32")
    else:
        out = be.zeros((2 * G, No), "bf16")
    xg_out = be.zeros((2 * G, N), "bf16") if epi == 2 else None
    ssq_out = be.zeros((16, npart), "f32") if epi == 2 else None
    Xd, ssqd, gwd = be.bf16(X), (be.f32(ssq) if norm else None), (be.f32(gw) if gw is not None else None)
    be.lib.vck_gemv_split(be.ptr(Xd), be.ptr(Wp), be.ptr(wsc), be.ptr(out), be.ptr(ssqd), be.ptr(ssq_out), be.ptr(gwd),
                          be.ptr(xg_out), ctypes.c_int(npart), ctypes.c_float(1e-5), M, N, K, No, epi, G, None)
    be.sync()
    o = be.host_f32(out).astype(np.float64)
    got = (o[:M] + o[G:G + M]) if epi in (0, 3) else o[:M]
    e = rel_err(got, ref)
    assert e < 3e-5, f"gemv_split M{M} N{N} K{K} epi{epi} fp8{fp8}: rel err {e}"
    if epi == 2:
        xg = be.host_f32(xg_out).astype(np.float64)
        e2 = rel_err(xg[:M] + xg[G:G + M], got * gw)
        assert e2 < 2e-5, f"gemv_split xg_out: {e2}"
        so = be.host_f32(ssq_out)[:M, : N // 16].astype(np.float64).sum(-1)
        assert np.abs(so / (got ** 2).sum(-1) - 1).max() < 1e-5
        assert np.array_equal(o[M:], r0[M:].astype(np.float64)), "residual rows beyond M were touched"
    return e


def check_gemv_split_groups_agree(be, N, K, epi, seed=0):
    """a row gets the same bits from the G = 8 form (hi / lo share one MFMA row group) as from the G = 16 form (two row
    groups): what lets a request decode in the pool (G = 16) or on its own loop (G = 8 for <= 8 rows) with identical ids"""
    rng = np.random.RandomState(seed)
    M = 5
    Xf = rng.randn(M, K).astype(np.float32)
    hi, lo = split_hi_lo(Xf)
    W = bf16_round(rng.randn(N, K) * 0.05)
    Wd, Wp = be.bf16(W), be.zeros((N * K,), "bf16")
    _call(be, "vck_pack_weight", Wd, Wp, N, K)
    No = N // 2 if epi == 3 else N
    outs = []
    for G in (8, 16):
        X = np.zeros((2 * G, K), np.float32)
        X[:M], X[G:G + M] = hi, lo
        out = be.zeros((2 * G, No), "bf16") if epi in (0, 3) else be.zeros((16, No), "f32")
        Xd = be.bf16(X)
        be.lib.vck_gemv_split(be.ptr(Xd), be.ptr(Wp), None, be.ptr(out), None, None, None, None, ctypes.c_int(16),
                              ctypes.c_float(1e-5), M, N, K, No, epi, G, None)
        be.sync()
        o = be.host_f32(out)
        outs.append((o[:M], o[G:G + M]) if epi in (0, 3) else (o[:M],))
    for a, b in zip(*outs):
        assert np.array_equal(a, b), "the G = 8 and G = 16 forms of the split GEMV disagree"


def check_norm_split(be, rows, D, rms=True, seed=0):
    rng = np.random.RandomState(seed)
    x = (rng.randn(rows, D) * 2).astype(np.float32)
    w = (rng.rand(D) + 0.5).astype(np.float32)
    b = (rng.randn(D) * 0.1).astype(np.float32)
    x64 = x.astype(np.float64)
    if rms:
        ref = x64 / np.sqrt((x64 ** 2).mean(-1, keepdims=True) + 1e-5) * w
    else:
        mu = x64.mean(-1, keepdims=True)
        ref = (x64 - mu) / np.sqrt(((x64 - mu) ** 2).mean(-1, keepdims=True) + 1e-5) * w + b
    ld = 2 * D + 64
    y = be.zeros((rows, ld), "bf16")
    xd, wd, bd = be.f32(x), be.f32(w), be.f32(b)
    if rms:
        be.lib.vck_rmsnorm_split(be.ptr(xd), None, be.ptr(wd), be.ptr(y), rows, D, ctypes.c_float(1e-5), ld, ctypes.c_uint64(D), None)
    else:
        be.lib.vck_layernorm_split(be.ptr(xd), be.ptr(wd), be.ptr(bd), be.ptr(y), rows, D, ctypes.c_float(1e-5), ld, ctypes.c_uint64(D), None)
    be.sync()
    e = rel_err(_split_value(be, y, D), ref)
    assert e < 2e-5, f"norm_split rows{rows} D{D} rms{rms}: {e}"


def check_qkv_split32_and_attention_split(be, B, H, T, hd, causal, rope=True, seed=0, spike=False):
    """vck_qkv_split32 + vck_attention_split as the split prefill chains them: fp32 fused-QKV rows -> RoPE -> hi / lo planes
    (+ fp32 cache rows) -> flash attention with 3 MFMAs per product, against the fp32 oracle (no rounding points)."""
    rng = np.random.RandomState(seed)
    D = H * hd
    Ts = (T + 63) // 64 * 64
    S_cap = Ts + 64
    qkv = rng.randn(B * T, 3 * D).astype(np.float32)
    if spike:
        qkv.reshape(B, T, 3, H, hd)[:, T - 3, 1] = qkv.reshape(B, T, 3, H, hd)[:, T - 1, 0] * 4
    cos, sin = rope_tables(S_cap, hd)
    planes = lambda *shape: [be.zeros(shape, "bf16") for _ in range(2)]
    (qh, ql), (kh, kl), (vh, vl) = planes(B, H, Ts, hd), planes(B, H, Ts, hd), planes(B, H, hd, Ts)
    k32, v32 = be.zeros((B, H, S_cap, hd), "f32"), be.zeros((B, H, S_cap, hd), "f32")
    qd, cd, sd = be.f32(qkv), be.f32(cos), be.f32(sin)
    be.lib.vck_qkv_split32(be.ptr(qd), be.ptr(qh), be.ptr(ql), be.ptr(kh), be.ptr(kl), be.ptr(vh), be.ptr(vl), be.ptr(k32),
                           be.ptr(v32), B, T, H, hd, Ts, Ts, Ts, S_cap, be.ptr(cd) if rope else None, be.ptr(sd) if rope else None,
                           None)
    be.sync()
    x = torch.from_numpy(qkv).view(B, T, 3, H, hd).permute(2, 0, 3, 1, 4).contiguous()   # [3,B,H,T,hd]
    q, k, v = x[0], x[1], x[2]
    if rope:
        c, s_ = torch.from_numpy(cos[:T]), torch.from_numpy(sin[:T])
        def rot(t):
            a, b_ = t[..., : hd // 2], t[..., hd // 2:]
            return torch.cat([a * c - b_ * s_, b_ * c + a * s_], -1)
        q, k = rot(q), rot(k)
    gk32, gv32 = be.host_f32(k32), be.host_f32(v32)
    assert np.abs(gk32[:, :, :T] - k.numpy()).max() < 1e-5 and np.array_equal(gv32[:, :, :T], v.numpy())
    assert not gk32[:, :, T:].any() and not gv32[:, :, T:].any()
    for (h_, l_), ref_ in (((qh, ql), q), ((kh, kl), k)):
        g_ = be.host_f32(h_)[:, :, :T].astype(np.float64) + be.host_f32(l_)[:, :, :T].astype(np.float64)
        assert np.abs(g_ - ref_.numpy()).max() < 2e-5 * max(1.0, float(ref_.abs().max()))
    gvt = be.host_f32(vh).astype(np.float64) + be.host_f32(vl).astype(np.float64)     # the scratch's key order (vt_key_pos)
    rvt = np.zeros_like(gvt)
    rvt[..., :T] = v.numpy().transpose(0, 1, 3, 2)
    assert np.abs(gvt - rvt[..., vt_key_pos(Ts)]).max() < 2e-5 * float(v.abs().max())
    scale = 1.0 / math.sqrt(hd)
    ldo = 2 * D + 64
    out = be.zeros((B * T, ldo), "bf16")
    be.lib.vck_attention_split(be.ptr(qh), be.ptr(ql), be.ptr(kh), be.ptr(kl), be.ptr(vh), be.ptr(vl), be.ptr(out), B, H, T, hd,
                               Ts, Ts, int(causal), ctypes.c_float(scale), ldo, D, None)
    be.sync()
    ref = cpu_ref.softmax_attention(q, k, v, scale, causal, cpu_ref.Rounder(False))
    ref = ref.transpose(1, 2).reshape(B * T, D).numpy()
    err = np.abs(_split_value(be, out, D) - ref).max()
    assert err < 3e-5 * max(1.0, np.abs(ref).max()), f"attention_split B{B} H{H} T{T} hd{hd} causal{causal}: abs err {err}"
    return err


def check_attention_decode_kv32(be, B, H, hd, pos, seed=0):
    """fused decode attention over fp32 qkv / fp32 caches (split mode), a position per row, one row inactive; output as
    stacked hi / lo row groups"""
    rng = np.random.RandomState(seed)
    D = H * hd
    G = 8 if B <= 8 else 16
    poss = [max(1, pos - 13 * b) for b in range(B)]
    S = (pos + 1 + 63) // 64 * 64 + 64
    qkv = rng.randn(B, 3 * D).astype(np.float32)
    k_old, v_old = rng.randn(B, H, S, hd).astype(np.float32), rng.randn(B, H, S, hd).astype(np.float32)
    kd, vd = be.f32(k_old), be.f32(v_old)
    nrows_out = ((B + G - 1) // G) * 2 * G
    out = be.zeros((nrows_out, D), "bf16")
    cos, sin = rope_tables(S, hd)
    scale = 1.0 / math.sqrt(hd)
    qd, cd, sd = be.f32(qkv), be.f32(cos), be.f32(sin)
    inactive = B - 1 if B > 1 else -1
    rows = np.zeros((B, 4), np.int32)
    rows[:, 0] = 1
    rows[:, 1] = poss
    if inactive >= 0:
        rows[inactive, 0] = 0
    rd = be.i32(rows)
    base = rd.ctypes.data if isinstance(rd, np.ndarray) else rd.data_ptr()
    be.lib.vck_attention_decode_kv32(be.ptr(qd), be.ptr(kd), be.ptr(vd), be.ptr(out), B, H, hd, S, c_p(base + 4), 4, c_p(base),
                                     be.ptr(cd), be.ptr(sd), ctypes.c_float(scale), G, None)
    be.sync()
    gk, gv, go = be.host_f32(kd), be.host_f32(vd), be.host_f32(out).astype(np.float64)
    for b in range(B):
        pb = poss[b]
        orow = (b // G) * 2 * G + b % G
        if b == inactive:
            assert np.array_equal(gk[b], k_old[b]) and np.array_equal(gv[b], v_old[b]) and not go[orow].any()
            continue
        x = torch.from_numpy(qkv[b]).view(3, H, 1, hd)
        c, s_ = torch.from_numpy(cos[pb]), torch.from_numpy(sin[pb])
        def rot(t):
            a, b_ = t[..., : hd // 2], t[..., hd // 2:]
            return torch.cat([a * c - b_ * s_, b_ * c + a * s_], -1)
        q, kn, vn = rot(x[0]), rot(x[1]), x[2]
        assert np.abs(gk[b, :, pb] - kn[:, 0].numpy()).max() < 1e-5 and np.array_equal(gv[b, :, pb], vn[:, 0].numpy())
        keep = np.ones(S, bool)
        keep[pb] = False
        assert np.array_equal(gk[b][:, keep], k_old[b][:, keep]) and np.array_equal(gv[b][:, keep], v_old[b][:, keep])
        k_all = torch.cat([torch.from_numpy(k_old[b, :, :pb]), kn], 1)[None]
        v_all = torch.cat([torch.from_numpy(v_old[b, :, :pb]), vn], 1)[None]
        ref = cpu_ref.softmax_attention(q[None], k_all, v_all, scale, False, cpu_ref.Rounder(False))
        ref = ref.transpose(1, 2).reshape(D).numpy()
        err = np.abs(go[orow] + go[orow + G] - ref).max()
        assert err < 3e-5 * max(1.0, np.abs(ref).max()), f"decode attention kv32 row {b}: {err}"


# ---- the workgroup-shared-activation decode GEMV of precision mode "split" (gemv_wg_kernel) -------------------------------------
def _gemv_full(be, X, Wp, out, ssq_in, ssq_out, xg_w, xg_out, npart, M, N, K, ldo, epi, G=0, ksplit=0, sk=None):
    """vck_gemv_full through raw pointers (bf16 weights); sk = (scratch f32, counters i32) with their true capacities"""
    be.lib.vck_gemv_full(be.ptr(X), be.ptr(Wp), None, be.ptr(out), be.ptr(ssq_in), be.ptr(ssq_out), be.ptr(xg_w), be.ptr(xg_out),
                         ctypes.c_int(npart), ctypes.c_float(1e-5), be.ptr(sk[0]) if sk else None,
                         ctypes.c_ulonglong(sk[2] if sk else 0), be.ptr(sk[1]) if sk else None, ctypes.c_int(sk[3] if sk else 0),
                         ctypes.c_int(ksplit), M, N, K, ldo, epi, G, None)
    be.sync()


def _wg_case(be, rng, rows, N, K, epi, norm, G, w16=False):
    """inputs of a wg-GEMV case over `rows` activation rows; G > 0: the stacked hi / lo form of precision mode split; w16: an
    fp16-valued weight (packed hi and lo planes)"""
    npart = (max(K, N) // 16 + 15) // 16 * 16
    Xf = rng.randn(rows, K).astype(np.float32)
    if G:
        hi, lo = split_hi_lo(Xf)
    else:
        Xf = bf16_round(Xf)
        hi, lo = Xf, None
    W = fp16_valued(rng.randn(N, K) * 0.05) if w16 else bf16_round(rng.randn(N, K) * 0.05)
    Wd, Wp = be.bf16(bf16_round(W)), be.zeros((N * K,), "bf16")
    _call(be, "vck_pack_weight", Wd, Wp, N, K)
    Wp_lo = None
    if w16:
        Wp_lo, Wl_d = be.zeros((N * K,), "bf16"), be.bf16(W - bf16_round(W))
        _call(be, "vck_pack_weight", Wl_d, Wp_lo, N, K)
        be.sync()
    ssq = np.zeros((32, npart), np.float32)
    ssq[:, : K // 16] = rng.rand(32, K // 16).astype(np.float32) + 0.5
    ref = Xf.astype(np.float64) @ W.T.astype(np.float64)
    if norm:
        ref = ref / np.sqrt(ssq[:rows, : K // 16].astype(np.float64).sum(-1, keepdims=True) / K + 1e-5)
    No = N // 2 if epi == 3 else N
    if epi == 3:
        t = torch.from_numpy(ref)
        ref = (torch.nn.functional.silu(t[:, 0::2]) * t[:, 1::2]).numpy()
    gw = (rng.rand(N).astype(np.float32) + 0.5) if epi == 2 else None
    r0 = rng.randn(32, No).astype(np.float32)
    if epi == 2:
        ref = ref + r0[:rows]
    sk = (be.zeros((8 * (N // 16) * 2 * 256,), "f32"), be.zeros((N // 16 * 2,), "i32"), 8 * (N // 16) * 2 * 256, N // 16 * 2)
    return dict(npart=npart, hi=hi, lo=lo, Wp=Wp, Wp_lo=Wp_lo, Wd=Wd, ssq=ssq, ref=ref, No=No, gw=gw, r0=r0, sk=sk, N=N, K=K, epi=epi, norm=norm)


def _wg_run(be, c, M, G, ksplit=0, row0=0, wlo=False):
    """one launch over rows [row0, row0 + M) of the case -> (values [M, No] float64 (hi + lo summed for bf16-valued outputs in
    split form), raw outputs for bit comparisons)"""
    N, K, epi, No = c["N"], c["K"], c["epi"], c["No"]
    if G:
        X = np.full((2 * G, K), 7.0, np.float32)     # rows of the group beyond M must not reach any result
        X[:M], X[G:G + M] = c["hi"][row0:row0 + M], c["lo"][row0:row0 + M]
    else:
        X = np.full((32, K), 7.0, np.float32)
        X[:M] = c["hi"][row0:row0 + M]
    rows_out = 2 * G if (G and epi in (0, 3)) else 32
    if epi == 2:
        out = be.f32(c["r0"][row0:].copy() if row0 else c["r0"].copy())
        if row0:
            pad = np.zeros((32, No), np.float32)
            pad[: 32 - row0] = c["r0"][row0:]
            out = be.f32(pad)
    elif epi == 1:
        out = be.zeros((32, No), "f32")
    else:
        out = be.zeros((rows_out, No), "bf16")
    xg_out = be.zeros((2 * G if G else 32, N), "bf16") if epi == 2 else None
    ssq_out = be.zeros((32, c["npart"]), "f32") if epi == 2 else None
    ssq = np.zeros_like(c["ssq"])
    ssq[:M] = c["ssq"][row0:row0 + M]
    Xd, ssqd, gwd = be.bf16(X), (be.f32(ssq) if c["norm"] else None), (be.f32(c["gw"]) if c["gw"] is not None else None)
    if wlo:
        sk = c["sk"]
        be.lib.vck_gemv_split_wlo(be.ptr(Xd), be.ptr(c["Wp"]), be.ptr(c["Wp_lo"]), be.ptr(out), be.ptr(ssqd), be.ptr(ssq_out), be.ptr(gwd),
                                  be.ptr(xg_out), ctypes.c_int(c["npart"]), ctypes.c_float(1e-5), be.ptr(sk[0]), ctypes.c_ulonglong(sk[2]),
                                  be.ptr(sk[1]), ctypes.c_int(sk[3]), ctypes.c_int(ksplit), M, N, K, No, epi, G, None)
        be.sync()
    else:
        _gemv_full(be, Xd, c["Wp"], out, ssqd, ssq_out, gwd, xg_out, c["npart"], M, N, K, No, epi, G=G, ksplit=ksplit, sk=c["sk"])
    o = be.host_f32(out)
    assert not be.host_i32(c["sk"][1]).any(), "arrival counters must be re-armed"
    if G and epi in (0, 3):
        val, raw = o[:M].astype(np.float64) + o[G:G + M].astype(np.float64), (o[:M].copy(), o[G:G + M].copy())
    else:
        val, raw = o[:M].astype(np.float64), (o[:M].copy(),)
    extra = None
    if epi == 2:
        xg = be.host_f32(xg_out)
        extra = dict(xg=(xg[:M].copy(), xg[G:G + M].copy()) if G else (xg[:M].copy(),), ssq=be.host_f32(ssq_out)[:M, : N // 16].copy(),
                     untouched=o[M:].copy())
    return val, raw, extra


def check_gemv_wg(be, M, N, K, epi, norm=True, G=8, ksplit=0, seed=0):
    """gemv_wg_kernel (the GEMV of precision mode "split": hi / lo rows, one weight pass) against the float64 product: every
    epilogue, RMSNorm folding on both sides, explicit K-slices with the cross-workgroup hand-off"""
    assert G > 0, "the workgroup-shared form serves the split step only (its bf16 form was removed in round 5)"
    be.lib.vck_set_gemv_variant(1)
    be.lib.vck_gemv_wg_launches.restype = ctypes.c_ulonglong
    n0 = be.lib.vck_gemv_wg_launches()
    try:
        rng = np.random.RandomState(seed)
        c = _wg_case(be, rng, M, N, K, epi, norm, G)
        val, raw, extra = _wg_run(be, c, M, G, ksplit)
        assert be.lib.vck_gemv_wg_launches() == n0 + 1, "the call was not served by gemv_wg_kernel"
        e = rel_err(val, c["ref"])
        tol = 3e-5 if G else (2 ** -8 if epi in (0, 3) else 2e-5)
        assert e < tol, f"gemv_wg M{M} N{N} K{K} epi{epi} G{G} ks{ksplit}: rel err {e}"
        if epi == 2:
            xg = sum(x.astype(np.float64) for x in extra["xg"])
            assert rel_err(xg, val * c["gw"]) < (2e-5 if G else 2 ** -8)
            assert np.abs(extra["ssq"].astype(np.float64).sum(-1) / (val ** 2).sum(-1) - 1).max() < 1e-5
            assert np.array_equal(extra["untouched"], c["r0"][M:]), "residual rows beyond M were touched"
        return e
    finally:
        be.lib.vck_set_gemv_variant(-1)


def check_gemv_wg_rows_agree(be, N, K, epi, norm=True, G=True, ksplit=0, seed=0):
    """the pool's promise for the wg form: a row gets the same BITS from a 29-row pass as from the 8-row (5-row) pass and the
    13-row pass that hold it — the k order of a sum is a function of the matrix alone; in split form G = 32 vs G = 8 / 16"""
    be.lib.vck_set_gemv_variant(1)
    try:
        rng = np.random.RandomState(seed)
        c = _wg_case(be, rng, 29, N, K, epi, norm, 32 if G else 0)
        _, big, _ = _wg_run(be, c, 29, 32 if G else 0, ksplit)
        for row0, M, Gs in ((0, 5, 8), (0, 8, 8), (16, 13, 16), (8, 16, 16)):
            _, small, _ = _wg_run(be, c, M, Gs if G else 0, ksplit, row0=row0)
            if epi == 2 and row0:
                continue   # (the in-place residual form adds into the rows the buffer holds: only row0 == 0 lines up)
            for a, b in zip(big, small):
                assert np.array_equal(a[row0:row0 + M], b), f"rows {row0}..{row0 + M - 1} differ between the 29-row and the {M}-row pass"
    finally:
        be.lib.vck_set_gemv_variant(-1)


def check_gemv_wg_six_waves(be, N, K, epi, rows, norm, ksplit=0, seed=0):
    """gemv_wg_kernel with SIX waves (= tiles) per workgroup (round 6: the geometry of the matrices whose four-wave launch leaves CUs
    with twice the tiles of others; vck_set_gemv_variant(3) forces it on any matrix, 2 forces four): the K-slices and the order
    of every sum are the four-wave geometry's, so without the folded RMSNorm the BITS are equal; with it only the rstd partials
    are added per wave (6 shares instead of 4): rstd moves by an fp32 ulp, a hi + lo output by up to 2^-17 — within tolerance of
    float64 either way"""
    try:
        rng = np.random.RandomState(seed)
        c = _wg_case(be, rng, 32, N, K, epi, norm, 32)
        for M in rows:
            G = 8 if M <= 8 else 16 if M <= 16 else 24 if M <= 24 else 32
            be.lib.vck_set_gemv_variant(2)
            v4, raw4, ex4 = _wg_run(be, c, M, G, ksplit)
            be.lib.vck_set_gemv_variant(3)
            v6, raw6, ex6 = _wg_run(be, c, M, G, ksplit)
            e = rel_err(v6, c["ref"][:M])
            assert e < 3e-5, f"gemv_wg six waves M{M} N{N} K{K} epi{epi} ks{ksplit}: rel err {e}"
            if not norm:
                for a, b in zip(raw4, raw6):
                    assert np.array_equal(a, b), f"M{M} N{N} K{K} epi{epi} ks{ksplit}: six waves per workgroup changed the bits"
            else:
                assert rel_err(v6, v4) < 3e-5   # (the float64 tolerance: bf16-valued outputs are hi + lo pairs, ~2^-17 each)
    finally:
        be.lib.vck_set_gemv_variant(-1)


def check_gemv_wide(be, N, K, epi, rows, norm=True, seed=0):
    """the "wide" geometry of the ring kernel (vck_set_gemv_wide: ceil(tiles / 256) tiles per 4-wave workgroup, deep ring; on by
    default for the classes that measured faster, 2 = every class): bit for bit the pair geometry's result at every row count
    (ragged last workgroup included), and within tolerance of float64"""
    be.lib.vck_gemv_wide_launches.restype = ctypes.c_ulonglong
    try:
        rng = np.random.RandomState(seed)
        c = _wg_case(be, rng, 32, N, K, epi, norm, 0)
        worst = 0.0
        for M in rows:
            be.lib.vck_set_gemv_wide(0)
            _, raw0, ex0 = _wg_run_plain(be, c, M)
            be.lib.vck_set_gemv_wide(2)
            n0 = be.lib.vck_gemv_wide_launches()
            val, raw1, ex1 = _wg_run_plain(be, c, M)
            assert be.lib.vck_gemv_wide_launches() == n0 + 1, f"M{M} N{N} epi{epi}: not served by the wide geometry"
            assert np.array_equal(raw0, raw1), f"M{M} N{N} K{K} epi{epi}: the wide geometry changed the bits"
            e = rel_err(val, c["ref"][:M])
            assert e < (2 ** -8 if epi in (0, 3) else 2e-5), f"gemv wide M{M} N{N} K{K} epi{epi}: rel err {e}"
            worst = max(worst, e)
        return worst
    finally:
        be.lib.vck_set_gemv_wide(-1)


def _wg_run_plain(be, c, M, sk=None):
    """one bf16 launch over the first M rows of a _wg_case (sk: the split-K buffers, or none) -> (float64 values, raw output, extras)"""
    N, K, epi, No = c["N"], c["K"], c["epi"], c["No"]
    X = np.full((32, K), 7.0, np.float32)
    X[:M] = c["hi"][:M]
    out = be.f32(c["r0"].copy()) if epi == 2 else be.zeros((32, No), "f32" if epi == 1 else "bf16")
    xg_out = be.zeros((32, N), "bf16") if epi == 2 else None
    ssq_out = be.zeros((32, c["npart"]), "f32") if epi == 2 else None
    ssq = np.zeros_like(c["ssq"])
    ssq[:M] = c["ssq"][:M]
    Xd, ssqd, gwd = be.bf16(X), (be.f32(ssq) if c["norm"] else None), (be.f32(c["gw"]) if c["gw"] is not None else None)
    _gemv_full(be, Xd, c["Wp"], out, ssqd, ssq_out, gwd, xg_out, c["npart"], M, N, K, No, epi, sk=sk)
    o = be.host_f32(out)
    extra = None
    if epi == 2:
        extra = dict(xg=be.host_f32(xg_out)[:M].copy(), ssq=be.host_f32(ssq_out)[:M, : N // 16].copy(), untouched=o[M:].copy())
    return o[:M].astype(np.float64), o[:M].copy(), extra


# ---- fp24 KV caches of precision mode "split" (rows of hd x u16 | hd x u8: the top 24 bits of fp32, RNE) ------------------------
def f24_round(x):
    """fp32 -> the nearest fp24 value (round to nearest even on bit 8), as float32"""
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7F + ((u >> 8) & 1)) & 0xFFFFFF00
    return u.astype(np.uint32).view(np.float32).reshape(np.shape(x))


def f24_pack(x):
    """[..., S, hd] float32 -> uint8 rows [..., S, 3 * hd] in the cache layout"""
    r = f24_round(x).view(np.uint32)
    hi = (r >> 16).astype(np.uint16)
    lo = ((r >> 8) & 0xFF).astype(np.uint8)
    return np.concatenate([hi.view(np.uint8).reshape(*hi.shape[:-1], -1), lo], axis=-1)


def f24_unpack(rows, hd):
    """uint8 rows [..., 3 * hd] -> float32 [..., hd]"""
    hi = np.ascontiguousarray(rows[..., : 2 * hd]).view(np.uint16).astype(np.uint32)
    lo = rows[..., 2 * hd:].astype(np.uint32)
    return ((hi << 16) | (lo << 8)).view(np.float32)


def _u8_dev(be, a):
    a = np.ascontiguousarray(a, dtype=np.uint8)
    if be.name == "hip":
        return torch.from_numpy(a).to(be.dev)
    z = be.zeros(a.shape, "u8")
    z[...] = a
    return z


def _u8_host(be, a):
    return a.cpu().numpy() if be.name == "hip" else np.array(a)


def check_kv24(be, B, H, hd, pos, T_prefill=70, seed=0):
    """the fp24 cache end to end at kernel level: vck_qkv_split24 writes rows that decode to fp24(RoPE(k)) / fp24(v) exactly
    (RNE), and vck_attention_decode_kv24 — positions per row, one row inactive, append of the new row — equals the fp32 oracle
    on the fp24-rounded cache to 3e-5."""
    rng = np.random.RandomState(seed)
    D = H * hd
    # ---- prefill writer
    T = T_prefill
    Ts = (T + 63) // 64 * 64
    S_cap = Ts + 64
    qkv = rng.randn(B * T, 3 * D).astype(np.float32)
    cos, sin = rope_tables(max(S_cap, pos + 130), hd)
    planes = lambda *shape: [be.zeros(shape, "bf16") for _ in range(2)]
    (qh, ql), (kh, kl), (vh, vl) = planes(B, H, Ts, hd), planes(B, H, Ts, hd), planes(B, H, hd, Ts)
    k24, v24 = be.zeros((B, H, S_cap, 3 * hd), "u8"), be.zeros((B, H, S_cap, 3 * hd), "u8")
    qd, cd, sd = be.f32(qkv), be.f32(cos), be.f32(sin)
    be.lib.vck_qkv_split24(be.ptr(qd), be.ptr(qh), be.ptr(ql), be.ptr(kh), be.ptr(kl), be.ptr(vh), be.ptr(vl), be.ptr(k24),
                           be.ptr(v24), B, T, H, hd, Ts, Ts, Ts, S_cap, be.ptr(cd), be.ptr(sd), None)
    be.sync()
    x = torch.from_numpy(qkv).view(B, T, 3, H, hd).permute(2, 0, 3, 1, 4).contiguous()
    c, s_ = torch.from_numpy(cos[:T]), torch.from_numpy(sin[:T])
    rot = lambda t, c=c, s_=s_: torch.cat([t[..., : hd // 2] * c - t[..., hd // 2:] * s_, t[..., hd // 2:] * c + t[..., : hd // 2] * s_], -1)
    gk, gv = f24_unpack(_u8_host(be, k24), hd), f24_unpack(_u8_host(be, v24), hd)
    kr = rot(x[1]).numpy()
    # the device rotates in fp32 with fma contraction possible: allow one fp24 step around the reference rounding
    assert np.abs(gk[:, :, :T] - kr).max() <= 2.0 ** -16 * np.abs(kr).max() and np.array_equal(gv[:, :, :T], f24_round(x[2].numpy()))
    assert not _u8_host(be, k24)[:, :, T:].any() and not _u8_host(be, v24)[:, :, T:].any()
    # ---- decode attention over fp24 caches
    G = 8 if B <= 8 else (16 if B <= 16 else 32)
    poss = [max(1, pos - 13 * b) for b in range(B)]
    S = (pos + 1 + 63) // 64 * 64 + 64
    q1 = rng.randn(B, 3 * D).astype(np.float32)
    k_old, v_old = f24_round(rng.randn(B, H, S, hd).astype(np.float32)), f24_round(rng.randn(B, H, S, hd).astype(np.float32))
    kd, vd = _u8_dev(be, f24_pack(k_old)), _u8_dev(be, f24_pack(v_old))
    nrows_out = ((B + G - 1) // G) * 2 * G
    out = be.zeros((nrows_out, D), "bf16")
    scale = 1.0 / math.sqrt(hd)
    q1d = be.f32(q1)
    inactive = B - 1 if B > 1 else -1
    rows = np.zeros((B, 4), np.int32)
    rows[:, 0] = 1
    rows[:, 1] = poss
    if inactive >= 0:
        rows[inactive, 0] = 0
    rd = be.i32(rows)
    base = rd.ctypes.data if isinstance(rd, np.ndarray) else rd.data_ptr()
    be.lib.vck_attention_decode_kv24(be.ptr(q1d), be.ptr(kd), be.ptr(vd), be.ptr(out), B, H, hd, S, c_p(base + 4), 4, c_p(base),
                                     be.ptr(cd), be.ptr(sd), ctypes.c_float(scale), G, None)
    be.sync()
    gk, gv, go = f24_unpack(_u8_host(be, kd), hd), f24_unpack(_u8_host(be, vd), hd), be.host_f32(out).astype(np.float64)
    for b in range(B):
        pb = poss[b]
        orow = (b // G) * 2 * G + b % G
        if b == inactive:
            assert np.array_equal(gk[b], k_old[b]) and np.array_equal(gv[b], v_old[b]) and not go[orow].any()
            continue
        xx = torch.from_numpy(q1[b]).view(3, H, 1, hd)
        cb, sb = torch.from_numpy(cos[pb]), torch.from_numpy(sin[pb])
        q, kn, vn = rot(xx[0], cb, sb), rot(xx[1], cb, sb), xx[2]
        assert np.abs(gk[b, :, pb] - kn[:, 0].numpy()).max() <= 2.0 ** -16 * np.abs(kn.numpy()).max()
        assert np.array_equal(gv[b, :, pb], f24_round(vn[:, 0].numpy()))
        keep = np.ones(S, bool)
        keep[pb] = False
        assert np.array_equal(gk[b][:, keep], k_old[b][:, keep]) and np.array_equal(gv[b][:, keep], v_old[b][:, keep])
        k_all = torch.cat([torch.from_numpy(k_old[b, :, :pb]), torch.from_numpy(gk[b, :, pb:pb + 1])], 1)[None]
        v_all = torch.cat([torch.from_numpy(v_old[b, :, :pb]), torch.from_numpy(gv[b, :, pb:pb + 1])], 1)[None]
        ref = cpu_ref.softmax_attention(q[None], k_all, v_all, scale, False, cpu_ref.Rounder(False))
        ref = ref.transpose(1, 2).reshape(D).numpy()
        err = np.abs(go[orow] + go[orow + G] - ref).max()
        assert err < 3e-5 * max(1.0, np.abs(ref).max()), f"decode attention kv24 row {b}: {err}"


# ---- e4m3 KV caches of the fp8 weight format -------------------------------------------------------------------------------------
def e4m3_round(x):
    """nearest OCP e4m3fn value (saturating at +-448), float32 — torch's cast, which the device's software encode equals"""
    return torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).clamp(-448.0, 448.0).to(torch.float8_e4m3fn).float().numpy()


def e4m3_bytes(x):
    return torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).clamp(-448.0, 448.0).to(torch.float8_e4m3fn).view(torch.uint8).numpy()


def e4m3_from_bytes(b):
    return torch.from_numpy(np.ascontiguousarray(b, dtype=np.uint8)).view(torch.float8_e4m3fn).float().numpy()


def check_kv8(be, B, H, hd, pos, T_prefill=70, seed=0):
    """the e4m3 KV cache at kernel level: vck_qkv_split_kv8 writes bytes equal to torch's e4m3 cast of the bf16 K (after RoPE) /
    V rows (and the bf16 K scratch + V^T the flash kernel reads, as before); vck_attention_decode_kv8 — a position per row, one
    row inactive, the new row appended in e4m3 — equals the fp32 oracle on the dequantised cache to one bf16 output rounding."""
    rng = np.random.RandomState(seed)
    D = H * hd
    T = T_prefill
    Ts = (T + 63) // 64 * 64
    S_cap = Ts + 64
    qkv = bf16_round(rng.randn(B * T, 3 * D))
    cos, sin = rope_tables(max(S_cap, pos + 130), hd)
    q, k, vt = be.zeros((B, H, Ts, hd), "bf16"), be.zeros((B, H, Ts, hd), "bf16"), be.zeros((B, H, hd, Ts), "bf16")
    k8, v8 = be.zeros((B, H, S_cap, hd), "u8"), be.zeros((B, H, S_cap, hd), "u8")
    qd, cd, sd = be.bf16(qkv), be.f32(cos), be.f32(sin)
    be.lib.vck_qkv_split_kv8(be.ptr(qd), be.ptr(q), be.ptr(k), be.ptr(k8), be.ptr(v8), be.ptr(vt), B, T, H, hd, Ts, Ts, Ts, S_cap,
                             be.ptr(cd), be.ptr(sd), None)
    be.sync()
    gk = be.host_f32(k)[:, :, :T]                       # the bf16 rows (checked against the oracle by check_qkv_split)
    x = qkv.reshape(B, T, 3, H, hd).transpose(2, 0, 3, 1, 4)
    assert np.array_equal(_u8_host(be, k8)[:, :, :T], e4m3_bytes(gk)), "K cache bytes != e4m3(bf16 K rows)"
    assert np.array_equal(_u8_host(be, v8)[:, :, :T], e4m3_bytes(x[2])), "V cache bytes != e4m3(V rows)"
    assert not _u8_host(be, k8)[:, :, T:].any() and not _u8_host(be, v8)[:, :, T:].any()
    # ---- decode step over e4m3 caches
    poss = [max(1, pos - 13 * b) for b in range(B)]
    S = (pos + 1 + 63) // 64 * 64 + 64
    q1 = bf16_round(rng.randn(B, 3 * D))
    k_old, v_old = e4m3_round(rng.randn(B, H, S, hd) * 1.5), e4m3_round(rng.randn(B, H, S, hd) * 1.5)
    kd, vd = _u8_dev(be, e4m3_bytes(k_old)), _u8_dev(be, e4m3_bytes(v_old))
    out = be.zeros((B, D), "bf16")
    scale = 1.0 / math.sqrt(hd)
    q1d = be.bf16(q1)
    inactive = B - 1 if B > 1 else -1
    rows = np.zeros((B, 4), np.int32)
    rows[:, 0] = 1
    rows[:, 1] = poss
    if inactive >= 0:
        rows[inactive, 0] = 0
    rd = be.i32(rows)
    base = rd.ctypes.data if isinstance(rd, np.ndarray) else rd.data_ptr()
    be.lib.vck_attention_decode_kv8(be.ptr(q1d), be.ptr(kd), be.ptr(vd), be.ptr(out), B, H, hd, S, c_p(base + 4), 4, c_p(base),
                                    be.ptr(cd), be.ptr(sd), ctypes.c_float(scale), None)
    be.sync()
    gk8, gv8, got = e4m3_from_bytes(_u8_host(be, kd)), e4m3_from_bytes(_u8_host(be, vd)), be.host_f32(out)
    for b in range(B):
        pb = poss[b]
        if b == inactive:
            assert np.array_equal(gk8[b], k_old[b]) and np.array_equal(gv8[b], v_old[b]) and not got[b].any()
            continue
        rq, rk, rv = _split_ref(q1[b:b + 1], 1, 1, H, hd, True, pos0=pb)   # roped + bf16-rounded q, k and raw v of the new token
        # the appended rows: e4m3 of the bf16 values (the device's fp32 RoPE may land one bf16 step off the reference rounding)
        assert np.abs(gk8[b, :, pb] - rk[0, :, 0]).max() <= 2 ** -3 * np.abs(rk).max() and np.array_equal(gv8[b, :, pb], e4m3_round(rv[0, :, 0]))
        keep = np.ones(S, bool)
        keep[pb] = False
        assert np.array_equal(gk8[b][:, keep], k_old[b][:, keep]) and np.array_equal(gv8[b][:, keep], v_old[b][:, keep])
        k_all = np.concatenate([k_old[b:b + 1, :, :pb], gk8[b:b + 1, :, pb:pb + 1]], 2)
        v_all = np.concatenate([v_old[b:b + 1, :, :pb], gv8[b:b + 1, :, pb:pb + 1]], 2)
        ref = cpu_ref.softmax_attention(torch.from_numpy(rq), torch.from_numpy(k_all), torch.from_numpy(v_all), scale, False,
                                        cpu_ref.Rounder(False))
        ref = ref.transpose(1, 2).reshape(D).numpy()
        err = np.abs(got[b] - ref).max()
        assert err < 2 ** -7 * max(1.0, np.abs(ref).max()), f"decode attention kv8 row {b} pos {pb}: abs err {err}"
