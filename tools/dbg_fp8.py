import sys, ctypes, numpy as np, torch
sys.path.insert(0, "tests"); sys.path.insert(0, "oracle")
import kernel_cases as kc
from vcoder_amd import quant
be = kc.HipBackend()
N, K, M = 16, 64, 16
# weights: bytes 0..255 spread over [N,K] = 1024 entries; scale 1
codes = (np.arange(N * K) % 256).astype(np.uint8).reshape(N, K)
codes[codes == 0x7F] = 0; codes[codes == 0xFF] = 0
Wq = torch.from_numpy(quant.pack_supertiles(codes)).cuda()
sc = torch.ones(N, device="cuda")
for kk in range(0, K, 16):
    X = np.zeros((M, K), np.float32)
    for m in range(16):
        X[m, kk + m] = 1.0
    Xd = be.bf16(X)
    out = be.zeros((M, N), "f32")
    be.lib.vck_gemv_fp8(None, None, None, ctypes.c_int(16), ctypes.c_float(1e-5), be.ptr(Xd), be.ptr(Wq), be.ptr(sc), be.ptr(out), None, M, N, K, N, 1, None)
    be.sync()
    got = be.host_f32(out)          # got[m][n] = W[n][kk+m]
    ref = quant.e4m3_decode(codes)[:, kk:kk + 16].T
    bad = np.argwhere(got != ref)
    print("kk", kk, "mismatches", len(bad))
    for m, n in bad[:6]:
        print("  k", kk + m, "n", n, "code", hex(codes[n, kk + m]), "got", got[m, n], "ref", ref[m, n])
