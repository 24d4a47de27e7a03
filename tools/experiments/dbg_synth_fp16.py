"""Round 5 debugging aid: the device generator of the fp16- / fp32-valued checkpoint classes (vck_synth_f32_rounded) against
vcoder_amd/synth.py (numpy) over 2^20 elements of an offset tensor.  It found that hipcc contracted the generator's multiply-add
into one fma (-ffp-contract=fast; __fmul_rn / __fadd_rn are plain operators in this ROCm): 5 % of the fp32 values were an ulp off
numpy's.  Fixed with `#pragma clang fp contract(off)` in csrc/misc.hip synth_value_f32; prints 0 / 0 mismatches since."""
import sys, ctypes, numpy as np
sys.path[:0] = ['/root/repo', '/root/repo/oracle', '/root/repo/tests']
import kernel_cases as kc
from vcoder_amd import synth
be = kc.HipBackend()
name = "model.layers.3.mlp.up_proj.weight"; n = 1 << 20
ts = synth.tensor_seed(name, 42)
for rounding, code in (("fp16", 1), ("fp32", 2)):
    o = be.zeros((n,), "f32")
    be.lib.vck_synth_f32_rounded(be.ptr(o), ctypes.c_uint64(n), ctypes.c_uint32(ts), ctypes.c_float(1.0), ctypes.c_float(0.1), code, None)
    be.sync()
    got = be.host_f32(o); want = synth.synth_tensor(name, (n,), 42, 1.0, 0.1, rounding)
    bad = np.nonzero(got != want)[0]
    print(rounding, "mismatches", len(bad))
    raw = synth.synth_tensor(name, (n,), 42, 1.0, 0.1, "fp32")
    for i in bad[:6]:
        print("  ", i, repr(got[i]), repr(want[i]), hex(np.float32(got[i]).view(np.uint32)), hex(np.float32(want[i]).view(np.uint32)), "raw", hex(np.float32(raw[i]).view(np.uint32)))
