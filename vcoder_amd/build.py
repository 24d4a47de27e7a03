"""Builds libvcoder_hip.so (gfx950) in-tree with hipcc.  `python -m vcoder_amd.build` or __graft_entry__.build().

hipcc cross-compiles without a GPU; the .so is git-ignored but travels with the repo snapshot to the GPU box.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libvcoder_hip.so")
SOURCES = ["gemm.hip", "norm.hip", "attn.hip", "decode.hip", "misc.hip", "select.hip", "strict.hip", "preprocess.hip", "engine.hip", "comm.hip", "kernel_api.cpp"]
# (engine.hip is one translation unit made of engine_*.inc parts: they count as its headers for the staleness check)
HEADERS = ["vc_device.h", "kernels.h", "engine_ctx.h", "engine_weights.inc", "engine_linears.inc", "engine_vision.inc", "engine_llm.inc",
           "engine_abi.inc", "engine_pool.inc", "engine_profile.inc", os.path.join("..", "..", "include", "vcoder_hip.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-Wno-unused-result"]


def hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (needed to build libvcoder_hip.so for gfx950)")


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


LIB_F16 = os.path.join(LIBDIR, "libvcoder_hip_f16.so")


def build(force: bool = False, verbose: bool = True, operands: str = "bf16") -> str:
    """operands="bf16": libvcoder_hip.so (the benchmarked path); "fp16": libvcoder_hip_f16.so — the same sources with -DVC_F16 (fp16
    MFMA operands, include/vcoder_hip.h vc_operand_format); "all": both, returns the bf16 library's path."""
    if operands == "all":
        build(force, verbose, "fp16")
        return build(force, verbose, "bf16")
    f16 = operands == "fp16"
    return _build(force, verbose, LIB_F16 if f16 else LIB, "obj_f16" if f16 else "obj", ["-DVC_F16"] if f16 else [])


def _build(force: bool, verbose: bool, LIB: str, objname: str, extra) -> str:
    os.makedirs(LIBDIR, exist_ok=True)
    objdir = os.path.join(LIBDIR, objname)
    os.makedirs(objdir, exist_ok=True)
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    cc = hipcc()
    objs, procs = [], []
    for src in SOURCES:
        sp = os.path.join(CSRC, src)
        obj = os.path.join(objdir, os.path.splitext(src)[0] + ".o")
        objs.append(obj)
        if force or _stale(obj, [sp] + hdrs):
            cmd = [cc] + FLAGS + extra + ["-x", "hip", "-c", sp, "-o", obj]
            if verbose:
                print("[vcoder_amd.build]", " ".join(cmd), flush=True)
            procs.append((src, subprocess.Popen(cmd, cwd=CSRC)))
    for src, p in procs:
        if p.wait() != 0:
            raise RuntimeError(f"hipcc failed on {src}")
    if force or procs or _stale(LIB, objs):
        cmd = [cc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", LIB]
        if verbose:
            print("[vcoder_amd.build]", " ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, operands="all" if "--all" in sys.argv else ("fp16" if "--fp16" in sys.argv else "bf16")))
