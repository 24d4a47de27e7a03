#!/usr/bin/env python
"""ISA check for the count-waited register loads of csrc/vc_device.h (gld16_async / pin_loaded).

The compiler believes the destination of a gld16_async exists at once; the kernel waits for it by count and pins it before the first
use.  What must not happen is that the compiler touches such a register between the load and its pin (a copy, a spill, a move to an
AGPR would read bytes that have not landed), or that the register file spills at all.  This script compiles the given HIP source for
gfx950, and for every kernel that contains `vc_async_load` markers walks the ISA in program order with one state per destination
register: IN FLIGHT after a load, PINNED after its pin; any other instruction naming a register that is IN FLIGHT is an error.
(The kernels issue / consume in the textual order pin -> uses -> re-load, in straight-line code and inside their unrolled loop
bodies, so program-text order is the order that matters.)

usage: tools/check_async_loads.py [vcoder_amd/csrc/decode.hip] [-DVC_F16]   -> exit code 0 / 1
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REG = re.compile(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b")


def regs_of(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(3) is not None:
            out.add(int(m.group(3)))
        else:
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    return out


BRANCH = re.compile(r"^\s*(s_cbranch_\w+|s_branch)\s+(\.LBB\w+)")
LABEL = re.compile(r"^(\.LBB\w+):")


def classify(raw):
    """-> (kind, payload): 'load' (dst set, addr set) | 'pin' (regs) | 'drain' | 'other' (regs named) | None"""
    code, _, comment = raw.partition(";")
    if "vc_async_load" in comment:
        m = re.search(r"global_load_dwordx4\s+v\[(\d+):(\d+)\]\s*,\s*(v\[\d+:\d+\])", code)
        if not m:
            return "bad", raw.strip()
        return "load", (set(range(int(m.group(1)), int(m.group(2)) + 1)), regs_of(m.group(3)))
    if "vc_async_pin" in raw:
        return "pin", regs_of(raw.split("vc_async_pin", 1)[1])
    s = code.strip()
    if not s or s.startswith(".") or s.endswith(":"):
        return None, None
    if re.match(r"s_waitcnt\b.*vmcnt\(0\)", s):
        return "drain", None
    return "other", (regs_of(code), s)


def check_kernel(name, lines):
    """forward dataflow over the kernel's basic blocks: a VGPR is IN FLIGHT from an async load that targets it until its pin (or a
    vmcnt(0) wait); merging paths keeps IN FLIGHT if any predecessor has it.  Any other instruction that names a register which is
    IN FLIGHT at that point is an error."""
    # basic blocks
    blocks, cur, label_of = [], [], {}
    def close():
        nonlocal cur
        if cur:
            blocks.append(cur)
            cur = []
    for ln, raw in lines:
        m = LABEL.match(raw)
        if m:
            close()
            label_of[m.group(1)] = len(blocks)
        cur.append((ln, raw))
        if BRANCH.match(raw) or raw.strip().startswith("s_endpgm"):
            close()
    close()
    succ = []
    for bi, blk in enumerate(blocks):
        last = blk[-1][1]
        m = BRANCH.match(last)
        out = []
        if m:
            if m.group(2) in label_of:
                out.append(label_of[m.group(2)])
            if m.group(1) != "s_branch" and bi + 1 < len(blocks):
                out.append(bi + 1)
        elif not last.strip().startswith("s_endpgm") and bi + 1 < len(blocks):
            out.append(bi + 1)
        succ.append(out)

    def transfer(blk, state, report=None):
        state = set(state)
        for ln, raw in blk:
            kind, pay = classify(raw)
            if kind == "load":
                dst, addr = pay
                if report is not None and addr & state:
                    report.append((ln, f"address v{sorted(addr & state)} of an async load is itself in flight: " + raw.strip()))
                state |= dst
            elif kind == "pin":
                state -= pay
            elif kind == "drain":
                state.clear()
            elif kind == "other":
                regs, text = pay
                if report is not None:
                    if regs & state:
                        report.append((ln, f"v{sorted(regs & state)} in flight, touched by: " + text))
                    if "scratch_" in text:
                        report.append((ln, "scratch access in a kernel with async register loads: " + text))
            elif kind == "bad" and report is not None:
                report.append((ln, "unparsed async load: " + pay))
        return state

    entry = [set() for _ in blocks]
    work = list(range(len(blocks)))
    while work:
        bi = work.pop(0)
        out = transfer(blocks[bi], entry[bi])
        for sj in succ[bi]:
            if not out <= entry[sj]:
                entry[sj] |= out
                if sj not in work:
                    work.append(sj)
    errors = []
    for bi, blk in enumerate(blocks):
        transfer(blk, entry[bi], errors)
    loads = sum(1 for _, r in lines if "vc_async_load" in r)
    pins = sum(1 for _, r in lines if "vc_async_pin" in r)
    return loads, pins, errors


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("-D")]
    defs = [a for a in sys.argv[1:] if a.startswith("-D")]
    src = args[0] if args else os.path.join(ROOT, "vcoder_amd", "csrc", "decode.hip")
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
        subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-x", "hip", "--cuda-device-only", "-S", src, "-o", out] + defs,
                              stderr=subprocess.DEVNULL)
        text = open(out).read().splitlines()
    kernels, cur, name = [], None, None
    for i, l in enumerate(text):
        m = re.match(r"^(_Z\w+):", l)
        if m:
            name, cur = m.group(1), []
            kernels.append((name, cur))
        elif l.strip().startswith(".end_amdhsa_kernel") or l.strip().startswith("s_endpgm") and cur is not None and False:
            pass
        if cur is not None:
            cur.append((i + 1, l))
    bad = 0
    for name, lines in kernels:
        if not any("vc_async_load" in l for _, l in lines):
            continue
        loads, pins, errors = check_kernel(name, lines)
        spill = [l for _, l in lines if "ScratchSize:" in l or ".private_segment_fixed_size:" in l]
        scratch = [int(re.search(r"(\d+)", l.split(":", 1)[1]).group(1)) for l in spill if re.search(r"\d+", l.split(":", 1)[1])]
        vg = [l.strip() for _, l in lines if "; NumVgprs:" in l or "; NumAgprs:" in l]
        nscr = sum(1 for _, l in lines if re.search(r"\bscratch_(load|store)", l.partition(";")[0]))
        status = "ok" if not errors and not nscr else "FAIL"
        print(f"{status}  {name}: {loads} async loads, {pins} pins, scratch {max(scratch) if scratch else 0} B reserved / {nscr} scratch instructions, {' '.join(vg)}")
        for ln, e in errors[:12]:
            print(f"      line {ln}: {e}")
        if errors or nscr:
            bad += 1
    if bad:
        print(f"{bad} kernel(s) violate the async-load discipline")
        return 1
    return 0


if __name__ == "__main__":
    sys.exit(main())
