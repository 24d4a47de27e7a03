#!/usr/bin/env python
"""MFMA-pipe busy fraction per kernel from a tools/pmc_summary.py text of a rocprofv3 --pmc SQ pass (SQ_VALU_MFMA_BUSY_CYCLES,
GRBM_GUI_ACTIVE, SQ_LDS_BANK_CONFLICT, SQ_LDS_IDX_ACTIVE collected over `tools/kbench.py gemm attn gemm_f8`).
  busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs)
SQ_VALU_MFMA_BUSY_CYCLES is summed over the chip's SIMDs and equals 16 cycles x the number of 16x16x32 bf16 MFMAs of the problem;
GRBM_GUI_ACTIVE is summed over the 8 XCDs.  usage: python tools/pmc_mfma_busy.py <pmc_sq_summary.txt> [out.md]"""
import re
import sys


def main():
    txt = open(sys.argv[1]).read()
    rows = []
    for b in re.split(r"\n(?=\S)", txt):
        lines = b.strip().split("\n")
        if not lines or "grid=" not in lines[0]:
            continue
        d = {}
        for l in lines[1:]:
            m = re.match(r"\s+(\S+)\s+n=\s*(\d+)\s+avg=\s*([\d.]+)", l)
            if m:
                d[m.group(1)] = float(m.group(3))
        if d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) <= 0 or d.get("GRBM_GUI_ACTIVE", 0) <= 0:
            continue
        cyc = d["GRBM_GUI_ACTIVE"] / 8.0
        rows.append((lines[0], d["SQ_VALU_MFMA_BUSY_CYCLES"], cyc, d["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * cyc),
                     d.get("SQ_LDS_BANK_CONFLICT", 0), d.get("SQ_LDS_IDX_ACTIVE", 0)))
    out = ["| kernel (grid = work-items) | MFMA busy cycles (all SIMDs) | kernel cycles | MFMA busy | LDS bank-conflict cycles / LDS active |",
           "|---|---|---|---|---|"]
    for n, mb, cyc, u, bc, la in rows:
        out.append(f"| `{n[:110]}` | {mb / 1e6:.1f} M | {cyc / 1e6:.3f} M | **{100 * u:.1f} %** | {bc:.0f} / {la / 1e6:.1f} M |")
    text = "\n".join(out)
    print(text)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(text + "\n")


if __name__ == "__main__":
    main()
