#!/usr/bin/env python
"""Reads a rocprofv3 --pmc FETCH_SIZE (rocpd sqlite) run over `tools/kbench.py gemv_rows dattn_rows` and reports the HBM read
bytes per launch of the two decode-step kernels, by the number of rows a launch serves, corrected as MI355X_MICROARCH.md
(HBM section) prescribes for gfx950: FETCH_SIZE counts 64 B per 128-B request on wide coalesced streams -> x2; FETCH_SIZE
unit = KiB.  usage: python tools/pmc_traffic.py <results.db> [out.json]

kbench runs, in order, M = 8, 16, 24, 32 for every GEMV shape (warm-up 3 + 40 timed launches each) and B = 8, 16, 24, 32
for the decode attention; launches are attributed to a row count by their kernel name (the GEMV's template arguments
include the 8-row piece count XP = M / 8; the attention grid's y extent is B)."""
import json
import re
import sqlite3
import sys

db = sys.argv[1]
c = sqlite3.connect(db)
rows = c.execute("select kernel_name, grid_size, workgroup_size, counter_name, value from counters_collection").fetchall()
gemv, att = {}, {}
for k, g, w, cn, v in rows:
    if cn != "FETCH_SIZE":
        continue
    m = re.search(r"gemv_dma_kernel<([^>]*)>", k)
    if m:
        args = [a.strip() for a in m.group(1).split(",")]
        xp = int(args[5]) if len(args) > 5 else 2          # <WAVES, NT, R, EPI, FP8, XP>
        a = gemv.setdefault(8 * xp, {}).setdefault(m.group(0), [0, 0.0])
        a[0] += 1
        a[1] += float(v)
    elif "attention_decode_fused_kernel" in k:
        nrows = int(g) // (512 * 32) if g else 0           # grid = (H = 32 heads, B rows) x 512 threads
        a = att.setdefault(nrows, [0, 0.0])
        a[0] += 1
        a[1] += float(v)

# one decode step of VCoder-DS 7b: 32 x (qkv + o + gate_up + down) + lm_head; shapes are told apart by their template
# arguments: EPI 0 = qkv (bf16), 2 = o / down (residual), 3 = gate_up (SwiGLU), 1 = lm_head (fp32)
out = {"source": "rocprofv3 --pmc FETCH_SIZE -- python tools/kbench.py gemv_rows dattn_rows (separate PMC pass; weights / KV rotated over "
                 "several copies so the Infinity Cache cannot hold them)",
       "correction": "FETCH_SIZE(KiB) * 1024 * 2 — gfx950 rocprofv3 counts 64 B per 128-B request on wide coalesced streams "
                     "(MI355X_MICROARCH.md, HBM section)",
       "gemv_per_kernel_MB_corrected": {}, "hbm_read_bytes_per_launch_by_rows": {}, "attention_hbm_read_bytes_per_launch_by_rows": {}}
for M, ks in sorted(gemv.items()):
    per = {}
    tot = 0.0
    for name, (n, kib) in ks.items():
        b = 2 * kib * 1024.0 / n
        per[name] = round(b / 1e6, 2)
        epi = int(name.split(",")[3])
        tot += b * ({0: 32, 2: 64, 3: 32, 1: 1}[epi])       # o and down share one instantiation: their mean x 64 launches
    out["gemv_per_kernel_MB_corrected"][str(M)] = per
    out["hbm_read_bytes_per_launch_by_rows"][str(M)] = tot / 129.0
for B, (n, kib) in sorted(att.items()):
    out["attention_hbm_read_bytes_per_launch_by_rows"][str(B)] = 2 * kib * 1024.0 / n
out["algorithmic_gemv_bytes_per_launch"] = (2.0 * 32 * (4 * 4096 * 4096 + 3 * 4096 * 11008) + 2.0 * 4096 * 32000) / 129.0
print(json.dumps(out, indent=1))
if len(sys.argv) > 2:
    json.dump(out, open(sys.argv[2], "w"), indent=1)
