#!/usr/bin/env python
"""Reads a rocprofv3 --pmc FETCH_SIZE (rocpd sqlite) run over `tools/kbench.py gemm` and reports, per GEMM launch shape, the HBM-side
read bytes against the compulsory ones (A once + W once [+ the fp32 residual a RESID epilogue reads]) — corrected as
MI355X_MICROARCH.md's HBM section prescribes for gfx950 (FETCH_SIZE in KiB, x2 for wide coalesced streams; Infinity-Cache hits
are counted, so this is traffic BEHIND the L2, not DRAM traffic).  usage: python tools/pmc_gemm_traffic.py <results.db> [out.md]

kbench gemm launches, per shape, vck_gemm (no workspace) and vck_gemm_ws (split-K remainder round) 3 + 10 times each: the two forms
are told apart by their grids (the split-K form launches more workgroups)."""
import re
import sqlite3
import sys

# (llm o and llm down launch the same grid with the same epilogue: one row, K = the mean of the two)
SHAPES = [(9728, 12288, 4096, 0, "llm qkv"), (9728, 4096, (4096 + 11008) / 2, 4, "llm o + down (mean of the two)"), (9728, 22016, 4096, 5, "llm gate-up (SwiGLU)"),
          (9728, 22016, 4096, 0, "gate-up / bf16 out"), (9728, 22016, 4096, 3, "gate-up / f32 out"),
          (13848, 3072, 1024, 0, "vit qkv"), (13848, 4096, 1024, 1, "vit fc1"), (13848, 1024, 4096, 4, "vit fc2"),
          (13824, 4096, 4096, 0, "adapter 2")]
c = sqlite3.connect(sys.argv[1])
rows = c.execute("select kernel_name, grid_size, counter_name, value from counters_collection").fetchall()
agg = {}
for k, g, cn, v in rows:
    if cn != "FETCH_SIZE":
        continue
    m = re.search(r"gemm_bf16_8phase_kernel<(\d+)", k)
    if not m:
        continue
    a = agg.setdefault((int(m.group(1)), int(g)), [0, 0.0])
    a[0] += 1
    a[1] += float(v)
lines = ["| shape (M x N x K, epilogue) | grid (workgroups) | FETCH_SIZE x 2 per launch | compulsory A + W (+ residual) | ratio | with perfect sharing inside each XCD's 4 x 8 tile block | ratio |",
         "|---|---|---|---|---|---|---|"]
for (M, N, K, epi, name) in SHAPES:
    t256 = ((M + 255) // 256) * ((N + 255) // 256)
    comp = 2.0 * M * K + 2.0 * N * K + (4.0 * M * N if epi == 4 else 0.0)
    # the two launch forms of kbench: t256 workgroups, and (csrc/gemm.hip launch_gemm) the split-K remainder round when the last
    # round holds at most 128 tiles: its rem tiles as ks = min(256 / rem, 8, K / 64) K-slices each
    rem = t256 % 256
    ks = min(256 // rem, 8, int(K) // 64) if (t256 > 256 and 0 < rem <= 128) else 1
    grids = {t256: ""}
    if ks > 1:
        grids[t256 - rem + rem * ks] = f" (split-K round: {rem} tiles x {ks})"
    # the XCD-partitioned L2: an XCD works on a block of gm x gn = 4 x 8 tiles at a time (tile_group = 4 m-tiles sweep n), so even
    # with perfect sharing inside the XCD every tile costs A_panel / gn + W_panel / gm behind the L2
    per_xcd = t256 * (2.0 * 256 * K / 8 + 2.0 * 256 * K / 4) + (4.0 * M * N if epi == 4 else 0.0)
    for (e, g), (n, kib) in sorted(agg.items()):
        wgs = g // 512
        if e != epi or wgs not in grids:
            continue
        b = 2.0 * 1024.0 * kib / n
        lines.append(f"| {name} {M} x {N} x {K:g}, epi {epi} | {wgs}{grids[wgs]} | {b / 1e6:.0f} MB | {comp / 1e6:.0f} MB | {b / comp:.2f} | {per_xcd / 1e6:.0f} MB | {b / per_xcd:.2f} |")
out = "\n".join(lines)
print(out)
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write("FETCH_SIZE of the 8-phase GEMM at the engine's shapes (tools/pmc_gemm_traffic.py; rocprofv3 --pmc FETCH_SIZE -- python tools/kbench.py gemm).\n"
                                 "Traffic behind the L2 (Infinity-Cache hits included), gfx950 x2 correction applied.\n\n" + out + "\n")
