#!/usr/bin/env python
"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel count / total / avg / share.
usage: python tools/rocpd_summary.py <results.db> [out.md]"""
import re
import sqlite3
import sys


def short(name: str) -> str:
    name = re.sub(r"\(.*\)$", "", name)
    name = name.replace("void ", "").replace("vc::", "")
    return name[:90]


def main():
    db = sys.argv[1]
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    # total work-items of the launch, whatever this rocprofv3 calls the columns: the decode attention keeps ONE kernel name for every
    # row count (grid = heads x rows workgroups of 512 threads), so its launches are told apart by the grid
    gcols = [g for g in ("grid_size", "grid_x", "grid_y", "grid_z", "grid_size_x", "grid_size_y", "grid_size_z") if g in cols]
    sel = "name, start, end" + "".join(", " + g for g in gcols)
    rows = c.execute(f"select {sel} from kernels").fetchall()
    agg = {}
    for row in rows:
        name, s, e = row[:3]
        k = short(name)
        if "attention_decode_fused_kernel" in name and gcols:
            g = dict(zip(gcols, row[3:]))
            total = g.get("grid_size") or ((g.get("grid_x") or g.get("grid_size_x") or 1) * (g.get("grid_y") or g.get("grid_size_y") or 1) *
                                           (g.get("grid_z") or g.get("grid_size_z") or 1))
            k += f" [{int(total) // 512} workgroups = heads x rows]"
        a = agg.setdefault(k, [0, 0])
        a[0] += 1
        a[1] += e - s
    total = sum(v[1] for v in agg.values())
    lines = ["| kernel | calls | total ms | avg us | % |", "|---|---|---|---|---|"]
    for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        lines.append(f"| `{k}` | {n} | {t / 1e6:.3f} | {t / n / 1e3:.2f} | {100.0 * t / total:.1f} |")
    lines.append(f"| **total kernel time** | {len(rows)} | {total / 1e6:.3f} | | 100 |")
    out = "\n".join(lines)
    print(out)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(out + "\n")


if __name__ == "__main__":
    main()
