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
    rows = c.execute("select name, start, end from kernels").fetchall()
    agg = {}
    for name, s, e in rows:
        k = short(name)
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
