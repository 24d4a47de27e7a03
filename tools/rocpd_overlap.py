#!/usr/bin/env python
"""Splits the launches of the decode-step kernels in a rocprofv3 (rocpd sqlite) kernel trace of bench.py into those that ran ALONE
on the GPU and those that ran while a kernel of another in-flight call (encode / prefill: GEMMs, flash attention, norms, ...) was
executing, and reports count / average duration of each class per kernel.  This is the explanation of why a pooled decode step's
GEMV is slower "in situ" than in an isolated replay: a 32-row-span step that runs beside another session's prefill shares the CUs
with 256 x 256 GEMM workgroups.  usage: python tools/rocpd_overlap.py <results.db> [out.md]"""
import bisect
import re
import sqlite3
import sys


def short(name: str) -> str:
    name = re.sub(r"\(.*\)$", "", name)
    return name.replace("void ", "").replace("vc::", "")[:90]


STEP = ("gemv_dma_kernel", "gemv_wg_kernel", "attention_decode_fused_kernel", "select_embed_kernel", "stamp_")


def main():
    c = sqlite3.connect(sys.argv[1])
    rows = c.execute("select name, start, end from kernels").fetchall()
    step = [(short(n), s, e) for n, s, e in rows if any(t in n for t in STEP)]
    other = sorted((s, e) for n, s, e in rows if not any(t in n for t in STEP) and e - s > 20000)   # > 20 us: GEMMs, attention, norms
    starts = [s for s, _ in other]
    # running maximum of the end times, so that "some earlier interval still covers t" is one lookup
    run_end, m = [], 0
    for _, e in other:
        m = max(m, e)
        run_end.append(m)

    def overlapped(s, e):
        i = bisect.bisect_left(starts, e)          # intervals starting before this launch ends
        return i > 0 and run_end[i - 1] > s        # ... one of which ends after it starts
    agg = {}
    for n, s, e in step:
        a = agg.setdefault(n, [0, 0, 0, 0])
        if overlapped(s, e):
            a[2] += 1
            a[3] += e - s
        else:
            a[0] += 1
            a[1] += e - s
    lines = ["| decode-step kernel | alone: launches | alone: avg us | beside another call's kernels: launches | avg us | slowdown |",
             "|---|---|---|---|---|---|"]
    tot = [0, 0, 0, 0]
    for n, a in sorted(agg.items(), key=lambda kv: -(kv[1][1] + kv[1][3])):
        al = a[1] / a[0] / 1e3 if a[0] else float("nan")
        co = a[3] / a[2] / 1e3 if a[2] else float("nan")
        lines.append(f"| `{n}` | {a[0]} | {al:.2f} | {a[2]} | {co:.2f} | {co / al:.2f}x |" if a[0] and a[2] else
                     f"| `{n}` | {a[0]} | {al:.2f} | {a[2]} | {co:.2f} | |")
        for i in range(4):
            tot[i] += a[i]
    lines.append(f"| **all decode-step launches** | {tot[0]} | | {tot[2]} | | time beside other kernels: "
                 f"{100.0 * tot[3] / max(tot[1] + tot[3], 1):.1f} % of their total |")
    out = "\n".join(lines)
    print(out)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(out + "\n")


if __name__ == "__main__":
    main()
