#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd database (`rocprofv3 --kernel-trace --stats ...`) into a markdown table.

usage: rocpd_summary.py results.db [out.md] [--steady] [--per N_KERNEL_SUBSTR:COUNT]

--steady           drop everything up to the last MIOpen `naive_conv*` dispatch (MIOpen's first-call solver
                   search runs naive reference kernels; they are warm-up artefacts, not the timed region)
--per SUBSTR:COUNT normalise to "per forward": a forward contains COUNT dispatches of the kernel whose
                   name contains SUBSTR (e.g. --per "conv3x3_kernel<2, 2, 4, 4, 2>:32")
Durations in the rocpd `kernels` view are nanoseconds."""
import sqlite3
import sys
from collections import defaultdict


def main():
    args = sys.argv[1:]
    steady = "--steady" in args
    per = None
    if "--per" in args:
        i = args.index("--per")
        sub, cnt = args[i + 1].rsplit(":", 1)
        per = (sub, int(cnt))
        del args[i:i + 2]
    args = [a for a in args if a != "--steady"]
    con = sqlite3.connect(args[0])
    rows = con.execute("select name, start, end from kernels order by start").fetchall()
    if steady:
        naive = [e for n, s, e in rows if n.startswith("naive_conv")]
        if naive:
            rows = [r for r in rows if r[1] > max(naive)]
    fw = 1.0
    if per:
        fw = sum(1 for n, _, _ in rows if per[0] in n) / per[1]
    agg = defaultdict(lambda: [0, 0])
    for n, s, e in rows:
        agg[n][0] += 1
        agg[n][1] += e - s
    tot = sum(d for _, d in agg.values())
    unit = "per forward" if per else "total"
    lines = [f"| kernel | calls ({unit}) | ms ({unit}) | avg us | % of kernel time |", "|---|---:|---:|---:|---:|"]
    for n, (c, d) in sorted(agg.items(), key=lambda x: -x[1][1]):
        if d / tot < 0.0005:
            continue
        short = n if len(n) < 100 else n[:97] + "..."
        lines.append(f"| `{short}` | {c / fw:.1f} | {d / 1e6 / fw:.3f} | {d / c / 1e3:.1f} | {100 * d / tot:.2f} |")
    lines.append("")
    lines.append(f"forwards in window: {fw:.2f}; kernel time {tot / 1e6 / fw:.2f} ms {unit}; "
                 f"window span {(rows[-1][2] - rows[0][1]) / 1e6 / fw:.2f} ms {unit} (includes host gaps and, under the profiler, tracing overhead)")
    text = "\n".join(lines)
    if len(args) > 1:
        open(args[1], "w").write(text + "\n")
    print(text)


if __name__ == "__main__":
    main()
