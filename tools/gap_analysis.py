#!/usr/bin/env python
"""Idle time between consecutive kernels of a rocprofv3 --kernel-trace CSV (same queue): how much of a denoise step is not
inside any kernel, and after which kernels the gaps sit.  usage: gap_analysis.py <..._kernel_trace.csv> [--last N]"""
import collections
import csv
import sys


def main():
    path = sys.argv[1]
    last = int(sys.argv[sys.argv.index("--last") + 1]) if "--last" in sys.argv else 600
    rows = []
    for r in csv.DictReader(open(path)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    rows = rows[-last:]
    busy = sum(e - s for s, e, _ in rows)
    span = rows[-1][1] - rows[0][0]
    gaps = collections.defaultdict(lambda: [0, 0])
    for (s0, e0, n0), (s1, e1, n1) in zip(rows, rows[1:]):
        g = max(0, s1 - e0)
        key = n0.split("(")[0][-60:] + "  ->  " + n1.split("(")[0][-60:]
        gaps[key][0] += g
        gaps[key][1] += 1
    print(f"{len(rows)} kernels, span {span / 1e6:.3f} ms, busy {busy / 1e6:.3f} ms, idle {(span - busy) / 1e6:.3f} ms "
          f"({100.0 * (span - busy) / span:.1f} %)")
    for k, (t, n) in sorted(gaps.items(), key=lambda kv: -kv[1][0])[:14]:
        print(f"  {t / 1e3:9.1f} us total  {t / n / 1e3:7.2f} us avg x{n:4d}   {k}")


if __name__ == "__main__":
    main()
