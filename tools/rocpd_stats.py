#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd database (…_results.db): per-kernel launches, total/avg/min/max duration.
Usage: python tools/rocpd_stats.py gpurun_out/prof/bench_results.db [--md]"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*\)$", "", name)
    return name if len(name) < 110 else name[:107] + "..."


def main():
    db = sys.argv[1]
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    namecol = "name" if "name" in cols else "kernel_name"
    rows = c.execute("select %s, start, end from kernels" % namecol).fetchall()
    agg = {}
    for n, s, e in rows:
        a = agg.setdefault(short(n), [0, 0, 1 << 62, 0])
        d = e - s
        a[0] += 1
        a[1] += d
        a[2] = min(a[2], d)
        a[3] = max(a[3], d)
    tot = sum(a[1] for a in agg.values())
    t0, t1 = min(r[1] for r in rows), max(r[2] for r in rows)
    print("| kernel | launches | total ms | avg us | min us | max us | % |")
    print("|---|---|---|---|---|---|---|")
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print("| %s | %d | %.3f | %.2f | %.2f | %.2f | %.1f |" % (k, a[0], a[1] / 1e6, a[1] / a[0] / 1e3, a[2] / 1e3, a[3] / 1e3, 100.0 * a[1] / tot))
    print("\nkernel time total %.3f ms over %d launches; first-to-last kernel span %.3f ms" % (tot / 1e6, len(rows), (t1 - t0) / 1e6))


if __name__ == "__main__":
    main()
