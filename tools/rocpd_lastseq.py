#!/usr/bin/env python
"""Kernel list of the LAST `ms` milliseconds of a rocprofv3 rocpd database (one steady-state sequence), aggregated by
kernel name in launch order of first appearance, plus the raw timeline.  Usage: rocpd_lastseq.py results.db [ms]"""
import sqlite3, sys, re, collections
db = sys.argv[1]
ms = float(sys.argv[2]) if len(sys.argv) > 2 else 23.0
c = sqlite3.connect(db)
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
namecol = "name" if "name" in cols else "kernel_name"
rows = c.execute("select start, end, %s from kernels order by start" % namecol).fetchall()
t1 = max(r[1] for r in rows)
lo = t1 - ms * 1e6
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n); n = re.sub(r"^void ", "", n)
    m = re.match(r"_ZN12_GLOBAL__N_111gemm_kernelI(\w+?)Li(\d)ELi(\d+)ELi(\d+)ELi(\d)ELi(\d)ELi(\d+)", n)
    if m:
        return "gemm<%s ld%s %sx%s w%sx%sxk%s>" % m.groups()
    return n[:48]
sel = [(s, e, short(n)) for s, e, n in rows if s >= lo]
agg = collections.OrderedDict()
for s, e, n in sel:
    a = agg.setdefault(n, [0, 0.0])
    a[0] += 1; a[1] += (e - s) / 1e3
busy = sum(e - s for s, e, _ in sel) / 1e6
print("last %.1f ms: %d kernels, sum of durations %.2f ms" % (ms, len(sel), busy))
for n, (k, us) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%6d x %8.2f us = %8.1f us  %s" % (k, us / k, us, n))
print("---- timeline")
prev = None
for s, e, n in sel:
    print("%9.1f +%5.1f %7.1f  %s" % ((s - lo) / 1e3, 0.0 if prev is None else (s - prev) / 1e3, (e - s) / 1e3, n))
    prev = max(prev or 0, e)
