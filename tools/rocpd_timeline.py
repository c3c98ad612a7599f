#!/usr/bin/env python
"""Kernel timeline of one steady-state window from a rocprofv3 rocpd database.
Usage: python tools/rocpd_timeline.py results.db [start_fraction] [window_ms]"""
import sqlite3, sys, re
db = sys.argv[1]
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.9
win = float(sys.argv[3]) if len(sys.argv) > 3 else 3.0
c = sqlite3.connect(db)
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
namecol = "name" if "name" in cols else "kernel_name"
qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else "0")
rows = c.execute("select start, end, %s, %s from kernels order by start" % (namecol, qcol)).fetchall()
t0, t1 = rows[0][0], max(r[1] for r in rows)
lo = t0 + frac * (t1 - t0)
hi = lo + win * 1e6
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n); n = re.sub(r"^void ", "", n)
    m = re.match(r"_ZN12_GLOBAL__N_111gemm_kernelI(\w+?)Li(\d)ELi(\d)ELi(\d)ELi(\d)ELi(\d)ELi(\d)", n)
    if m:
        return "gemm<%s ld%s %sx%s w%sx%sxk%s>" % (m.group(1), m.group(2), m.group(3), m.group(4), m.group(5), m.group(6), m.group(7))
    return n[:44]
qs = {}
prev_end = None
for s, e, n, q in rows:
    if s < lo or s > hi:
        continue
    qi = qs.setdefault(q, len(qs))
    gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
    print("%9.1f us  +%6.1f  dur %7.1f  q%d  %s" % ((s - lo) / 1e3, gap, (e - s) / 1e3, qi, short(n)))
    prev_end = max(prev_end or 0, e)
