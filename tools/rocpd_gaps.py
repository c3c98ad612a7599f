#!/usr/bin/env python
"""GPU idle analysis of a rocprofv3 rocpd database: union of kernel-busy intervals, idle gaps between them, and the
critical-path view per stream.  Usage: python tools/rocpd_gaps.py results.db [last_fraction]"""
import sqlite3, sys
db = sys.argv[1]
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.4
c = sqlite3.connect(db)
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
namecol = "name" if "name" in cols else "kernel_name"
qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
rows = c.execute("select start, end, %s%s from kernels order by start" % (namecol, (", " + qcol) if qcol else "")).fetchall()
t0, t1 = rows[0][0], max(r[1] for r in rows)
lo = t1 - frac * (t1 - t0)                      # steady state: the last part of the run
rows = [r for r in rows if r[0] >= lo]
busy, gaps, cur_s, cur_e = 0, [], rows[0][0], rows[0][1]
for r in rows[1:]:
    if r[0] > cur_e:
        busy += cur_e - cur_s
        gaps.append((r[0] - cur_e, r[2]))
        cur_s, cur_e = r[0], r[1]
    else:
        cur_e = max(cur_e, r[1])
busy += cur_e - cur_s
span = rows[-1][1] - rows[0][0] if rows else 0
span = max(r[1] for r in rows) - rows[0][0]
print("window %.2f ms, %d kernels, busy (>=1 kernel running) %.2f ms = %.1f %%, idle %.2f ms in %d gaps" %
      (span / 1e6, len(rows), busy / 1e6, 100.0 * busy / span, (span - busy) / 1e6, len(gaps)))
ksum = sum(r[1] - r[0] for r in rows)
print("sum of kernel durations %.2f ms -> average concurrency while busy %.2f" % (ksum / 1e6, ksum / busy))
import collections
h = collections.Counter()
for g, _ in gaps:
    b = 1 if g < 1000 else 2 if g < 2000 else 5 if g < 5000 else 10 if g < 10000 else 50 if g < 50000 else 1000
    h[b] += g
print("idle time by gap size (us bucket upper bound -> ms):", {k: round(v / 1e6, 2) for k, v in sorted(h.items())})
big = sorted(gaps, reverse=True)[:8]
print("largest gaps (us, next kernel):", [(round(g / 1e3, 1), n[:40]) for g, n in big])
after = collections.Counter()
for g, n in gaps:
    after[n[:60]] += g
print("idle before kernel (ms):", [(n, round(v / 1e6, 2)) for n, v in after.most_common(8)])
# context of the larger gaps: the kernel before and the kernel after
allr = rows
ends = []
cur_e = allr[0][1]; last_name = allr[0][2]
ctx = []
for r in allr[1:]:
    if r[0] > cur_e and r[0] - cur_e > 15000:
        ctx.append((r[0] - cur_e, last_name[:50], r[2][:50]))
    if r[1] > cur_e:
        cur_e = r[1]; last_name = r[2]
agg = collections.Counter(); cnt = collections.Counter()
for g, a, b in ctx:
    agg[(a, b)] += g; cnt[(a, b)] += 1
print("gaps > 15 us by (kernel before -> kernel after): total ms, count")
for k, v in agg.most_common(12):
    print("   %6.2f ms  x%-3d %s  ->  %s" % (v / 1e6, cnt[k], k[0], k[1]))
