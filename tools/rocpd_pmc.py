#!/usr/bin/env python
"""Per-kernel average of one PMC counter from a rocprofv3 rocpd database (view pmc_events).
Usage: python tools/rocpd_pmc.py results.db [--json out.json]"""
import collections, json, re, sqlite3, sys
db = sys.argv[1]
c = sqlite3.connect(db)
agg = collections.defaultdict(lambda: [0, 0.0, 0])
cname = None
for name, counter, value, dur in c.execute("select name, counter_name, counter_value, duration from pmc_events"):
    a = agg[name]
    a[0] += 1; a[1] += value; a[2] += dur
    cname = counter
rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
out = {"counter": cname, "kernels": {}}
print("| kernel | launches | %s total | per launch | avg us (serialised by the counter pass) |" % cname)
print("|---|---|---|---|---|")
for n, (k, v, d) in rows[:14]:
    short = re.sub(r"\(anonymous namespace\)::", "", n)[:90]
    print("| %s | %d | %.1f | %.2f | %.1f |" % (short, k, v, v / k, d / k / 1e3))
    out["kernels"][n] = {"launches": k, "per_launch": v / k}
if "--json" in sys.argv:
    json.dump(out, open(sys.argv[sys.argv.index("--json") + 1], "w"), indent=1)
