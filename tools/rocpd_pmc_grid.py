#!/usr/bin/env python
"""Per-(kernel, grid) average of one PMC counter from a rocprofv3 rocpd database: pmc_events joined to the kernel
dispatches by dispatch_id, so launches of one kernel with different grids (e.g. sp3_gemm vs sp3_gemm2: grid.y 2 vs 4)
are reported apart.  Usage: python tools/rocpd_pmc_grid.py results.db [--json out.json]"""
import collections, json, sqlite3, sys
db = sys.argv[1]
c = sqlite3.connect(db)
names = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
kt = None
for n in names:
    cols = [r[1] for r in c.execute("pragma table_info(%s)" % n)]
    if "dispatch_id" in cols and any(x.startswith("grid") for x in cols) and n != "pmc_events":
        kt, kcols = n, cols
        break
if kt is None:
    print("no dispatch table with grid columns; tables:", names)
    sys.exit(1)
g = [x for x in kcols if x.startswith("grid")][:3]
w = [x for x in kcols if x.startswith("workgroup")][:3]
grid = {}
for row in c.execute("select dispatch_id, %s from %s" % (", ".join(g + w), kt)):
    grid[row[0]] = tuple(row[1:])
agg = collections.defaultdict(lambda: [0, 0.0])
cname = None
for name, did, counter, value in c.execute("select name, dispatch_id, counter_name, counter_value from pmc_events"):
    a = agg[(name, grid.get(did))]
    a[0] += 1; a[1] += value
    cname = counter
out = {"counter": cname, "dispatch_table": kt, "grid_columns": g + w, "rows": []}
print("| kernel | grid / workgroup | launches | %s per launch |" % cname)
print("|---|---|---|---|")
for i, ((n, gr), (k, v)) in enumerate(sorted(agg.items(), key=lambda kv: -kv[1][1])):
    if i < 40:                                   # the table: the 40 largest; the JSON: every (kernel, grid) -- round 5's file cut both at 40,
        print("| %s | %s | %d | %.2f |" % (n.replace("(anonymous namespace)::", "")[:100], gr, k, v / k))   # and four small writers lost their WRITE rows
    out["rows"].append({"kernel": n, "grid": gr, "launches": k, "per_launch": v / k})
if "--json" in sys.argv:
    json.dump(out, open(sys.argv[sys.argv.index("--json") + 1], "w"), indent=1)
