#!/usr/bin/env python
"""Average every counter of a rocprofv3 --pmc CSV (counter_collection.csv) per (kernel name, grid).
Usage: pmc_csv.py file.csv [substring ...]   -- kernels whose name contains one of the substrings (default: gemm)"""
import csv
import re
import sys
from collections import defaultdict

agg = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for r in csv.DictReader(open(sys.argv[1])):
    k = re.sub(r"\(.*", "", r.get("Kernel_Name", "?").replace("(anonymous namespace)::", "").replace("void ", ""))[:90]
    a = agg[(k, r.get("Grid_Size", ""))][r["Counter_Name"]]
    a[0] += float(r["Counter_Value"])
    a[1] += 1
for (k, g), cs in sorted(agg.items()):
    if not any(w in k for w in (sys.argv[2:] or ["gemm"])):
        continue
    print(k, "grid", g)
    for c, (s, n) in sorted(cs.items()):
        print("    %-28s %16.1f  (avg of %d dispatches)" % (c, s / n, n))
