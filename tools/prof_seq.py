#!/usr/bin/env python
"""Per-kernel HIP-event breakdown of one sequence (eager launches): python tools/prof_seq.py [--size 224] [--frames 10] [--train-policy] [--precision bf16]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from spann3r_amd import ops  # noqa: E402
from spann3r_amd.runner import make_sequence  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=224)
ap.add_argument("--frames", type=int, default=10)
ap.add_argument("--precision", default="bf16")
ap.add_argument("--train-policy", action="store_true")
args = ap.parse_args()
dev = torch.device("cuda", 0)
model, _ = bench.build_model(args.precision, dev, args.train_policy)
seq = make_sequence(0, args.frames, args.size, args.size, device=dev)
model.use_graphs = False
model(seq)
model(seq)
prof = ops.Profiler()
print("event bracket cost %.2f us" % (1e3 * prof.calibrate()))
ops.set_profiler(prof)
model(seq)
ops.set_profiler(None)
agg = prof.summary()
tot = sum(a["ms"] for a in agg.values())
print("%-46s %7s %9s %8s %9s %9s" % ("kernel", "launch", "ms", "avg us", "TFLOP/s", "GB/s"))
for k, a in sorted(agg.items(), key=lambda kv: -kv[1]["ms"]):
    print("%-46s %7d %9.3f %8.2f %9.1f %9.1f" % (k, a["launches"], a["ms"], 1e3 * a["ms"] / a["launches"], a["flops"] / a["ms"] / 1e9, a["bytes"] / a["ms"] / 1e6))
print("total %.3f ms" % tot)
for r in prof.region_summary():
    print("region %s M=%d: %.1f us over %d launches" % (r["key"], r["info"]["M"], 1e3 * r["ms"], r["launches"]))
