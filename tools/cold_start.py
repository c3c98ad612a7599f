#!/usr/bin/env python
"""Cold path of Spann3R.forward: the FIRST call of a fresh model (demo.py:123-127 calls forward once per scene), the calls after
it, and the eager (no hipGraph) steady state.  Weight packing (the engine build) is excluded: it is a per-checkpoint cost.
Needs an MI355X.

  python tools/cold_start.py [--frames 10] [--size 224] [--profile]      (--profile: cProfile of the first call's host time)
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from spann3r_amd import Spann3R, FULL  # noqa: E402
from spann3r_amd.runner import make_sequence  # noqa: E402
from spann3r_amd.weights import synth_state_dict  # noqa: E402


def fresh(precision, train_policy):
    m = Spann3R(dus3r_name=None, cfg=FULL, init_weights=False)
    m.load_state_dict(synth_state_dict(0, FULL))
    m = m.cuda().eval().set_precision(precision)
    if train_policy:
        m.train()
        m.mem_dropout.eval()
    m.engine                      # weight packing: per checkpoint, not per call
    torch.cuda.synchronize()
    return m


def timed(m, seq):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with torch.no_grad():
        m(seq)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    return t1 - t0, t2 - t0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=10)
    ap.add_argument("--size", type=int, default=224)
    ap.add_argument("--precision", default="bf16")
    ap.add_argument("--train-policy", action="store_true")
    ap.add_argument("--profile", action="store_true")
    ap.add_argument("--calls", type=int, default=6)
    ap.add_argument("--empty-allocator", action="store_true", help="torch.cuda.empty_cache() AFTER the engine build: the first call then "
                    "pays a device allocation for every buffer (a fresh process leaves the blocks the weight packing freed in torch's cache)")
    a = ap.parse_args()
    seqs = [make_sequence(s, a.frames, a.size, a.size, device="cuda") for s in range(a.calls)]
    # a throw-away model first: the process-level one-time costs (library load, HIP module load of every kernel, allocator warm-up)
    # are not the model's cold path either
    w = fresh(a.precision, a.train_policy)
    timed(w, seqs[0])
    del w
    torch.cuda.empty_cache()
    m = fresh(a.precision, a.train_policy)
    if a.empty_allocator:
        torch.cuda.empty_cache()
    if a.profile:
        import cProfile
        import pstats
        pr = cProfile.Profile()
        pr.enable()
        h, t = timed(m, seqs[0])
        pr.disable()
        pstats.Stats(pr).sort_stats("tottime").print_stats(25)
        print("call 1 (profiled): host %.2f ms, wall %.2f ms" % (1e3 * h, 1e3 * t))
    rows = []
    for i in range(0 if not a.profile else 1, a.calls):
        h, t = timed(m, seqs[i])
        rows.append((i + 1, h, t))
        print("call %d: host %.2f ms, wall %.2f ms -> %.1f frames/s" % (i + 1, 1e3 * h, 1e3 * t, a.frames / t))
    run = list(m._runners.values())[0]
    print("graphs held: %d, keys seen: %d" % (len(run.graphs), len(run.seen)))
    m.use_graphs = False
    for i in range(3):
        h, t = timed(m, seqs[i])
        print("eager %d: host %.2f ms, wall %.2f ms -> %.1f frames/s" % (i + 1, 1e3 * h, 1e3 * t, a.frames / t))


if __name__ == "__main__":
    main()
