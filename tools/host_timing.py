#!/usr/bin/env python
"""Where does the HOST spend its time per step of Spann3R.forward (graph replays, read-backs, clones)?  Needs an MI355X."""
import os, sys, time, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from spann3r_amd import Spann3R, FULL, model as M
from spann3r_amd.runner import make_sequence
from spann3r_amd.weights import synth_state_dict
m = Spann3R(dus3r_name=None, cfg=FULL, init_weights=False); m.load_state_dict(synth_state_dict(0, FULL)); m = m.cuda().eval().set_precision("bf16")
for off in filter(None, (sys.argv[1] if len(sys.argv) > 1 else "").split(",")):
    setattr(m, off, False)
seq = make_sequence(0, 10, 224, 224, device="cuda")
for _ in range(3):
    m(seq)
torch.cuda.synchronize()
acc = collections.defaultdict(float)
def wrap(obj, name, label=None):
    f = getattr(obj, name)
    def g(*a, **k):
        t = time.perf_counter(); r = f(*a, **k); acc[label or name] += time.perf_counter() - t; return r
    setattr(obj, name, g)
run = list(m._runners.values())[0]
orig_graphed = run._graphed
def graphed(key, fn, ug):
    t = time.perf_counter(); orig_graphed(key, fn, ug); acc["replay:" + key[0]] += time.perf_counter() - t
run._graphed = graphed
for nm in ("save_dec2", "load_pair", "finish_head2", "encode_sequence"):
    wrap(run, nm)
wrap(run.mem, "sim_verdict"); wrap(run.mem, "finish_staged"); wrap(run.mem, "fetch_scores_async")
wrap(run, "run", "run(total)")
N = 10
t0 = time.perf_counter()
for _ in range(N):
    m(seq)
t_launch = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print("per sequence: host %.2f ms, wall %.2f ms" % (1e3 * t_launch / N, 1e3 * t_all / N))
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
    print("  %-22s %8.3f ms per sequence" % (k, 1e3 * v / N))
