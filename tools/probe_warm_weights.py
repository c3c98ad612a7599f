#!/usr/bin/env python
"""Upper bound of what a weight prefetcher could buy the 224 x 224 step: the headline workload with every decoder layer (and every
value-encoder layer) ALIASED to layer 0's packed weights, so each launch finds its weights in the L2 / MALL instead of cold in HBM.
The outputs are meaningless; only the time is read.  python tools/probe_warm_weights.py [--alias dec,val,none]"""
import argparse, os, re, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

ap = argparse.ArgumentParser()
ap.add_argument("--alias", default="none,dec,dec+val,enc,dec+val+enc")
a = ap.parse_args()
dev = torch.device("cuda:0")
from spann3r_amd.runner import make_sequence
seqs = [make_sequence(300 + i, 10, 224, 224, device=dev) for i in range(4)]
for mode in a.alias.split(","):
    model, _ = bench.build_model("bf16", dev)
    model(seqs[0])                      # builds the engine
    w = model._engine.w
    n = 0
    if "dec" in mode:
        for k in list(w):
            if k.startswith("decg_") and not k.startswith("decg_0."):
                w[k] = w["decg_0." + k.split(".", 1)[1]]
                n += 1
    if "enc" in mode:
        for k in list(w):
            if re.match(r"enc\d+\.", k) and not k.startswith("enc0."):
                w[k] = w["enc0." + k.split(".", 1)[1]]
                n += 1
    if "val" in mode:
        for k in list(w):
            if k.startswith("val") and k[3:4].isdigit() and not k.startswith("val0."):
                w[k] = w["val0." + k.split(".", 1)[1]]
                n += 1
    model._runners = {}                  # drop graphs captured with the old pointers
    fr, s = bench.time_sequences(model, seqs, 12, 4)
    print("alias=%-8s (%3d tensors re-pointed): %.1f frames/s  %.3f ms per 10-frame sequence" % (mode, n, fr / s, 1e3 * s / 12), flush=True)
    del model
    torch.cuda.empty_cache()
