#!/usr/bin/env python
"""Per-launch times of the training attention path at the decoder's geometry (needs an MI355X):
head shuffle (split + RoPE + transposes), flash forward, dq, dk / dv, the in-place inverse-RoPE shuffle, and for comparison the
materialised path (_MHA: grouped GEMMs + softmax kernels).  Every variant is captured into one hipGraph of `reps` forward + backward
passes and timed with events around the replay.
   python tools/bench_train_attention.py [--B 4] [--N 196] [--H 12] [--precision bf16] [--cross]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from spann3r_amd import train as T  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--B", type=int, default=4)
ap.add_argument("--N", type=int, default=196)
ap.add_argument("--H", type=int, default=12)
ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32"])
ap.add_argument("--cross", action="store_true")
ap.add_argument("--reps", type=int, default=20)
args = ap.parse_args()
dev = "cuda"
B, N, H = args.B, args.N, args.H
C = 64 * H
nh = int(N ** 0.5)
assert nh * nh == N, "N must be a square token grid"
ys, xs = torch.meshgrid(torch.arange(nh, device=dev), torch.arange(nh, device=dev), indexing="ij")
pos = torch.stack((ys.reshape(-1), xs.reshape(-1)), -1)[None].expand(B, -1, -1).contiguous()
T.set_precision(args.precision)


def one(flash):
    T.FLASH_ATTENTION = flash
    g = torch.Generator(device=dev).manual_seed(0)
    if args.cross:
        ins = [torch.randn(B, N, C, device=dev, generator=g).requires_grad_(True) for _ in range(3)]
        fn = lambda: T._mha(ins[0], ins[1], ins[2], pos, pos, H, 0.125, 100.0)
    else:
        ins = [torch.randn(B, N, 3 * C, device=dev, generator=g).requires_grad_(True)]
        fn = lambda: T._mha(ins[0], None, None, pos, pos, H, 0.125, 100.0)
    d = torch.randn(B, N, C, device=dev, generator=g)

    def step():
        for t in ins:
            t.grad = None
        fn().backward(d)
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(args.reps):
            step()
    gr.replay()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        gr.replay()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / args.reps)
    return best


try:
    for name, flash in (("flash (shuffle + fwd | shuffle + dq + dk/dv + shuffle)", True), ("materialised (shuffles around GEMMs + softmax)", False)):
        print("%-60s %8.1f us per forward + backward  (B %d, N %d, H %d, %s, %s)" % (name, one(flash), B, N, H, args.precision, "cross" if args.cross else "self"))
    print("per-kernel split: rocprofv3 --kernel-trace --stats -- python tools/bench_train_attention.py ...")
finally:
    T.FLASH_ATTENTION = True
    T.set_precision("fp32")
