#!/usr/bin/env python
"""Tile sweep of sp3_gemm on the per-frame step's 196-row weight-streaming shapes (needs an MI355X).
Every launch of a timed graph reads a DIFFERENT copy of the weights (the copies exceed the 256 MB Infinity Cache), as in
the model, where one step streams ~620 MB of weights.   python tools/bench_gemm2.py [--M 196] [--big]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from spann3r_amd import ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--M", type=int, default=196)
ap.add_argument("--big", action="store_true", help="the many-row shapes (whole-sequence encoder, 512x512 steps)")
ap.add_argument("--dtype", default="bf16")
ap.add_argument("--only", default="", help="comma list of op names")
ap.add_argument("--tiles", default="")
ap.add_argument("--mb", type=int, default=320, help="MB of distinct weight copies to cycle through")
ap.add_argument("--train", action="store_true", help="the training step's Linear shapes at batch 4 (784 rows; dW: contraction over the rows), fp32 row-major output")
ap.add_argument("--mode", default="", help="product mode of fp32 operands (f16x3, f32x3, f32x6): with --dtype fp32")
ap.add_argument("--halves", action="store_true", help="f16x3: weights as their (h, l) fp16 planes (PackedWeight(halves=True))")
ap.add_argument("--ksweep", action="store_true", help="one many-row shape at several K: fixed cost vs cost per k-block")
args = ap.parse_args()
dev = "cuda"
if args.mode:
    ops.set_product_mode(args.mode)
DT = torch.bfloat16 if args.dtype == "bf16" else torch.float32
ES = 2 if args.dtype == "bf16" else 4


def timeit_rot(fns):
    """us per launch over one hipGraph that runs every fn once, back to back"""
    for f in fns[:3]:
        f()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for f in fns:
            f()
    g.replay()
    torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / len(fns))
    return min(ts)


if args.big:
    # (name, M, N, K, batch)
    shapes = [("enc qkv", 1960, 3072, 1024, 1), ("enc fc1", 1960, 4096, 1024, 1), ("enc fc2", 1960, 1024, 4096, 1),
              ("enc proj", 1960, 1024, 1024, 1), ("c3 dec qkv", 1024, 2304, 768, 2), ("c3 dec fc1", 1024, 3072, 768, 2),
              ("c3 dec fc2", 1024, 768, 3072, 2), ("c3 dec proj", 1024, 768, 768, 2), ("c3 val fc1", 1024, 4096, 1024, 1),
              ("c3 val proj", 1024, 1024, 1024, 1), ("c3 val fc2", 1024, 1024, 4096, 1), ("c3 key 2", 1024, 1024, 1792, 2)]
    shapes += [("enc16 qkv", 16384, 3072, 1024, 1), ("enc16 fc2", 16384, 1024, 4096, 1), ("read S", 1024, 50176, 1024, 1),
               ("read PV", 1024, 1024, 50176, 1), ("train fc1", 784, 4096, 1024, 1), ("train dec", 784, 768, 3072, 2),
               ("sq4k", 4096, 4096, 4096, 1), ("sq8k", 8192, 4096, 4096, 1), ("enc16 fc1", 16384, 4096, 1024, 1)]
    tiles = [1, 5, 6, 20, 21, 22, 23]
else:
    M = args.M
    shapes = [("dec qkv", M, 2304, 768, 2), ("dec proj", M, 768, 768, 2), ("dec ckv", M, 1536, 768, 2), ("dec fc1", M, 3072, 768, 2),
              ("dec fc2", M, 768, 3072, 2), ("val qkv", M, 3072, 1024, 1), ("val proj", M, 1024, 1024, 1), ("val fc1", M, 4096, 1024, 1),
              ("val fc2", M, 1024, 4096, 1), ("key 0", M, 1792, 1792, 2), ("key 2", M, 1024, 1792, 2)]
    tiles = [0, 13, 14, 16, 17]

if args.train:
    shapes = []
    for C, Hd, tag in ((768, 3072, "dec"), (1024, 4096, "val")):
        for nm, N, K in (("proj", C, C), ("qkv", 3 * C, C), ("fc1", Hd, C), ("fc2", C, Hd)):
            shapes.append(("%s %s" % (tag, nm), 784, N, K, 1))            # forward; dX is the same launch with N and K swapped
            shapes.append(("%s %s dW" % (tag, nm), N, K, 832, 1))         # dW = dY^T . X, contraction over the 784 rows (padded to 64)
    tiles = [0, 1, 21, 22, 23, 24, 25]
if args.ksweep:
    shapes = [("k%d" % K, 1960, 4096, K, 1) for K in (128, 512, 1024, 2048, 4096)] + [("n1k k%d" % K, 1960, 1024, K, 1) for K in (128, 1024, 4096)]
    tiles = [5, 20, 21, 22]
print("%-12s %-18s %-5s %-3s %8s %9s %9s" % ("op", "MxNxK x batch", "tile", "sk", "us", "TFLOP/s", "W GB/s"))
if args.tiles:
    tiles = [int(t) for t in args.tiles.split(',')]
for name, M, N, K, G in shapes:
    if args.only and name not in args.only.split(','):
        continue
    wbytes = G * N * K * ES
    ncopy = max(1, min(64, args.mb * (1 << 20) // wbytes))
    Ws = []
    for c in range(ncopy):
        ws = [ops.PackedWeight((torch.randn(N, K, device=dev) * 0.05).to(DT), halves=args.halves) for _ in range(G)]
        Ws.append(ops.PackedWeightGroup(ws) if G > 1 else ws[0])
    if G > 1:
        A = ops.PackedAct.group(G, M, K, DT, dev)
        A.data.copy_(torch.randn(A.data.shape, device=dev).to(DT))
        out = ops.PackedAct.group(G, M, N, DT, dev)
    else:
        A = ops.PackedAct.from_dense(torch.randn(M, K, device=dev).to(DT))
        out = torch.empty(M, N, device=dev) if args.train else ops.PackedAct(M, N, DT, dev)
    bias = torch.randn(G, N, device=dev)
    ACT = ops.ACT_GELU if (("fc1" in name or name == "key 0") and not args.train) else ops.ACT_NONE
    part = torch.empty(8 * G * M * N, device=dev)
    for tile in tiles:
        BN = {0: 32, 9: 32, 5: 128, 20: 128, 21: 128, 2: 128, 24: 32, 25: 32}.get(tile, 64)
        BM = {0: 32, 4: 32, 18: 16, 20: 256, 21: 128, 22: 128, 23: 64, 24: 64, 25: 32, 8: 208, 12: 208, 15: 208, 5: 128, 6: 128, 1: 64, 2: 64}.get(tile, 112)
        wgs = ((M + BM - 1) // BM) * ((N + BN - 1) // BN) * G
        sks = [0]
        if not args.big and tile != 0 and G == 1:
            sks += [s for s in (2,) if wgs * s <= 300 and K // 64 // s >= 4]
        for sk in sks:
            def mk(W):
                kw = dict(M=M, N=N, K=K, lda=K, ldc=N, tile=tile)
                if sk:
                    return lambda: ops.gemm(A, W, part, splitk=sk, **kw)
                if G > 1:
                    return lambda: ops.gemm(A, W, out, bias=bias, act=ACT, batch=G, strideA=A.stride, strideW=W.stride,
                                            strideC=out.stride, sb={"bias": N * 4}, **kw)
                return lambda: ops.gemm(A, W, out, bias=bias, act=ACT, **kw)
            try:
                fns = [mk(W) for W in Ws]
                fns = fns * max(1, -(-48 // len(fns)))          # at least 48 launches per graph (replay overhead)
                us = timeit_rot(fns)
            except RuntimeError as e:
                print("%-12s %-18s %-5d %-3d  failed: %s" % (name, "%dx%dx%d x%d" % (M, N, K, G), tile, sk, str(e)[:60]))
                continue
            print("%-12s %-18s %-5d %-3d %8.2f %9.1f %9.0f   (%d wgs, %d weight copies)" % (
                name, "%dx%dx%d x%d" % (M, N, K, G), tile, sk, us, 2.0 * G * M * N * K / us / 1e6, wbytes / us / 1e3, wgs * max(sk, 1), ncopy))
