#!/usr/bin/env python
"""Micro-benchmark of sp3_gemm on the hot path's shapes (needs an MI355X).
python tools/bench_gemm.py [--adt bf16|f32] [--wdt bf16|f32]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from spann3r_amd import ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--adt", default="bf16")
ap.add_argument("--wdt", default="bf16")
ap.add_argument("--iters", type=int, default=50)
ap.add_argument("--packed", type=int, default=1)
ap.add_argument("--M", type=int, default=0, help="only the encoder shapes at this M")
args = ap.parse_args()
DT = {"bf16": torch.bfloat16, "f32": torch.float32}
dev = "cuda"
shapes = [  # M, N, K, splitk list
    (196, 3072, 1024, [0]), (196, 4096, 1024, [0]), (196, 1024, 4096, [0, 1, 2, 4]), (196, 1024, 1024, [0, 1, 2]),
    (196, 2304, 768, [0]), (196, 768, 3072, [0, 2, 4]), (196, 768, 768, [0, 1]), (392, 4096, 1024, [0]),
]


if args.M:
    shapes = [(args.M, 3072, 1024, [0]), (args.M, 4096, 1024, [0]), (args.M, 1024, 4096, [0]), (args.M, 1024, 1024, [0])]
from tools.timing import timeit  # noqa: E402


print("%-22s %-6s %-5s %9s %9s %10s" % ("shape MxNxK", "splitk", "tile", "us", "TFLOP/s", "W GB/s"))
for M, N, K, sks in shapes:
    A = torch.randn(M, K, device=dev).to(DT[args.adt])
    W = (torch.randn(N, K, device=dev) * 0.05).to(DT[args.wdt])
    Wbytes = W.numel() * W.element_size()
    if args.packed:
        W = ops.PackedWeight(W)
    if args.packed > 1:
        A = ops.PackedAct.from_dense(A)
    bias = torch.randn(N, device=dev)
    for sk in sks:
        for tile in ((0, 3, 1, 2, 5, 6) if (args.packed == 2 and args.adt == 'bf16' and K % 64 == 0) else (0, 3, 1, 2)):
            if sk == 0:
                out = torch.empty(M, N, device=dev, dtype=DT[args.adt])
                fn = lambda: ops.gemm(A, W, out, M=M, N=N, K=K, lda=K, ldc=N, bias=bias, act=ops.ACT_GELU, tile=tile)
            else:
                out = torch.empty(sk, M, N, device=dev)
                fn = lambda: ops.gemm(A, W, out, M=M, N=N, K=K, lda=K, ldc=N, splitk=sk, tile=tile)
            us = timeit(fn, args.iters)
            print("%-22s %-6d %-5d %9.2f %9.1f %10.0f" % ("%dx%dx%d" % (M, N, K), sk, tile, us, 2.0 * M * N * K / us / 1e6,
                                                      Wbytes / us / 1e3))
# LayerNorm / reduce_ln / attention for reference
x = torch.randn(196, 1024, device=dev)
g_, b_ = torch.ones(1024, device=dev), torch.zeros(1024, device=dev)
o = torch.empty(196, 1024, device=dev, dtype=torch.bfloat16)
print("layernorm 196x1024: %.2f us" % timeit(lambda: ops.layernorm(x, g_, b_, 1e-6, o, rows=196, C_=1024), args.iters))
part = torch.randn(4, 196, 1024, device=dev)
print("reduce_ln S=4 196x1024: %.2f us" % timeit(lambda: ops.reduce_ln(part, 4, 196, 1024, bias=b_, res=x, x_out=x, ln1=(g_, b_), out1=o), args.iters))
qk = torch.randn(196, 2048, device=dev).to(torch.bfloat16)
vt = torch.randn(16 * 64, 256, device=dev).to(torch.bfloat16)
ao = torch.empty(196, 1024, device=dev, dtype=torch.bfloat16)
print("attention 16h 196x196: %.2f us" % timeit(lambda: ops.attention(qk, 196 * 2048, 2048, qk[:, 1024:], 196 * 2048, 2048, vt, 256, ao, 1024, B=1, heads=16, Nq=196, Nk=196, scale=0.125), args.iters))
fill = torch.empty(1024, device=dev)
print("fill 1024 (launch floor): %.2f us" % timeit(lambda: ops.fill(fill, 1.0), args.iters))
