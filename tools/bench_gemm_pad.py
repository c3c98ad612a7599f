#!/usr/bin/env python
"""Does a power-of-two row stride hurt the direct-to-register operand loads? (needs an MI355X)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from spann3r_amd import ops  # noqa: E402
from tools.timing import timeit  # noqa: E402

dev = "cuda"
print("%-20s %-5s %-6s %-6s %8s %8s" % ("shape", "tile", "padW", "padA", "us", "TF/s"))
for (M, N, K) in [(196, 4096, 1024), (196, 3072, 1024), (196, 1024, 1024), (196, 1024, 4096)]:
    for tile in (0, 3):
        for padw, pada in ((0, 0), (64, 0), (64, 64), (8, 8), (32, 32), (128, 128)):
            A = torch.randn(M, K + pada, device=dev).to(torch.bfloat16)
            W = (torch.randn(N, K + padw, device=dev) * 0.05).to(torch.bfloat16)
            out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
            us = timeit(lambda: ops.gemm(A, W, out, M=M, N=N, K=K, lda=K + pada, ldc=N, ldw=K + padw, tile=tile))
            print("%-20s %-5d %-6d %-6d %8.2f %8.1f" % ("%dx%dx%d" % (M, N, K), tile, padw, pada, us, 2.0 * M * N * K / us / 1e6))
