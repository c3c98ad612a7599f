#!/usr/bin/env python
"""sp3_gemm time vs K (slope = per-k-block cost, intercept = fixed cost).  Needs an MI355X."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from spann3r_amd import ops  # noqa: E402
from tools.timing import timeit  # noqa: E402

dev = "cuda"
print("%-8s %-6s %-5s %-8s %8s %8s" % ("N", "K", "tile", "epi", "us", "TB/s(L2)"))
for N in (1024, 4096):
    for tile in (0, 3):
        for K in (256, 512, 1024, 2048, 4096, 8192):
            for epi in ("partial", "plain"):
                M = 196
                A = ops.PackedAct.from_dense(torch.randn(M, K, device=dev).to(torch.bfloat16))
                W = ops.PackedWeight((torch.randn(N, K, device=dev) * 0.05).to(torch.bfloat16))
                if epi == "partial":
                    out = torch.empty(1, M, N, device=dev)
                    fn = lambda: ops.gemm(A, W, out, M=M, N=N, K=K, lda=K, ldc=N, splitk=1, tile=tile)
                else:
                    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
                    fn = lambda: ops.gemm(A, W, out, M=M, N=N, K=K, lda=K, ldc=N, tile=tile)
                us = timeit(fn)
                bm = 32 if tile == 0 else 64
                traffic = ((M + bm - 1) // bm) * (N // bm) * 2 * bm * K * 2
                print("%-8d %-6d %-5d %-8s %8.2f %8.2f" % (N, K, tile, epi, us, traffic / us / 1e6))
