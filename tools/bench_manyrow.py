#!/usr/bin/env python
"""The many-row lean GEMMs (csrc/gemm_sm.hip bm_kernel) at the whole-sequence-encoder shapes of a 512 x 512 sequence (16 frames x 1024
tokens) and of the 224 x 224 headline (10 x 196), launch by launch from one hipGraph over rotating weight copies.
  python tools/bench_manyrow.py [--rows 16384,1960]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from spann3r_amd import ops
from spann3r_amd.engine import _rope_tables

ap = argparse.ArgumentParser()
ap.add_argument("--rows", default="16384,8192,1960")
a = ap.parse_args()
dev, BF = "cuda", torch.bfloat16


def timeit(fns):
    for f in fns[:2]:
        f()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for f in fns:
            f()
    g.replay()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / len(fns))
    return best


print("%-10s %-20s %9s %9s" % ("op", "M x N x K", "us", "TFLOP/s"))
for M in [int(r) for r in a.rows.split(",")]:
    for name, N, K in (("qkv", 3072, 1024), ("proj", 1024, 1024), ("fc1", 4096, 1024), ("fc2", 1024, 4096)):
        ncopy = 6
        Ws = [ops.PackedWeight((torch.randn(N, K, device=dev) * 0.05).to(BF)) for _ in range(ncopy)]
        A = ops.PackedAct.from_dense(torch.randn(M, K, device=dev).to(BF))
        bias = torch.randn(N, device=dev)
        if name == "qkv":
            P = 1024 if M % 1024 == 0 else 196
            B = M // P
            npad = (P + 63) // 64 * 64
            qkp = ops.PackedAct(B * npad, 2048, BF, dev)
            vtp = torch.zeros(B * 16 * npad * 64, device=dev, dtype=BF)
            nh = int(P ** 0.5)
            ys, xs = torch.meshgrid(torch.arange(nh), torch.arange(nh), indexing="ij")
            pos = torch.stack((ys.reshape(-1), xs.reshape(-1)), -1)[None].expand(B, -1, -1).reshape(-1, 2).to(torch.int32).to(dev).contiguous()
            cos, sin = _rope_tables(64, 100.0, dev)
            st = torch.randn(M, K // 32, 2, device=dev).abs() + 40.0
            sW = torch.randn(N, device=dev)
            fns = [(lambda W=W: ops.proj_rope_vt(A, W, bias, qkp.data, 0, vtp, npad, M=M, N=N, K=K, lda=K, rope_cols=2048, pos=pos, cos=cos, sin=sin,
                                                  tokens=P, heads=16, qkv_packed=True, ln=ops.LnFold(st, K, sW, 1e-6))) for W in Ws]
        elif name == "fc1":
            out = ops.PackedAct(M, N, BF, dev)
            fns = [(lambda W=W: ops.gemm(A, W, out, M=M, N=N, K=K, lda=K, ldc=N, bias=bias, act=ops.ACT_GELU)) for W in Ws]
        else:
            out, res = torch.empty(M, N, device=dev), torch.randn(M, N, device=dev)
            so, c2 = torch.empty(M, N // 32, 2, device=dev), ops.PackedAct(M, N, BF, dev)
            fns = [(lambda W=W: ops.gemm(A, W, out, M=M, N=N, K=K, lda=K, ldc=N, bias=bias, res1=res, ldr1=N, stats_out=so, c2=c2)) for W in Ws]
        us = timeit(fns * 4)
        print("%-10s %-20s %9.2f %9.1f" % (name, "%d x %d x %d" % (M, N, K), us, 2.0 * M * N * K / us / 1e6), flush=True)
