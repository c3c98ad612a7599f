#!/usr/bin/env python
"""Timeline of the first workgroups of a role-loop GEMM launch (tiles 13-15; sp3_gemm_desc.trace): shader-clock stamps at
entry, per 4 k-blocks for the loader and for consumer wave 0, loop end, epilogue end.  Weights are HBM-cold (a different
copy per launch).   python tools/trace_gemm.py [tile]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from spann3r_amd import ops  # noqa: E402

tile = int(sys.argv[1]) if len(sys.argv) > 1 else 13
dev, DT = "cuda", torch.bfloat16
SHAPES = (("val proj", 196, 1024, 1024), ("val fc1", 196, 4096, 1024), ("dec fc1", 196, 3072, 768), ("val fc2", 196, 1024, 4096))
if tile >= 20:                       # the pipelined many-row tiles: fixed cost (K = 64) and the encoder's shapes
    SHAPES = (("k64", 1960, 4096, 64), ("enc fc1", 1960, 4096, 1024), ("enc fc2", 1960, 1024, 4096), ("enc proj", 1960, 1024, 1024))
ACT = ops.ACT_NONE if (len(sys.argv) > 2 and sys.argv[2] == "noact") else ops.ACT_GELU
for name, M, N, K in SHAPES:
    Ws = [ops.PackedWeight((torch.randn(N, K, device=dev) * 0.05).to(DT)) for _ in range(max(4, 400 * (1 << 20) // (N * K * 2)))]
    A = ops.PackedAct.from_dense(torch.randn(M, K, device=dev).to(DT))
    out = ops.PackedAct(M, N, DT, dev)
    bias = torch.randn(N, device=dev)
    trace = torch.zeros(8 * 64, dtype=torch.int64, device=dev)
    for W in Ws:                       # the traced launch is the last one: its weights were touched longest ago
        ops.gemm(A, W, out, M=M, N=N, K=K, lda=K, ldc=N, bias=bias, act=ACT, tile=tile)
    torch.cuda.synchronize()
    ops.gemm(A, Ws[0], out, M=M, N=N, K=K, lda=K, ldc=N, bias=bias, act=ACT, tile=tile, trace=trace)
    torch.cuda.synchronize()
    t = trace.cpu().view(8, 64)
    print("== %s %dx%dx%d tile %d" % (name, M, N, K, tile))
    for wg in range(3):
        r = t[wg]
        t0 = int(r[0])
        clk_per_us = (int(r[2]) - t0) / max((int(r[4]) - int(r[3])) * 0.01, 1e-9)
        nkb = K // 64
        ld = [int(r[8 + i]) - t0 for i in range((nkb + 3) // 4)]
        cs = [int(r[32 + i]) - t0 for i in range((nkb + 3) // 4)]
        if tile < 13 or tile >= 20:
            extra = " | stage 0 landed %5d" % (int(r[6]) - t0) if tile >= 20 else ""
            print("wg %d: setup %5d%s | loop done %6d | slab+sync %6d | end %6d clk (%.2f us)" % (wg, int(r[5]) - t0, extra, int(r[1]) - t0, int(r[48]) - t0, int(r[2]) - t0, (int(r[4]) - int(r[3])) * 0.01))
            continue
        print("wg %d: setup %5d  W issued %5d  A0 landed %5d | slab+sync %6d |" % (wg, int(r[5]) - t0, int(r[7]) - t0, int(r[6]) - t0, int(r[48]) - t0), end=" ")
        print("wg %d: loop done %6d  end %6d clk (%.0f clk/us, %.2f us) | loader barrier@kb 0,4,..: %s | consumer done@kb 0,4,..: %s"
              % (wg, int(r[1]) - t0, int(r[2]) - t0, clk_per_us, (int(r[4]) - int(r[3])) * 0.01, ld, cs))
