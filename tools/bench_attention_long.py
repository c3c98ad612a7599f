#!/usr/bin/env python
"""us per launch of sp3_attention_packed (config 3: 1024 tokens per frame; the 196-token launches of the headline), rotating operand
copies so that no launch finds its K / V^T in the L2 a previous one left them in.  (Round 5 A/B'd the workgroup -> (head, batch) map
with it, profiles/r05_xcd_locality_ab.txt; the XCD map is the only one since round 6.)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from spann3r_amd import ops  # noqa: E402

dev = "cuda"
for B, heads, N in ((2, 16, 1024), (2, 12, 1024), (16, 16, 1024), (2, 12, 196), (10, 16, 196), (2, 16, 196)):
    C = heads * 64
    Np = (N + 63) // 64 * 64
    copies = []
    for _ in range(6):
        qp = ops.PackedAct.from_dense(torch.randn(B * Np, C, device=dev).to(torch.bfloat16))
        kp = ops.PackedAct.from_dense(torch.randn(B * Np, C, device=dev).to(torch.bfloat16))
        vp = torch.randn(B * heads * (Np // 32) * 4 * 64 * 8, device=dev).to(torch.bfloat16)
        copies.append((qp, kp, vp))
    out = ops.PackedAct(B * Np, C, torch.bfloat16, dev)
    fn = lambda c: ops.attention_packed(c[0], C, 0, Np, c[1], C, 0, Np, c[2], out, C, B=B, heads=heads, Nq=N, Nk=N, scale=0.125)
    for c in copies:
        fn(c)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(5):
            for c in copies:
                fn(c)
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / (10 * 5 * len(copies))
    print("attention_packed B %2d heads %2d N %4d: %6.2f us = %4.0f TFLOP/s" % (B, heads, N, us, 4.0 * B * heads * N * N * 64 / us / 1e6))
