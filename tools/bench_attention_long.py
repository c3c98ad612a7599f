#!/usr/bin/env python
"""us per launch of sp3_attention_packed at long sequences (config 3: 1024 tokens per frame)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from spann3r_amd import ops  # noqa: E402

dev = "cuda"
for B, heads, N in ((1, 16, 1024), (2, 12, 1024)):
    C = heads * 64
    qp = ops.PackedAct.from_dense(torch.randn(B * N, C, device=dev).to(torch.bfloat16))
    kp = ops.PackedAct.from_dense(torch.randn(B * N, C, device=dev).to(torch.bfloat16))
    vp = torch.randn(B * heads * (N // 32) * 4 * 64 * 8, device=dev).to(torch.bfloat16)
    out = ops.PackedAct(B * N, C, torch.bfloat16, dev)
    fn = lambda: ops.attention_packed(qp, C, 0, N, kp, C, 0, N, vp, out, C, B=B, heads=heads, Nq=N, Nk=N, scale=0.125)
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200):
        fn()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 200
    print("attention_packed B %d heads %d N %d: %.1f us = %.0f TFLOP/s" % (B, heads, N, us, 4.0 * B * heads * N * N * 64 / us / 1e6))
