#!/usr/bin/env python
"""attention kernels, isolated (needs an MI355X)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from spann3r_amd import ops
from tools.timing import timeit
dev = "cuda"
for heads, N in ((16, 196), (12, 196), (16, 1024)):
    C = heads * 64
    npad = (N + 63) // 64 * 64
    qk = torch.randn(N, 2 * C, device=dev).to(torch.bfloat16)
    vt = torch.randn(heads * 64, npad, device=dev).to(torch.bfloat16)
    ao = torch.empty(N, C, device=dev, dtype=torch.bfloat16)
    t1 = timeit(lambda: ops.attention(qk, N * 2 * C, 2 * C, qk[:, C:], N * 2 * C, 2 * C, vt, npad, ao, C, B=1, heads=heads, Nq=N, Nk=N, scale=0.125))
    qkp = ops.PackedAct(npad, 2 * C, torch.bfloat16, dev)
    qkp.data.normal_()
    vtp = torch.randn(heads * npad * 64, device=dev).to(torch.bfloat16)
    aop = ops.PackedAct(N, C, torch.bfloat16, dev)
    t2 = timeit(lambda: ops.attention_packed(qkp, 2 * C, 0, npad, qkp, 2 * C, C, npad, vtp, aop, C, B=1, heads=heads, Nq=N, Nk=N, scale=0.125))
    fl = 4.0 * heads * N * N * 64
    print("heads %d N %d: row-major %.2f us (%.1f TF)  packed+prefetch %.2f us (%.1f TF)" % (heads, N, t1, fl / t1 / 1e6, t2, fl / t2 / 1e6))
