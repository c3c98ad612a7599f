#!/usr/bin/env python
"""DPT-head 3x3 convolutions at 224x224 input geometry: the LDS-tiled kernel vs sp3_gemm's conv loader (bf16 weights).
Needs an MI355X.  python tools/bench_conv.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from spann3r_amd import ops
from tools.timing import timeit
dev = "cuda"
print("%-26s %10s %10s %10s %10s" % ("B,H,W,Cin,Cout", "tile us", "TFLOP/s", "gemm us", "TFLOP/s"))
for (B, H, W, Cin, Cout) in [(1, 56, 56, 256, 256), (1, 28, 28, 256, 256), (1, 14, 14, 256, 256), (1, 112, 112, 256, 128),
                             (1, 224, 224, 128, 128), (1, 14, 14, 384, 256), (1, 7, 7, 768, 256)]:
    for in_dt in (torch.float32, torch.bfloat16):
        x = torch.randn(B, H, W, Cin, device=dev).to(in_dt)
        w = (torch.randn(Cout, 9 * Cin, device=dev) * 0.02).to(torch.bfloat16)
        wp = ops.PackedWeight(w)
        out = torch.empty(B, H, W, Cout, device=dev, dtype=in_dt)
        bias = torch.randn(Cout, device=dev)
        fl = 2.0 * B * H * W * Cout * 9 * Cin
        t_new = timeit(lambda: ops.conv3x3(x, wp, out, B=B, H=H, W_=W, Cin=Cin, Cout=Cout, bias=bias, relu_in=True, force_tile_kernel=True), 20)
        t_old = min(timeit(lambda: ops.conv3x3(x, wp, out, B=B, H=H, W_=W, Cin=Cin, Cout=Cout, bias=bias, relu_in=True, tile=t), 20)
                    for t in (0, 1))
        print("%-26s %10.2f %10.1f %10.2f %10.1f  (%s maps)" % ("%d,%d,%d,%d,%d" % (B, H, W, Cin, Cout), t_new, fl / t_new / 1e6,
                                                       t_old, fl / t_old / 1e6, "bf16" if in_dt == torch.bfloat16 else "fp32"))
