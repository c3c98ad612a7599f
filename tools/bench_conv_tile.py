"""Time sp3_conv3x3_tile with the 8 x 8 and the 8 x 16 pixel tile on the DPT-head shapes (hipGraph of 20 launches, HIP events):
python tools/bench_conv_tile.py  ->  one line per shape, us per launch and TFLOP/s for both tiles."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spann3r_amd import ops  # noqa: E402

SHAPES = [  # B, H, W, Cin, Cout (8x16n32 = the wide tile with 32 output channels per workgroup)          (config 2: 224 input -> 56 / 112 / 224 maps; config 3: 512 input -> 128 / 256 / 512)
    (1, 56, 56, 256, 256), (1, 48, 64, 256, 256), (2, 56, 56, 256, 256), (9, 56, 56, 256, 256), (1, 112, 112, 256, 256), (1, 112, 112, 256, 128), (9, 112, 112, 256, 256),
    (1, 224, 224, 128, 128), (9, 224, 224, 128, 128), (1, 128, 128, 256, 256), (1, 256, 256, 256, 256), (1, 256, 256, 256, 128),
    (1, 512, 512, 128, 128)]


def main():
    dev = "cuda"
    print("%-28s %10s %10s %10s %10s %10s %10s  %s" % ("B,H,W,Cin,Cout", "8x8 us", "TFLOP/s", "8x16 us", "TFLOP/s", "8x16n32 us", "TFLOP/s", "wide workgroups"))
    for B, H, W, Cin, Cout in SHAPES:
        x = torch.randn(B, H, W, Cin, device=dev).to(torch.bfloat16)
        wp = ops.PackedWeight((torch.randn(Cout, 9 * Cin, device=dev) * 0.02).to(torch.bfloat16))
        bias = torch.randn(Cout, device=dev)
        out = torch.empty(B, H, W, Cout, device=dev, dtype=torch.bfloat16)
        res = {}
        for px in ("8x8", "8x16", "8x16n32"):
            run = lambda: ops.conv3x3(x, wp, out, B=B, H=H, W_=W, Cin=Cin, Cout=Cout, bias=bias, relu_in=True, tile_px=px)
            run()
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                for _ in range(20):
                    run()
            g.replay()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            best = 1e9
            for _ in range(5):
                e0.record()
                g.replay()
                e1.record()
                torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) * 1000 / 20)
            res[px] = best
        fl = 2.0 * B * H * W * Cout * 9 * Cin
        wgs = ((W + 15) // 16) * ((H + 7) // 8) * B * (Cout // 64)
        print("%-28s %10.2f %10.1f %10.2f %10.1f %10.2f %10.1f  %d" % ("%d,%d,%d,%d,%d" % (B, H, W, Cin, Cout), res["8x8"], fl / res["8x8"] / 1e6,
                                                                     res["8x16"], fl / res["8x16"] / 1e6, res["8x16n32"], fl / res["8x16n32"] / 1e6, wgs), flush=True)


if __name__ == "__main__":
    main()
