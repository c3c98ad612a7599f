#!/usr/bin/env python
"""Which multi-stream hipGraph capture patterns survive on this ROCm?  (needs an MI355X; each variant in a subprocess)"""
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

VARIANT = sys.argv[1] if len(sys.argv) > 1 else None
if VARIANT is None:
    for v in ("forkjoin", "events_keep", "events_drop", "layers_forkjoin"):
        r = subprocess.run([sys.executable, __file__, v], capture_output=True, text=True)
        print(v, "->", "OK" if r.returncode == 0 else "FAIL rc=%d" % r.returncode, r.stdout.strip()[-100:], r.stderr.strip()[-200:].replace("\n", " | "))
    sys.exit(0)

import torch  # noqa: E402
from spann3r_amd import ops  # noqa: E402

dev = "cuda"
a = torch.zeros(1 << 16, device=dev)
b = torch.zeros(1 << 16, device=dev)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
keep = []


def body():
    main = torch.cuda.current_stream()
    if VARIANT == "forkjoin":
        s1.wait_stream(main); s2.wait_stream(main)
        with torch.cuda.stream(s1):
            ops.fill(a, 1.0); ops.fill(a, 2.0)
        with torch.cuda.stream(s2):
            ops.fill(b, 3.0); ops.fill(b, 4.0)
        main.wait_stream(s1); main.wait_stream(s2)
    elif VARIANT in ("events_keep", "events_drop"):
        s1.wait_stream(main); s2.wait_stream(main)
        e1 = e2 = None
        for i in range(4):
            with torch.cuda.stream(s1):
                if e2 is not None:
                    s1.wait_event(e2)
                ops.fill(a, float(i))
                n1 = s1.record_event()
            with torch.cuda.stream(s2):
                if e1 is not None:
                    s2.wait_event(e1)
                ops.fill(b, float(i) + 10)
                n2 = s2.record_event()
            if VARIANT == "events_keep":
                keep.extend([n1, n2])
            e1, e2 = n1, n2
        main.wait_stream(s1); main.wait_stream(s2)
    elif VARIANT == "layers_forkjoin":
        for i in range(4):
            s1.wait_stream(main); s2.wait_stream(main)
            with torch.cuda.stream(s1):
                ops.fill(a, float(i))
            with torch.cuda.stream(s2):
                ops.fill(b, float(i) + 10)
            main.wait_stream(s1); main.wait_stream(s2)
    ops.fill(a[:16], 7.0)


body()
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    body()
g.replay()
torch.cuda.synchronize()
print("a0=%g a_last=%g b0=%g" % (float(a[0]), float(a[-1]), float(b[0])))
