#!/usr/bin/env python
"""Per-kernel timing of one ViT block as the engine launches it (bf16, M = 196): encoder (C=1024, 16 heads) and
decoder (C=768, 12 heads) geometry.  Needs an MI355X.  python tools/bench_block.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from spann3r_amd import Spann3R, FULL, ops
from spann3r_amd.weights import synth_state_dict
from tools.timing import timeit

m = Spann3R(dus3r_name=None, cfg=FULL, init_weights=False); m.load_state_dict(synth_state_dict(0, FULL)); m = m.cuda().eval()
m.set_precision(sys.argv[1] if len(sys.argv) > 1 else "bf16")
eng = m.engine
R, P, B = 196, 196, 1
pos = eng.positions(1, 14, 14)[1]
for name, C, heads, pre in (("encoder", 1024, 16, "enc0."), ("decoder", 768, 12, "dec1_0.")):
    x = torch.randn(R, C, device="cuda")
    xpA, xpB = eng.wsp("b_xpA" + name, R, C), eng.wsp("b_xpB" + name, R, C)
    stA, stB = eng.stats("b_stA" + name, R, C), eng.stats("b_stB" + name, R, C)
    ao = eng.wsp("b_ao" + name, R, C)
    h = eng.wsp("b_h" + name, R, 4 * C)
    # a producer call initialises xpA / stA with sane statistics
    eng._update(ao, pre + "proj", R, C, C, x, x, xpA, stA)
    eng._update(ao, pre + "proj", R, C, C, x, x, xpB, stB)
    npad = 256
    qkp = eng.ws("qkp_b" + name, ops.packed_shape(B * npad, 2 * C, eng.wdt), eng.wdt, zero=True)
    vtp = eng.ws("vtp_b" + name, (B * heads * npad * 64,), eng.wdt, zero=True)
    w = eng.w
    t = {}
    if eng.packed_attn:
        t["qkv+rope+vt"] = timeit(lambda: ops.proj_rope_vt(xpA, w[pre + "qkv.w"], w[pre + "qkv.b"], qkp, 0, vtp, npad, M=R, N=3 * C, K=C, lda=C,
                                  rope_cols=2 * C, pos=pos, cos=eng.cos, sin=eng.sin, tokens=P, heads=heads, qkv_packed=True,
                                  ln=ops.LnFold(stA, C, w[pre + "qkv.s"], 1e-6)))
        t["attention"] = timeit(lambda: ops.attention_packed(qkp, 2 * C, 0, npad, qkp, 2 * C, C, npad, vtp, ao, C, B=B, heads=heads, Nq=P, Nk=P, scale=0.125))
    else:
        t["attn_core"] = timeit(lambda: eng._attn_core(xpA, stA, R, B, P, C, heads, pre, pos, ao, tag="b" + name)) 
    t["proj+res+stats"] = timeit(lambda: eng._update(ao, pre + "proj", R, C, C, x, x, xpB, stB))
    t["fc1+gelu"] = timeit(lambda: eng._mlp_fc1(xpB, stB, R, C, pre, h))
    t["fc2+res+stats"] = timeit(lambda: eng._update(h, pre + "fc2", R, C, 4 * C, x, x, xpA, stA))
    print(name, " ".join("%s %.2f" % kv for kv in t.items()), "| block %.2f us" % sum(t.values()))
