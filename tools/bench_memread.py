#!/usr/bin/env python
"""Per-launch timing of the spatial-memory read at a given bank length (both compositions), cold-ish operands
(each repetition works on its own copy of the bank so nothing is L2-resident from the previous one)."""
import argparse, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from spann3r_amd import ops

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tokens", type=int, default=1764)
    ap.add_argument("--rows", type=int, default=196)
    ap.add_argument("--copies", type=int, default=24)
    a = ap.parse_args()
    dev, wdt, C, P, M = "cuda", torch.bfloat16, 1024, a.rows, a.tokens
    cap = (M + 127) // 128 * 128
    Kp = cap
    nt = (M + 31) // 32
    g = torch.Generator().manual_seed(0)
    mk = lambda *s: torch.randn(*s, generator=g)
    banks = []
    for _ in range(a.copies):
        Kh = ops.PackedAct.from_dense((mk(cap, C) * 2).to(dev).to(wdt))
        Vt = ops.PackedAct.from_dense(mk(C, cap).to(dev).to(wdt))
        banks.append((Kh, Vt))
    q = mk(P, C).to(dev)
    qp = ops.PackedAct(P, C, wdt, dev); qs = torch.zeros(P, C // 32, 2, device=dev)
    ops.pack_stats(q, qp, qs, rows=P, C_=C)
    s_bank = torch.randn(cap, device=dev) * 0.01; b_bank = torch.zeros(cap, device=dev)
    S = torch.zeros(P, cap, device=dev); st = torch.zeros(P, nt, 2, device=dev); zk = torch.zeros(P, 4, device=dev)
    out = torch.zeros(P, C, device=dev); attn = torch.zeros(cap, device=dev)
    pk = torch.zeros(((P + 15) // 16 * 16) * Kp, device=dev, dtype=wdt)
    ln = ops.LnFold(qs, C, s_bank, 1e-5)
    part = torch.zeros(16 * P * C, device=dev)
    rowstat = torch.zeros(P * 4, device=dev)
    long_bank = M > 8192
    # the score-matrix-free composition (round 6; > 256 query rows): p~ + group statistics, merge, rescaling P.V stage, reduce, column sums
    rows_pad, ngc = (P + 255) // 256 * 256, cap // 64
    pstats, pscale = torch.zeros(ngc * rows_pad * 2, device=dev), torch.zeros(ngc * rows_pad, device=dev)
    pt = ops.PackedAct(P, cap, wdt, dev, data=pk) if P > 256 else None
    Sk = 8 if M <= 16384 else 16
    steps = {
        "S gemm": lambda b: ops.gemm(qp, ops.PackedWeight.wrap(b[0].data, M, C), S, M=P, N=M, K=C, lda=C, ldc=cap, alpha=1 / 32., bias=b_bank, ln=ln),
        "S gemm + stats": lambda b: ops.gemm(qp, ops.PackedWeight.wrap(b[0].data, M, C), S, M=P, N=M, K=C, lda=C, ldc=cap, alpha=1 / 32., bias=b_bank, ln=ln, sm_stats_out=st),
        **{"S gemm + stats, tile %d" % t: (lambda b, t=t: ops.gemm(qp, ops.PackedWeight.wrap(b[0].data, M, C), S, M=P, N=M, K=C, lda=C, ldc=cap, alpha=1 / 32., bias=b_bank, ln=ln, sm_stats_out=st, tile=t)) for t in (22, 23, 24, 25)},
        **{"PV gemm (packed P), tile %d" % t: (lambda b, t=t: ops.gemm(ops.PackedAct(P, Kp, wdt, dev, data=pk), ops.PackedWeight.wrap(b[1].data, C, cap), out, M=P, N=C, K=Kp, lda=Kp, ldc=C, ldw=cap, res1=q, ldr1=C, tile=t)) for t in (23, 24, 25)},
        "softmax_pack (2 launches)": lambda b: ops.softmax_pack(S, pk, rowstat, ld=cap, rows=P, M=M, thresh=5e-4, batch=1, strideS=P * cap, stride_packed=pk.numel()),
        "softmax_thresh": lambda b: ops.softmax_thresh(S, None, ld=cap, rows=P, M=M, Mpad=M, thresh=5e-4, batch=1, strideS=P * cap, packed=pk, stride_packed=pk.numel()),
        "PV gemm (packed P)": lambda b: ops.gemm(ops.PackedAct(P, Kp, wdt, dev, data=pk), ops.PackedWeight.wrap(b[1].data, C, cap), out, M=P, N=C, K=Kp, lda=Kp, ldc=C, ldw=cap, res1=q, ldr1=C),
        "colsum_packed": lambda b: ops.colsum_packed(pk, P, M, attn),
        "PV gemm (packed P, split-K 16 partials)": lambda b: ops.gemm(ops.PackedAct(P, Kp, wdt, dev, data=pk), ops.PackedWeight.wrap(b[1].data, C, cap), part, M=P, N=C, K=Kp, lda=Kp, ldc=C, ldw=cap, splitk=16),
        "reduce (16 partials + q)": lambda b: ops.reduce_ln(part, 16, P, C, res=q, ldres=C, x_out=out, ldx=C),
        "PV gemm (softmax loader)": lambda b: ops.gemm(S, ops.PackedWeight.wrap(b[1].data, C, cap), out, M=P, N=C, K=M, lda=cap, ldc=C, ldw=cap, res1=q, ldr1=C, softmax=(st, 5e-4, zk)),
        "PV gemm (softmax loader, 32x32)": lambda b: ops.gemm(S, ops.PackedWeight.wrap(b[1].data, C, cap), out, M=P, N=C, K=M, lda=cap, ldc=C, ldw=cap, res1=q, ldr1=C, softmax=(st, 5e-4, zk), tile=1),
        "colsum_softmax": lambda b: ops.colsum_softmax(S, cap, P, M, zk, 5e-4, attn),
    }
    if P > 256:
        steps.update({
            "prob: score stage (p~ + group stats)": lambda b: ops.gemm(qp, ops.PackedWeight.wrap(b[0].data, M, C), pt, M=P, N=M, K=C, lda=C, ldc=cap, alpha=1 / 32., bias=b_bank, ln=ln, sm_stats_out=pstats),
            "prob: merge": lambda b: ops.prob_merge(pstats, pscale, P, M, cap),
            **{"prob: P.V stage (split-K %d, rescale)" % k: (lambda b, k=k: ops.gemm(pt, ops.PackedWeight.wrap(b[1].data, C, cap), part, M=P, N=C, K=M, lda=cap, ldc=C, ldw=cap, splitk=k, softmax=(pscale, 0.0, None))) for k in (8, 16)},
            **{"prob: reduce (%d partials + q)" % k: (lambda b, k=k: ops.reduce_ln(part, k, P, C, res=q, ldres=C, x_out=out, ldx=C)) for k in (8, 16)},
            "prob: colsum": lambda b: ops.colsum_prob(pk, pscale, P, M, cap, attn),
        })
    for name, fn in steps.items():
        if long_bank and not name.startswith("prob") and ("tile 2" in name or "softmax loader" in name or "stats" in name or name == "colsum_softmax" or name == "PV gemm (packed P)"):
            continue                                   # long banks run: S gemm, softmax_thresh, split-K PV, reduce, colsum_packed
        if not long_bank and ("split-K" in name or "reduce" in name) and not name.startswith("prob"):
            continue
        for b in banks[:3]: fn(b)
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            for b in banks: fn(b)
        gr.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): gr.replay()
        e1.record(); torch.cuda.synchronize()
        print("%-34s %7.2f us" % (name, e0.elapsed_time(e1) * 1e3 / (5 * len(banks))))

if __name__ == "__main__":
    main()
