"""Forensics for the multi-stream nondeterminism: when a rope'd K/Q element of layer 0 comes out wrong, which term of
   out = rope(rstd*acc - rstd*mean*s + b) explains the wrong value?  (debug probe)"""
import sys, os, dataclasses; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from spann3r_amd import Spann3R, FULL, ops
from spann3r_amd.weights import synth_state_dict
sd = synth_state_dict(0, FULL)
m = Spann3R(dus3r_name=None, cfg=FULL, init_weights=False); m.load_state_dict(sd); m = m.cuda().eval()
m.set_precision("bf16")
eng = m.engine
eng.cfg = dataclasses.replace(eng.cfg, dec_depth=1)
torch.manual_seed(0)
f1 = torch.randn(1, 196, 1024, device="cuda"); f2 = torch.randn(1, 196, 1024, device="cuda")
eng.positions(1, 14, 14)
def data(t):
    return t.data if isinstance(t, ops.PackedAct) else t
def run(conc):
    main = torch.cuda.current_stream()
    st = eng.side_streams()
    if conc:
        st[3].wait_stream(main)
        with torch.cuda.stream(st[3]):
            eng._vit(eng.wsp("im2col_pre", 196, 768), 196, 1, 196, "patch", "enc", 24, eng.positions(1, 14, 14)[1], tag="_pre")
    eng.decoder(f1, f2, 1, 14, 14, 14, 14, streams=st)
    if conc:
        main.wait_stream(st[3])
    torch.cuda.synchronize()
    return {k: data(v).clone() for k, v in eng._ws.items() if "_pre" not in str(k[0] if k[0] != "packed" else k[1])}
run(True)
ref = run(False)
def get(d, name):
    return [v for k, v in d.items() if (k[1] if k[0] == "packed" else k[0]) == name][0]
D = 768
def truth(kind, s):
    o = 3 - s
    blk = "dust3r.%s.0." % ("dec_blocks" if s == 1 else "dec_blocks2")
    dev = "cuda"
    if kind == "ckp":
        x = get(ref, "dec%d_l0" % o)
        W = sd[blk + "cross_attn.projk.weight"].to(dev).float(); bias = sd[blk + "cross_attn.projk.bias"].to(dev).float()
        g = sd[blk + "norm_y.weight"].to(dev).float(); beta = sd[blk + "norm_y.bias"].to(dev).float()
    else:
        x = get(ref, "dec%d_l0" % s)
        W = sd[blk + "attn.qkv.weight"].to(dev).float()[:2 * D]; bias = sd[blk + "attn.qkv.bias"].to(dev).float()[:2 * D]
        g = sd[blk + "norm1.weight"].to(dev).float(); beta = sd[blk + "norm1.bias"].to(dev).float()
    x = x.double()
    mean = x.mean(1, keepdim=True); var = (x * x).mean(1, keepdim=True) - mean * mean
    rstd = 1.0 / torch.sqrt(var + 1e-6)
    Wf = (W * g[None]).bfloat16().double()
    sv = Wf.sum(1); b = (bias + W @ beta).double()
    xb = x.float().bfloat16().double()
    parts = torch.stack([xb[:, kb * 64:(kb + 1) * 64] @ Wf[:, kb * 64:(kb + 1) * 64].T for kb in range(12)])   # [12, R, N]
    acc = parts.sum(0)
    y = rstd * acc - rstd * mean * sv[None] + b[None]
    return dict(y=y, acc=acc, parts=parts, mean=mean, rstd=rstd, s=sv, b=b)
pos = eng.positions(1, 14, 14)[1].cpu()
cosT, sinT = eng.cos.cpu().double(), eng.sin.cpu().double()
found = 0
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 400):
    out = run(True)
    for k in ref:
        name = k[1] if k[0] == "packed" else k[0]
        if name[:3] not in ("ckp", "qkp"):
            continue
        a, b_ = ref[k].float().flatten(), out[k].float().flatten()
        idx = (a != b_).nonzero().flatten().tolist()
        if not idx:
            continue
        found += 1
        kind, s = name[:3], int(name[-1])
        T = truth(kind, s)
        nkb = ref[k].shape[1]
        print("run", it, name, "bad elements:", len(idx))
        bad = {}
        for i in idx:
            e, r, g, kb, rb = i % 16, (i // 16) % 16, (i // 256) % 4, (i // 1024) % nkb, i // (1024 * nkb)
            bad[(rb * 16 + r, kb * 64 + g * 16 + e)] = (a[i].item(), b_[i].item())
        done = set()
        for (row, col), (rv, ov) in sorted(bad.items()):
            c0 = col & ~16
            if (row, c0) in done:
                continue
            done.add((row, c0))
            c1 = c0 | 16
            hc = c0 & 63; axis = hc >> 5; i0 = hc & 15
            p = int(pos[row, axis]); cs, sn = cosT[p, i0].item(), sinT[p, i0].item()
            A, B = T["y"][row, c0].item(), T["y"][row, c1].item()
            t0, t1 = A * cs - B * sn, B * cs + A * sn
            w0 = bad.get((row, c0), (None, t0))[1]; w1 = bad.get((row, c1), (None, t1))[1]
            A2, B2 = w0 * cs + w1 * sn, -w0 * sn + w1 * cs
            rs, mu = T["rstd"][row].item(), T["mean"][row].item()
            print("  row %3d cols %4d/%4d e=%d pos=%2d cs=%.4f sn=%.4f | true out %.4f %.4f  wrong %.4f %.4f | A %.4f->%.4f  B %.4f->%.4f | dA %.4f dB %.4f"
                  % (row, c0, c1, c0 % 4, p, cs, sn, t0, t1, w0, w1, A, A2, B, B2, A2 - A, B2 - B))
            for nm, c in (("A", c0), ("B", c1)):
                acc = T["acc"][row, c].item()
                wave = [T["parts"][w::4, row, c].sum().item() * rs for w in range(4)]
                print("      %s: rstd*acc %.4f  rstd*mean*s %.4f  bias %.4f  rstd %.4f mean %.4f | rstd*partial(wave0..3) %s"
                      % (nm, rs * acc, rs * mu * T["s"][c].item(), T["b"][c].item(), rs, mu, ["%.4f" % v for v in wave]))
    if found >= 3:
        break
print("runs with diffs:", found)
