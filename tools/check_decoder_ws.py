"""Which workspace buffer goes wrong first when the decoder runs next to a busy third stream?  (debug probe)"""
import sys, os, dataclasses; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from spann3r_amd import Spann3R, FULL, ops
from spann3r_amd.weights import synth_state_dict
m = Spann3R(dus3r_name=None, cfg=FULL, init_weights=False); m.load_state_dict(synth_state_dict(0, FULL)); m = m.cuda().eval()
m.set_precision("bf16")
eng = m.engine
depth = int(sys.argv[1]) if len(sys.argv) > 1 else 2
eng.cfg = dataclasses.replace(eng.cfg, dec_depth=depth)
torch.manual_seed(0)
f1 = torch.randn(1, 196, 1024, device="cuda"); f2 = torch.randn(1, 196, 1024, device="cuda")
img = torch.randn(1, 3, 224, 224, device="cuda")
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
ref2 = run(False)
print("serial-vs-serial diffs:", [k for k in ref if not torch.equal(ref[k].view(torch.uint8), ref2[k].view(torch.uint8))])
found = 0
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 60):
    out = run(True)
    bad = [k for k in ref if k in out and not torch.equal(ref[k].view(torch.uint8), out[k].view(torch.uint8))]
    if bad:
        found += 1
        print("run", it, "differing buffers:")
        for k in bad:
            a, b = ref[k].float().flatten(), out[k].float().flatten()
            idx = (a != b).nonzero().flatten()
            if idx.numel() < 200:
                print("      idx/ref/out:", [(int(i), round(a[i].item(), 4), round(b[i].item(), 4)) for i in idx[:64]])
            print("   ", k, "n=%d of %d" % (idx.numel(), a.numel()), "first idx", idx[:6].tolist(), "last", idx[-1].item(),
                  "maxabs %.3g" % (a - b).abs().max().item())
        if found >= 3:
            break
print("runs with diffs:", found)
