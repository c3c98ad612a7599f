import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from spann3r_amd import Spann3R, TINY, FULL, ops
from spann3r_amd.weights import synth_state_dict, synth_frames
cfg = FULL if len(sys.argv) > 1 and sys.argv[1] == "full" else TINY
nfr = int(sys.argv[2]) if len(sys.argv) > 2 else 2
m = Spann3R(dus3r_name=None, cfg=cfg, init_weights=False); m.load_state_dict(synth_state_dict(0, cfg)); m = m.cuda().eval()
S = 224 if cfg is FULL else 64
frames = [{"img": f["img"].cuda()} for f in synth_frames(nfr, S, S)]
m.set_precision("bf16"); m.use_graphs = False
snaps = []
for it in range(3):
    out = m(frames)[0]
    torch.cuda.synchronize()
    eng = m.engine
    snap = {}
    for k, t in eng._ws.items():
        d = t.data if isinstance(t, ops.PackedAct) else t
        snap[str(k[:2])] = d.clone()
    snap["conf_last"] = out[-1]["conf"].clone()
    snaps.append(snap)
for it in (1, 2):
    bad = []
    for k in snaps[0]:
        a, b = snaps[0][k], snaps[it][k]
        if a.shape == b.shape and not torch.equal(a, b):
            if a.dtype in (torch.float32, torch.bfloat16):
                bad.append((k, float((a.float() - b.float()).abs().max())))
            else:
                bad.append((k, -1))
    print("run", it, "differing buffers:", len(bad))
    if it == 1:
        for k, d in sorted(bad):
            print("   %-60s %.4g" % (k, d))
        same = sorted(k for k in snaps[0] if k not in dict(bad))
        print("SAME:", " ".join(k.split(",")[0].replace("('packed'", "P").replace("(", "") for k in same))
